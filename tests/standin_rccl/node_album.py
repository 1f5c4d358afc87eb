#!/usr/bin/env python3
"""rg_node's RCCL exchange with several contexts on ONE device (test helper): ncclCommInitAll over N contexts, one host
thread per context in ncclAllGather at the same time, the fold of N different packs, every context running the percentile
-- over the stand-in transport of this directory (RCCL refuses the same device twice).

    node_album.py N_CONTEXTS OUT.json FILE [FILE ...]
"""
import json
import os
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402,F401

import mp3rgain_amd as rg  # noqa: E402
from mp3rgain_amd import _capi  # noqa: E402

n, out_path, files = int(sys.argv[1]), sys.argv[2], sys.argv[3:]
assert _capi.load().rg_comm_library(os.fsencode(str(HERE / "librccl_standin.so"))) == 0
out = {}
with rg.Node([0] * n) as node:
    node.set_exchange(rg.Node.EXCHANGE_RCCL)
    for key, lst in (("all", files), ("five", files[:5]), ("one", files[3:4])):
        res = node.analyze_album_files(lst)
        out[key] = {"album": [res.album_loudness_db, res.album_gain_db, res.album_peak],
                    "tracks": [[t.loudness_db, t.gain_db, t.peak, t.sample_rate, t.windows, int(t.file_type)] for t in res.tracks],
                    "owners": sorted(set(node.last_partition(len(lst))))}
    node.set_exchange(rg.Node.EXCHANGE_HOST)
    res = node.analyze_album_files(files)
    out["host_fold"] = {"album": [res.album_loudness_db, res.album_gain_db, res.album_peak]}
Path(out_path).write_text(json.dumps(out))
