#!/usr/bin/env python3
"""One rank of a sharded album on a box with fewer GPUs than ranks (test helper; tests/test_gpu_multirank.py starts `world`
of these).  The control plane is torch.distributed over gloo, the data plane the library's own communicator -- rg_comm_init
with world > 1, rg_album_exchange: all-gather of the packs on the batch's stream + device fold -- over the stand-in transport
of this directory.  Every rank uses device 0.

    rank_album.py RANK WORLD PORT OUT.json FILE [FILE ...]
"""
import json
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
import torch.distributed as dist  # noqa: E402

import mp3rgain_amd as rg  # noqa: E402
from mp3rgain_amd import album  # noqa: E402

rank, world, port, out_path = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
files = sys.argv[5:]
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
out = {"rank": rank}
with rg.Analyzer(0) as an:
    an.comm_init_torch(library=str(HERE / "librccl_standin.so"))
    try:
        res = album.analyze_album_files_sharded(an, files)
        out["album"] = [res.album_loudness_db, res.album_gain_db, res.album_peak]
        out["tracks"] = [[t.loudness_db, t.gain_db, t.peak, t.sample_rate, t.windows, int(t.file_type)] for t in res.tracks]
        # the merged histogram itself, after a second album over the same communicator (every pipeline slot's gather buffer
        # is a different one): this rank's share again, exchanged, read back
        sizes = [Path(f).stat().st_size for f in files]
        mine = album.shard_indices(len(files), world, rank, frames=sizes)
        an.analyze_album_files([files[i] for i in mine])
        an.album_exchange()
        an.album_result_enqueue()
        alb, hist = an.album_finish(want_hist=True)
        out["again"] = [alb.album_loudness_db, alb.album_gain_db, alb.album_peak, int(alb.windows)]
        out["hist_nonzero"] = {int(i): int(hist[i]) for i in np.flatnonzero(hist)}
    except album.AlbumAborted as ex:
        out["aborted"] = str(ex)
    an.comm_destroy()
Path(out_path).write_text(json.dumps(out))
dist.barrier()
dist.destroy_process_group()
