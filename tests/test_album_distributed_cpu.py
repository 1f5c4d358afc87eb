"""The N > 1 album path on CPU: world_size-2 gloo processes exercise the sharding, the histogram /
peak all-reduce and the input-order gather exactly as bench.py --gpus N and a multi-GPU host would,
with the per-track histograms coming from the oracle instead of the GPU (no GPU here)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

RATE = 44100
LENS = [44100 * 2, 30000, 2205 * 9, 44100 + 17, 50, 44100 * 3, 0]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _track_hist(po, t):
    n = LENS[t]
    l, r = po.synth_f32(900 + t, 0, RATE, n), po.synth_f32(900 + t, 1, RATE, n)
    res, h = po.analyze_pcm(l, r, RATE)
    return res, h


def _worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    from mp3rgain_amd import album
    from oracle import pyoracle as po

    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = album.shard_indices(len(LENS), world, rank)
    local_results, hist, peak = [], np.zeros(12000, dtype=np.uint32), 0.0
    for t in mine:
        res, h = _track_hist(po, t)
        local_results.append((t, res["loudness_db"], res["peak"]))
        hist += h  # what rg_album_merge_kernel does on the device for this rank's tracks
        peak = max(peak, res["peak"])
    hist_t = torch.from_numpy(hist.view(np.int32).copy())
    peak_t = torch.tensor([peak], dtype=torch.float64)
    album.allreduce_album(hist_t, peak_t)
    merged = hist_t.numpy().view(np.uint32)
    # the single-collective variant bench.py uses: all-gather the [histogram | peak] packs, fold locally
    pack = np.zeros(album.ALBUM_PACK_WORDS, dtype=np.uint32)
    pack[:12000] = hist
    pack[12000:] = np.array([peak], dtype=np.float64).view(np.uint32)
    pack_t = torch.from_numpy(pack.view(np.int32).copy())
    gathered_t = torch.empty(world * album.ALBUM_PACK_WORDS, dtype=torch.int32)
    album.allgather_album(pack_t, gathered_t)
    h2, p2 = album.fold_gathered(gathered_t.numpy().view(np.uint32), world)
    assert np.array_equal(h2, merged) and p2 == float(peak_t.item())
    alb = album.album_result_from_hist(merged, float(peak_t.item()))
    ordered = album.gather_track_results(local_results, len(LENS))
    np.save(Path(outdir) / f"hist_{rank}.npy", merged)
    import json

    (Path(outdir) / f"album_{rank}.json").write_text(json.dumps({"album": alb, "ordered": ordered}))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_indices_cover_the_album_once():
    from mp3rgain_amd import album

    for n in (0, 1, 7, 8, 8000):
        for world in (1, 2, 3, 8):
            seen = sorted(t for r in range(world) for t in album.shard_indices(n, world, r))
            assert seen == list(range(n))
    assert album.shard_indices(8000, 8, 3)[:3] == [3, 11, 19] and len(album.shard_indices(8000, 8, 3)) == 1000
    with pytest.raises(ValueError):
        album.shard_indices(4, 2, 2)


def test_two_rank_album_matches_the_sequential_reference(tmp_path, oracle, capi):
    import json

    import torch.multiprocessing as mp

    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)

    # the reference's sequential album (src/replaygain.rs:1048-1066) via the oracle
    wants = [_track_hist(oracle, t) for t in range(len(LENS))]
    want_alb, want_hist = oracle.album_from_hists([h for _, h in wants], [r["peak"] for r, _ in wants])
    for rank in range(world):
        got_hist = np.load(tmp_path / f"hist_{rank}.npy")
        got = json.loads((tmp_path / f"album_{rank}.json").read_text())
        assert np.array_equal(got_hist, want_hist)  # merged histogram == sum of per-track histograms
        assert got["album"]["album_loudness_db"] == want_alb["album_loudness_db"]
        assert got["album"]["album_gain_db"] == want_alb["album_gain_db"]
        assert got["album"]["album_peak"] == want_alb["album_peak"]
        assert got["album"]["windows"] == int(want_hist.sum())
        # per-track results come back in input order on every rank
        assert [o[0] for o in got["ordered"]] == list(range(len(LENS)))
        assert [o[1] for o in got["ordered"]] == [r["loudness_db"] for r, _ in wants]


def _abort_worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    from mp3rgain_amd import album

    dist.init_process_group("gloo", rank=rank, world_size=world)
    # rank 1 cannot decode one of its files; rank 0 is fine.  Both must give the album up, neither may hang.
    err = FileNotFoundError("Failed to open: /music/07.mp3") if rank == 1 else None
    outcome = "ok"
    try:
        album.abort_if_any_failed(err)
        h = torch.zeros(12000, dtype=torch.int32)
        album.allreduce_album(h, torch.zeros(1, dtype=torch.float64))  # must not be reached
    except album.AlbumAborted as ex:
        outcome = str(ex)
    (Path(outdir) / f"abort_{rank}.txt").write_text(outcome)
    # a clean second album on the same group works (nothing was left half-way in a collective)
    album.abort_if_any_failed(None)
    t = torch.ones(1, dtype=torch.int32)
    dist.all_reduce(t)
    assert int(t) == world
    dist.destroy_process_group()


def test_failing_rank_aborts_the_album_on_every_rank(tmp_path, capi):
    """src/replaygain.rs:1055: the first failing track ends analyze_album for good.  Sharded, every rank must learn of
    it before the exchange."""
    import torch.multiprocessing as mp

    world, port = 2, _free_port()
    mp.spawn(_abort_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for rank in range(world):
        assert (tmp_path / f"abort_{rank}.txt").read_text() == "rank 1: Failed to open: /music/07.mp3"


def test_shards_balance_by_frames():
    """shards balanced by total frames, not by count; results still come back in input order."""
    from mp3rgain_amd import album

    frames = [600] + [10] * 60 + [300, 300]  # one 10-minute track among sixty 10-second ones and two 5-minute ones
    world = 4
    shards = [album.shard_indices(len(frames), world, r, frames=frames) for r in range(world)]
    assert sorted(t for s in shards for t in s) == list(range(len(frames)))
    loads = [sum(frames[t] for t in s) for s in shards]
    assert max(loads) == 600 and min(loads) >= 190        # round robin would give one rank 600 + 15 * 10 + ... = 900
    rr = [sum(frames[t] for t in album.shard_indices(len(frames), world, r)) for r in range(world)]
    assert max(rr) > max(loads)
    assert all(s == sorted(s) for s in shards)             # each rank walks its tracks in input order
    assert shards == [album.shard_indices(len(frames), world, r, frames=frames) for r in range(world)]  # deterministic
    with pytest.raises(ValueError):
        album.shard_indices(3, 2, 0, frames=[1, 2])


def test_single_process_allreduce_is_a_no_op(capi):
    import torch

    from mp3rgain_amd import album

    h = torch.arange(12000, dtype=torch.int32)
    p = torch.tensor([0.5], dtype=torch.float64)
    album.allreduce_album(h, p)
    assert int(h[11999]) == 11999 and float(p) == 0.5
    r = album.album_result_from_hist(np.zeros(12000, dtype=np.uint32), 0.0)
    assert r["album_loudness_db"] == -20.0 and r["album_gain_db"] == pytest.approx(84.82)
