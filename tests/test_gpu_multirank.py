"""The library's multi-rank code on a ONE-GPU box.

RCCL refuses a communicator with two ranks on one device, so on this build's lease `rg_comm_init(world > 1)`, the all-gather
of several DIFFERENT [histogram | peak] packs on the batch's stream, `rg_comm_init_all` over several contexts, and the
world > 1 branch of bench.py had never executed anywhere before the driver's 8-GPU run.  Here they do: tests/standin_rccl is a
stand-in for the few librccl.so entry points the library resolves (shared memory between ranks that all sit on device 0),
handed to the library through `rg_comm_library`; everything above the collective call is the product's own code -- the
bootstrap of the unique id over torch.distributed, the sharding, the joint abort, the device fold, the percentile on every
rank, the gather of the per-track results in input order.  What this does NOT show is RCCL or xGMI themselves.

Every case runs in subprocesses (the stand-in must not leak into the other tests' process)."""
import json
import os
import shutil
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
STANDIN = ROOT / "tests" / "standin_rccl"
sys.path.insert(0, str(ROOT / "tests"))
import mp3gold  # noqa: E402

pytestmark = pytest.mark.gpu
RATE = 44100


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture(scope="module")
def standin():
    lib = STANDIN / "librccl_standin.so"
    if not lib.exists():
        subprocess.run(["make", "-C", str(STANDIN)], check=True)
    return lib


def _files(tmp_path):
    from wavutil import test_signal, wav_bytes

    srcs = [p for p in mp3gold.STREAMS if p.stat().st_size < 200000]
    files = []
    for k in range(2):
        for p in srcs:
            f = tmp_path / f"{k}_{p.name}"
            shutil.copyfile(p, f)
            files.append(f)
    w = tmp_path / "x.wav"
    w.write_bytes(wav_bytes(test_signal("s16", 48000, 48000 * 3 + 17, 2, seed=5), 48000, "s16"))
    files.insert(7, w)
    return files


def _run_ranks(world, files, tmp_path, tag):
    port = _port()
    outs = [tmp_path / f"{tag}_rank{r}.json" for r in range(world)]
    procs = [subprocess.Popen([sys.executable, str(STANDIN / "rank_album.py"), str(r), str(world), str(port), str(outs[r])]
                              + [str(f) for f in files], cwd=str(ROOT), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=300)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{logs[r][-3000:]}"
    return [json.loads(o.read_text()) for o in outs]


def _track_tuple(t):
    return [t.loudness_db, t.gain_db, t.peak, t.sample_rate, t.windows, int(t.file_type)]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_album_over_a_communicator_of_several_ranks(_ctx, standin, tmp_path, world):
    """analyze_album_files_sharded with world ranks: rg_comm_init(world, rank) on every rank, each rank's share analysed on its
    own context, ONE all-gather of `world` different packs + device fold on the batch's stream, the percentile on every rank.
    Every rank must return the single-context album, bit for bit, with the tracks in input order."""
    files = _files(tmp_path)
    _ctx.set_kernel(0)
    want = _ctx.analyze_album_files(files)
    outs = _run_ranks(world, files, tmp_path, f"w{world}")
    # the histogram the ranks merged is the sum of the per-file histograms of the single-context run
    res = _ctx.analyze_track_files(files)
    assert [_track_tuple(t) for t in res] == [_track_tuple(t) for t in want.tracks]
    for o in outs:
        assert "aborted" not in o
        assert o["album"] == [want.album_loudness_db, want.album_gain_db, want.album_peak]
        assert o["tracks"] == [_track_tuple(t) for t in want.tracks]
        assert o["again"][:3] == o["album"]
        assert o["hist_nonzero"] == outs[0]["hist_nonzero"] and sum(o["hist_nonzero"].values()) == o["again"][3]


def test_every_rank_aborts_when_one_rank_has_a_bad_file(_ctx, standin, tmp_path):
    """src/replaygain.rs:1055: the first failing track ends the album.  The rank that owns the bad file must not leave the other
    in the collective: nobody calls rg_album_exchange, every rank reports the failure."""
    files = _files(tmp_path)[:6]
    junk = tmp_path / "junk.mp3"
    junk.write_bytes(b"ID3" + bytes(5000))
    files.insert(2, junk)
    outs = _run_ranks(2, files, tmp_path, "abort")
    for o in outs:
        assert "album" not in o and "Failed to probe format" in o["aborted"]


def test_node_rccl_mode_over_three_contexts(_ctx, standin, tmp_path):
    """rg_node_set_exchange(RG_NODE_EXCHANGE_RCCL) with three contexts: rg_comm_init_all, three host threads inside the
    all-gather at once, the fold of three different packs, against the single-context album and the node's own host fold."""
    files = _files(tmp_path)
    _ctx.set_kernel(0)
    out_path = tmp_path / "node.json"
    p = subprocess.run([sys.executable, str(STANDIN / "node_album.py"), "3", str(out_path)] + [str(f) for f in files], cwd=str(ROOT),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-3000:]
    o = json.loads(out_path.read_text())
    for key, lst in (("all", files), ("five", files[:5]), ("one", files[3:4])):
        want = _ctx.analyze_album_files(lst)
        assert o[key]["album"] == [want.album_loudness_db, want.album_gain_db, want.album_peak], key
        assert o[key]["tracks"] == [_track_tuple(t) for t in want.tracks], key
    assert o["all"]["owners"] == [0, 1, 2]
    assert o["host_fold"]["album"] == o["all"]["album"]


def _bench_env(standin):
    env = dict(os.environ)
    env.update({"RG_BENCH_REHEARSAL": "1", "MP3RGAIN_AMD_RCCL_LIBRARY": str(standin), "MASTER_ADDR": "127.0.0.1"})
    return env


def _json_line(text):
    for line in reversed(text.splitlines()):
        line = line.strip()
        if line.startswith("{") and '"metric"' in line:
            return json.loads(line)
    raise AssertionError("no JSON line:\n" + text[-3000:])


def _album_oracle(oracle, n_tracks, frames):
    hist = np.zeros(12000, dtype=np.uint32)
    peak = 0.0
    for g in range(n_tracks):
        r, h = oracle.analyze_pcm(oracle.synth_f32(0x5EED0000 + g, 0, RATE, frames), oracle.synth_f32(0x5EED0000 + g, 1, RATE, frames), RATE)
        hist += h
        peak = max(peak, r["peak"])
    return oracle.hist_loudness(hist), peak


@pytest.mark.parametrize("world", [2, 8])
def test_bench_rank_mode_rehearsal(standin, oracle, world):
    """The driver's own command line for N > 1 (`python -m torch.distributed.run ... bench.py --gpus N`), rehearsed with all
    ranks on device 0: the world > 1 branch of bench.py -- sharding, library communicator, barrier + max-over-ranks timing,
    sum of frames over ranks, rank 0 alone prints -- runs to its ONE contract line, and the album it reports is the oracle's
    merge over every rank's tracks."""
    tracks_per_rank, minutes = 3, 0.1
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), "bench.py", "--gpus", str(world), "--steps", "4", "--warmup", "2", "--tracks-per-rank",
           str(tracks_per_rank), "--minutes", str(minutes), "--cpu-seconds", "0.2", "--pre-roll", "0.001", "--parity-tracks", "2"]
    p = subprocess.run(cmd, cwd=str(ROOT), env=_bench_env(standin), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip().startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, "exactly one rank prints the line"
    o = _json_line(p.stdout)
    frames = int(round(minutes * 60 * RATE))
    assert o["n_gpus"] == world and o["steps"] == 4 and o["warmup"] == 2 and o["scaling"] == "weak" and "rehearsal" in o
    assert o["config"]["total_tracks"] == tracks_per_rank * world and o["config"]["tracks_per_gpu"] == tracks_per_rank
    assert o["config"]["exchange"].startswith("rccl (library communicator")
    # value = the frames of ALL ranks over the slowest rank's time
    assert o["value"] == pytest.approx(tracks_per_rank * world * frames * 4 / (o["ms_per_step"] * 4e-3), rel=1e-9)
    loud, peak = _album_oracle(oracle, tracks_per_rank * world, frames)
    assert o["result"]["album_loudness_db"] == loud and o["result"]["album_peak"] == peak
    assert o["parity"]["differing_histogram_bins"] == 0 and o["parity"]["peaks_equal"]
    # what an auditor of the first real SCALE run needs in the line itself: the communicator's size, the transport, the CPU leg
    assert o["exchange"]["ranks"] == world and o["exchange"]["transport"].startswith("rccl") and "nccl_version" in o["exchange"]
    assert o["cpu_baseline"]["cores"] == 1 and o["cpu_baseline"]["value"] > 0


def test_bench_strong_scaling_rehearsal(standin, oracle):
    """`--scaling strong --total-tracks T`: the album is fixed and sharded by cumulative frames; the line says so."""
    total, minutes, world = 7, 0.1, 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), "bench.py", "--gpus", str(world), "--steps", "3", "--warmup", "1", "--scaling", "strong",
           "--total-tracks", str(total), "--minutes", str(minutes), "--cpu-seconds", "0", "--pre-roll", "0.001"]
    p = subprocess.run(cmd, cwd=str(ROOT), env=_bench_env(standin), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:]
    o = _json_line(p.stdout)
    frames = int(round(minutes * 60 * RATE))
    assert o["scaling"] == "strong" and o["config"]["total_tracks"] == total and o["exchange"]["ranks"] == world
    assert o["value"] == pytest.approx(total * frames * 3 / (o["ms_per_step"] * 3e-3), rel=1e-9)
    loud, peak = _album_oracle(oracle, total, frames)
    assert o["result"]["album_loudness_db"] == loud and o["result"]["album_peak"] == peak


def test_bench_node_mode_rehearsal(standin, oracle):
    """`bench.py --node --gpus 3` with three contexts on device 0: in-process communicators, one host thread per context."""
    tracks_per_rank, minutes = 2, 0.1
    cmd = [sys.executable, "bench.py", "--node", "--gpus", "3", "--steps", "4", "--warmup", "2", "--tracks-per-rank", str(tracks_per_rank),
           "--minutes", str(minutes), "--pre-roll", "0.001", "--cpu-seconds", "0.2", "--parity-tracks", "2"]
    p = subprocess.run(cmd, cwd=str(ROOT), env=_bench_env(standin), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:]
    o = _json_line(p.stdout)
    frames = int(round(minutes * 60 * RATE))
    assert o["n_gpus"] == 3 and "rehearsal" in o and o["result"]["every_device_agrees"]
    assert o["exchange"]["ranks"] == 3 and o["exchange"]["transport"].startswith("rccl")
    assert o["parity"]["differing_histogram_bins"] == 0 and o["parity"]["peaks_equal"] and o["parity"]["every_device_agrees_on_the_album"]
    assert o["cpu_baseline"]["cores"] == 1 and o["cpu_baseline"]["value"] > 0
    loud, _ = _album_oracle(oracle, tracks_per_rank * 3, frames)
    assert o["result"]["album_loudness_db"] == loud
