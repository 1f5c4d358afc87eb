"""include/mp3rgain_amd_node.h on the GPU box, with the built-in engine (rg_ctx): a node over the one leased device must
return the bits of the single-context calls it wraps (host fold and RCCL exchange -- a one-rank communicator from
ncclCommInitAll), and a node of TWO contexts on the same device runs the real multi-device path -- files dealt out, two
host threads, two concurrent contexts, the fold of two packs -- against the same answers.  (A communicator of more than one
rank cannot be built on a one-GPU lease: RCCL rejects two ranks on one device.  The multi-rank exchange runs over a stand-in
transport in tests/test_gpu_multirank.py)"""
import shutil
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))
import mp3gold  # noqa: E402
from mp3rgain_amd import mp3dec  # noqa: E402

pytestmark = pytest.mark.gpu


def _files(tmp_path, copies=2):
    srcs = [p for p in mp3gold.STREAMS if p.stat().st_size < 200000]
    files = []
    for k in range(copies):
        for p in srcs:
            f = tmp_path / f"{k}_{p.name}"
            shutil.copyfile(p, f)
            files.append(f)
    sys.path.insert(0, str(ROOT / "tests"))
    from wavutil import test_signal, wav_bytes

    w = tmp_path / "x.wav"
    w.write_bytes(wav_bytes(test_signal("s16", 48000, 48000 * 3 + 17, 2, seed=5), 48000, "s16"))
    files.insert(7, w)
    return files


def _same(a, b):
    assert (a.album_loudness_db, a.album_gain_db, a.album_peak) == (b.album_loudness_db, b.album_gain_db, b.album_peak)
    assert len(a.tracks) == len(b.tracks)
    for x, y in zip(a.tracks, b.tracks):
        assert (x.loudness_db, x.gain_db, x.peak, x.sample_rate, x.windows, x.file_type) == (y.loudness_db, y.gain_db, y.peak, y.sample_rate, y.windows, y.file_type)


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]], ids=["one-context", "two-contexts", "three-contexts"])
def test_node_album_is_the_single_context_album(_ctx, oracle, tmp_path, devices):
    import mp3rgain_amd as rg

    files = _files(tmp_path)
    _ctx.set_kernel(0)
    want = _ctx.analyze_album_files(files)
    with rg.Node(devices) as node:
        assert node.devices == len(devices)
        got = node.analyze_album_files(files)
        _same(got, want)
        own = node.last_partition(len(files))
        assert set(own) == set(range(len(devices)))  # every context had work
        again = node.analyze_album_files(files[:5])
        _same(again, _ctx.analyze_album_files(files[:5]))
        empty = node.analyze_album_files([])
        assert empty.tracks == [] and empty.album_loudness_db == -20.0
        one = node.analyze_album_files(files[3:4])  # fewer files than contexts: the others contribute empty packs
        _same(one, _ctx.analyze_album_files(files[3:4]))
    # and the album is the oracle's merge of the per-file histograms on the host decoder's PCM
    per = []
    for f in files:
        if f.suffix == ".wav":
            continue
        pcm, info = mp3dec.decode(f.read_bytes())
        per.append(oracle.analyze_pcm(pcm[0], pcm[1] if info.channels == 2 else None, info.sample_rate))
    mp3_only = [f for f in files if f.suffix != ".wav"]
    with rg.Node(devices) as node:
        got = node.analyze_album_files(mp3_only)
    ref, _ = oracle.album_from_hists([h for _, h in per], [r["peak"] for r, _ in per])
    assert (got.album_loudness_db, got.album_gain_db, got.album_peak) == (ref["album_loudness_db"], ref["album_gain_db"], ref["album_peak"])
    assert [t.loudness_db for t in got.tracks] == [r["loudness_db"] for r, _ in per]


def test_node_rccl_exchange_with_a_one_rank_communicator(_ctx, tmp_path):
    """RG_NODE_EXCHANGE_RCCL on the one leased device: ncclCommInitAll(1 device) -> ncclAllGather on the batch's stream ->
    device fold -> device percentile.  Same bits as the host fold."""
    import mp3rgain_amd as rg

    files = _files(tmp_path, copies=1)
    _ctx.set_kernel(0)
    want = _ctx.analyze_album_files(files)
    with rg.Node([0]) as node:
        node.set_exchange(rg.Node.EXCHANGE_RCCL)
        _same(node.analyze_album_files(files), want)
        _same(node.analyze_album_files(files[:3]), _ctx.analyze_album_files(files[:3]))
        node.set_exchange(rg.Node.EXCHANGE_HOST)
        _same(node.analyze_album_files(files), want)


def test_node_album_aborts_on_the_first_failing_file_in_input_order(_ctx, tmp_path):
    import mp3rgain_amd as rg

    files = _files(tmp_path, copies=1)
    bad_rate = tmp_path / "odd.wav"
    sys.path.insert(0, str(ROOT / "tests"))
    from wavutil import test_signal, wav_bytes

    bad_rate.write_bytes(wav_bytes(test_signal("s16", 44000, 5000, 2, seed=1), 44000, "s16"))
    junk = tmp_path / "junk.mp3"
    junk.write_bytes(b"ID3" + bytes(5000))
    with rg.Node([0, 0]) as node:
        for order, text, code in (([junk, bad_rate], "Failed to probe format", -9), ([bad_rate, junk], "Unsupported sample rate: 44000 Hz", -2)):
            lst = files[:4] + [order[0]] + files[4:9] + [order[1]] + files[9:] + [tmp_path / "missing.mp3"]
            with pytest.raises(rg.ReplayGainError, match=text) as ei:
                node.analyze_album_files(lst)
            assert ei.value.code == code
            # the single-context call reports the same file
            with pytest.raises(rg.ReplayGainError, match=text):
                _ctx.analyze_album_files(lst)
        with pytest.raises(rg.ReplayGainError, match="Track index 2 out of range"):
            node.analyze_album_files(files, track_index=2)
        _same(node.analyze_album_files(files), _ctx.analyze_album_files(files))  # still alive


def test_node_tracks_are_the_single_context_tracks(_ctx, tmp_path):
    import mp3rgain_amd as rg

    files = _files(tmp_path)
    files.insert(2, tmp_path / "missing.mp3")
    junk = tmp_path / "junk.mp3"
    junk.write_bytes(b"ID3" + bytes(500))
    files.insert(9, junk)
    _ctx.set_kernel(0)
    want = _ctx.analyze_track_files(files)
    with rg.Node([0, 0]) as node:
        got = node.analyze_track_files(files)
        single = node.analyze_track_file(files[0])
        pk = node.find_peak_amplitude_file(files[0])
    assert len(got) == len(want)
    for f, a, b in zip(files, got, want):
        if isinstance(b, rg.ReplayGainError):
            assert isinstance(a, rg.ReplayGainError) and (a.code, str(a)) == (b.code, str(b)), f.name
        else:
            assert (a.loudness_db, a.gain_db, a.peak, a.sample_rate, a.windows) == (b.loudness_db, b.gain_db, b.peak, b.sample_rate, b.windows), f.name
    assert (single.loudness_db, single.peak) == (want[0].loudness_db, want[0].peak)
    assert pk.peak == _ctx.find_peak_amplitude_file(files[0]).peak


def test_bench_one_process_mode_prints_the_contract_line():
    """`bench.py --node`: the album workload through rg_node in one process (a host thread per device, the library's
    in-process communicators).  One device is all a test box has; the line must carry the contract's fields and an album
    result every device agrees on."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    p = subprocess.run([sys.executable, str(root / "bench.py"), "--node", "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--tracks-per-rank", "8", "--minutes", "0.25", "--pre-roll", "0.001"],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])  # RCCL prints its banner on stdout too
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["value"] > 0
    assert "workload" in line["config"] and "ONE process" in line["config"]["launch"]
    assert line["result"]["every_device_agrees"] is True
