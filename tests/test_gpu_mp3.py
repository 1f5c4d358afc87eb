"""The split MP3 decoder on the GPU box (include/mp3rgain_amd_dec.h, rg_mp3dev.hip): stage A (frame walk, side
information, bit reservoir, scalefactors, Huffman) on the host, stages B-E (requantisation, joint stereo, reordering,
alias reduction, IMDCT + overlap, polyphase synthesis) on the device.  The device half is written to reproduce the host
decoder's float arithmetic exactly (same tables, same summation order, no FMA contraction), so the bar is EQUALITY
with mp3dec.decode on every golden stream -- which in turn is pinned against ffmpeg's decoder by tests/test_mp3dec.py --
and the file-level entry points must give identical ReplayGain results with either decoder."""
import shutil
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from mp3rgain_amd import mp3dec  # noqa: E402

sys.path.insert(0, str(ROOT / "tests"))
import mp3gold  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = ROOT / "tests" / "golden" / "mp3"
FIX = ROOT / "tests" / "golden" / "fixtures"
STREAMS = sorted(GOLD.glob("*.mp3")) + sorted(FIX.glob("*.mp3"))


# tuning key 6: 1 = scalefactors + Huffman on the host, stages B-E on the device; 2 = the host walks the frames and parses
# their side information (where each granule's bits lie), scalefactors and Huffman run on the device as well; 3 = the
# host only strips headers and side information from the stream, the device parses them too
DEFAULT_ROUTE = 3


@pytest.fixture(params=[1, 2, 3], ids=["huffman-on-host", "huffman-on-device", "frames-on-device"])
def split_mode(_ctx, request):
    _ctx.set_tuning(6, request.param)
    yield request.param
    _ctx.set_tuning(6, DEFAULT_ROUTE)  # the library's default


@pytest.mark.parametrize("path", STREAMS, ids=lambda p: p.stem)
def test_device_half_reproduces_the_host_decoder(_ctx, path, split_mode):
    data = path.read_bytes()
    want, wi = mp3dec.decode(data)
    got, gi = _ctx.decode_mp3_device(data)
    assert (gi.frames, gi.channels, gi.sample_rate, gi.audio_frames, gi.skipped_frames) == (wi.frames, wi.channels, wi.sample_rate, wi.audio_frames, wi.skipped_frames)
    assert got.shape == want.shape
    if not np.array_equal(got, want):
        d = np.abs(got.astype(np.float64) - want.astype(np.float64))
        bad = np.argwhere(d > 0)
        raise AssertionError(f"{len(bad)} of {got.size} samples differ, max {d.max():.3g} (peak {np.abs(want).max():.3g}), first at {bad[0]}")


@pytest.mark.parametrize("path", mp3gold.STREAMS, ids=lambda p: p.stem)
def test_device_pcm_matches_ffmpeg(_ctx, oracle, path, split_mode, tmp_path):
    """The pin on the HIP decode path itself, not through the host decoder: PCM decoded on the device (every route)
    against ffmpeg's decode of the same stream (tests/golden/mp3/*.ffmpeg.np[yz]), max |delta| <= 1.5 and RMS <= 0.6
    steps of 2^-15 -- on the reference's fixtures, the sixteen syntax-walking streams and the dense music-like encodes
    (tools/make_mp3_dense.py).  Then the file-level entry point on the same file, packet semantics of
    src/replaygain.rs:881-904 (nothing trimmed): loudness within 0.1 dB of the oracle's on ffmpeg's PCM -- where ffmpeg
    trimmed the encoder delay by the Info header, its samples are laid over this decoder's at their place."""
    data = path.read_bytes()
    gold = mp3gold.load_gold(path)
    got, gi = _ctx.decode_mp3_device(data)
    assert gold.shape[0] == gi.channels == got.shape[0] and gi.skipped_frames == 0
    mx, rms, off, n = mp3gold.compare_with_gold(got, gi.info_frame, gold)
    assert mx <= mp3gold.MAX_STEPS, f"max |delta| {mx:.2f} steps of 2^-15"
    assert rms <= mp3gold.RMS_STEPS, f"rms {rms:.3f}"
    f = tmp_path / path.name
    f.write_bytes(data)
    _ctx.set_kernel(0)
    res = _ctx.analyze_track_file(f)
    ref = got.copy()
    ref[:, off:off + n] = (gold[:, :n].astype(np.float64) / 32768.0).astype(np.float32)
    want, _ = oracle.analyze_pcm(ref[0], ref[1] if ref.shape[0] == 2 else None, gi.sample_rate)
    assert abs(res.loudness_db - want["loudness_db"]) <= 0.1, (res.loudness_db, want["loudness_db"])
    assert abs(res.peak - want["peak"]) <= 2.0 / 32768.0
    assert res.sample_rate == gi.sample_rate


def test_dropped_and_damaged_frames_behave_like_the_host_decoder(_ctx, split_mode):
    """Frames the reservoir cannot serve are dropped by stage A in both decoders: same PCM length, same PCM."""
    import random

    body = (GOLD / "v1_44k_mono_crc_reservoir.mp3").read_bytes()
    cut = body[417 * 2 + 1:]  # starts inside the second frame: the first whole frames reach behind the reservoir
    want, wi = mp3dec.decode(cut)
    got, gi = _ctx.decode_mp3_device(cut)
    assert wi.skipped_frames >= 1 and gi.skipped_frames == wi.skipped_frames and np.array_equal(got, want)
    rng = random.Random(5)
    srcs = [p.read_bytes() for p in STREAMS if p.stat().st_size < 20000]
    checked = 0
    for _ in range(60):
        d = bytearray(rng.choice(srcs))
        for _ in range(rng.randint(1, 12)):
            d[rng.randrange(len(d))] = rng.randrange(256)
        try:
            want, wi = mp3dec.decode(bytes(d))
        except mp3dec.Mp3DecodeError:
            continue
        if wi.channels not in (1, 2) or wi.frames == 0:
            continue
        got, gi = _ctx.decode_mp3_device(bytes(d))
        assert (gi.frames, gi.audio_frames, gi.skipped_frames) == (wi.frames, wi.audio_frames, wi.skipped_frames)
        # damaged side information can ask for enormous gains: compare where the host's output is finite
        ok = np.isfinite(want)
        assert np.array_equal(got[ok], want[ok])
        checked += 1
    assert checked >= 30


def test_damaged_files_through_the_file_level_entry_point(_ctx, tmp_path):
    """Mutated and truncated files through rg_analyze_track on the device route and on the host route: both fail, or both
    return the same result; nothing hangs or crashes, and the context stays usable."""
    import random

    import mp3rgain_amd as rg

    an = _ctx
    rng = random.Random(99)
    srcs = [p.read_bytes() for p in STREAMS if p.stat().st_size < 20000]
    agree = failed = 0
    for k in range(40):
        d = bytearray(rng.choice(srcs))
        kind = rng.randrange(3)
        if kind == 0:
            for _ in range(rng.randint(1, 20)):
                d[rng.randrange(len(d))] = rng.randrange(256)
        elif kind == 1:
            d = d[:rng.randrange(8, len(d))]
        else:
            a = rng.randrange(len(d))
            del d[a:a + rng.randint(1, 900)]
        f = tmp_path / f"damaged{k}.mp3"
        f.write_bytes(bytes(d))
        out = []
        for route in (3, 2, 0):
            an.set_tuning(6, route)
            try:
                r = an.analyze_track_file(f)
                out.append((r.loudness_db, r.peak, r.sample_rate, r.windows))
            except rg.ReplayGainError as ex:
                out.append(("error", ex.code))
        an.set_tuning(6, DEFAULT_ROUTE)
        assert out[0] == out[1], (k, out)  # the two device routes run the same frame logic (rg_mp3_frame.h)
        out = out[1:]
        if out[0][0] == "error" or out[1][0] == "error":
            assert out[0][0] == out[1][0] == "error", (k, out)
            failed += 1
            continue
        assert out[0] == out[1], (k, out)  # every route drops the same frames (another channel count included)
        agree += 1
    assert agree >= 20
    good = an.analyze_track_file(FIX / "test_vbr.mp3")  # still alive
    assert good.sample_rate == 44100


def test_file_level_results_do_not_depend_on_the_decoder(_ctx, oracle, tmp_path, split_mode):
    """rg_analyze_track / rg_analyze_album / rg_find_peak_amplitude with tuning key 6: identical results, and they are the
    oracle's on the host decoder's PCM."""
    an = _ctx
    names = ["test_joint_stereo.mp3", "test_mono.mp3", "test_vbr.mp3"]
    files = []
    for n in names:
        shutil.copyfile(FIX / n, tmp_path / n)
        files.append(tmp_path / n)
    for p in sorted(GOLD.glob("v1_44k_*.mp3"))[:3] + sorted(GOLD.glob("v2_*.mp3"))[:2]:
        shutil.copyfile(p, tmp_path / p.name)
        files.append(tmp_path / p.name)
    an.set_kernel(0)
    an.set_tuning(6, 0)
    host = [an.analyze_track_file(f) for f in files]
    host_album = an.analyze_album_files(files[:3])
    host_peak = an.find_peak_amplitude_file(files[0])
    an.set_tuning(6, split_mode)
    try:
        dev = [an.analyze_track_file(f) for f in files]
        dev_album = an.analyze_album_files(files[:3])
        dev_peak = an.find_peak_amplitude_file(files[0])
    finally:
        an.set_tuning(6, DEFAULT_ROUTE)
    for f, h, d in zip(files, host, dev):
        assert (h.loudness_db, h.gain_db, h.peak, h.sample_rate, h.windows) == (d.loudness_db, d.gain_db, d.peak, d.sample_rate, d.windows), f.name
        pcm, _ = mp3dec.decode(f.read_bytes())
        want, _ = oracle.analyze_pcm(pcm[0], pcm[1] if pcm.shape[0] == 2 else None, h.sample_rate)
        assert (d.loudness_db, d.peak) == (want["loudness_db"], want["peak"]), f.name
    assert (host_album.album_loudness_db, host_album.album_peak) == (dev_album.album_loudness_db, dev_album.album_peak)
    assert host_peak.peak == dev_peak.peak


def test_configs0_one_30_second_mp3_through_analyze_track(_ctx, oracle, tmp_path):
    """BASELINE configs[0]: analyze() on one 30 s 44.1 kHz stereo MP3 file.  The file is the frames of a golden stream
    repeated to 30 s behind an ID3v2 tag; rg_analyze_track(path) decodes it itself on each of the three routes and the
    result is the oracle's on the host decoder's PCM, bin for bin; find_peak_amplitude agrees with the PCM's maximum."""
    an = _ctx
    body = (GOLD / "v1_44k_ms_mixed.mp3").read_bytes()
    one = mp3dec.scan(body)
    reps = int(30.0 * one.sample_rate / one.frames) + 1
    data = b"ID3\x03\x00\x00\x00\x00\x00\x0a" + bytes(10) + body * reps
    f = tmp_path / "thirty seconds.mp3"
    f.write_bytes(data)
    pcm, info = mp3dec.decode(data)
    assert info.sample_rate == 44100 and info.channels == 2 and 30.0 <= info.frames / 44100 < 30.3 and info.id3v2_bytes == 20
    want, _ = oracle.analyze_pcm(pcm[0], pcm[1], 44100)
    an.set_kernel(0)
    try:
        for route in (3, 2, 1, 0):
            an.set_tuning(6, route)
            got = an.analyze_track_file(f)
            assert (got.loudness_db, got.gain_db, got.peak, got.sample_rate) == (want["loudness_db"], want["gain_db"], want["peak"], 44100), route
            assert 0 < got.windows <= int(np.ceil(info.frames / 2205))  # silent windows are dropped, never invented
            pk = an.find_peak_amplitude_file(f)
            assert pk.peak == float(np.abs(pcm).max()) and pk.sample_rate == 44100
    finally:
        an.set_tuning(6, DEFAULT_ROUTE)


def test_album_mixing_wav_and_mp3_files(_ctx, oracle, tmp_path):
    """An album whose files are partly RIFF/WAVE (de-interleaved on the device) and partly MPEG Layer III (decoded on the
    device): one arena, one batch; the album is the oracle's merge of the per-file histograms, on either decode route."""
    sys.path.insert(0, str(Path(__file__).resolve().parent))
    from wavutil import test_signal, wav_bytes

    an = _ctx
    an.set_kernel(0)
    chans = test_signal("s16", 44100, 44100 * 2 + 321, 2, seed=3)
    wav = tmp_path / "a.wav"
    wav.write_bytes(wav_bytes(chans, 44100, "s16"))
    files = [wav, FIX / "test_vbr.mp3", GOLD / "v1_44k_ms_mixed.mp3", wav]
    per = []
    for f in files:
        if f.suffix == ".wav":
            per.append(oracle.analyze_pcm(np.asarray(chans[0], np.int16), np.asarray(chans[1], np.int16), 44100))
        else:
            pcm, _ = mp3dec.decode(f.read_bytes())
            per.append(oracle.analyze_pcm(pcm[0], pcm[1], 44100))
    want, _ = oracle.album_from_hists([h for _, h in per], [r["peak"] for r, _ in per])
    try:
        for route in (3, 2, 0):
            an.set_tuning(6, route)
            got = an.analyze_album_files(files)
            assert (got.album_loudness_db, got.album_gain_db, got.album_peak) == (want["album_loudness_db"], want["album_gain_db"], want["album_peak"])
            assert [t.loudness_db for t in got.tracks] == [r["loudness_db"] for r, _ in per]
    finally:
        an.set_tuning(6, DEFAULT_ROUTE)


def test_sharded_album_over_files_with_the_library_communicator(_ctx, oracle, tmp_path):
    """album.analyze_album_files_sharded: files dealt out by size, per-rank decode + analysis, agreement that nobody
    failed, RCCL exchange of the album pack, results gathered in input order.  One GPU hosts a 1-rank communicator: the
    collective is the identity, the call sequence (sync album call -> exchange -> album result) is the real one."""
    import torch.distributed as dist

    from mp3rgain_amd import album
    import mp3rgain_amd as rg

    an = _ctx
    an.set_kernel(0)
    files = [FIX / "test_vbr.mp3", GOLD / "v1_44k_ms_mixed.mp3", FIX / "test_mono.mp3", GOLD / "v1_44k_stereo_long.mp3"]
    plain = an.analyze_album_files(files)
    own_group = not dist.is_initialized()
    if own_group:
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29591", rank=0, world_size=1)
    try:
        an.comm_init_torch()
        got = album.analyze_album_files_sharded(an, files, exchange_even_if_alone=True)
        assert (got.album_loudness_db, got.album_gain_db, got.album_peak) == (plain.album_loudness_db, plain.album_gain_db, plain.album_peak)
        assert [t.loudness_db for t in got.tracks] == [t.loudness_db for t in plain.tracks]
        # a missing file ends the album on every rank before anyone enters the collective
        with pytest.raises(album.AlbumAborted, match="Failed to open"):
            album.analyze_album_files_sharded(an, files + [tmp_path / "missing.mp3"], exchange_even_if_alone=True)
        again = album.analyze_album_files_sharded(an, files, exchange_even_if_alone=True)  # and the context is fine
        assert again.album_loudness_db == plain.album_loudness_db
    finally:
        an.comm_destroy()
        if own_group:
            dist.destroy_process_group()


def test_analyze_tracks_batch_has_per_file_outcomes(_ctx, oracle, tmp_path):
    """rg_analyze_tracks: one GPU batch, per-file results equal to rg_analyze_track's, and per-file failures with the
    reference's texts that leave the other files alone."""
    import mp3rgain_amd as rg

    sys.path.insert(0, str(Path(__file__).resolve().parent))
    from wavutil import test_signal, wav_bytes

    an = _ctx
    an.set_kernel(0)
    odd = tmp_path / "odd_rate.wav"
    odd.write_bytes(wav_bytes(test_signal("s16", 44000, 5000, 2, seed=1), 44000, "s16"))
    ok_wav = tmp_path / "ok.wav"
    ok_wav.write_bytes(wav_bytes(test_signal("f32", 48000, 48000 * 2, 2, seed=2), 48000, "f32"))
    junk = tmp_path / "junk.mp3"
    junk.write_bytes(b"ID3" + bytes(500))
    empty = tmp_path / "empty.mp3"
    empty.write_bytes(b"")
    tiny = tmp_path / "tiny.mp3"
    tiny.write_bytes(b"\xff\xfb\x90")
    files = [FIX / "test_vbr.mp3", tmp_path / "missing.mp3", ok_wav, odd, GOLD / "v2_22k_stereo.mp3", junk, FIX / "test_mono.mp3", empty, tiny]
    got = an.analyze_track_files(files)
    assert len(got) == len(files)
    for f, g in zip(files, got):
        try:
            want = an.analyze_track_file(f)
        except rg.ReplayGainError as ex:
            assert isinstance(g, rg.ReplayGainError) and g.code == ex.code and str(g) == str(ex), f.name
            continue
        assert not isinstance(g, rg.ReplayGainError), (f.name, g)
        assert (g.loudness_db, g.gain_db, g.peak, g.sample_rate, g.windows, g.file_type) == (want.loudness_db, want.gain_db, want.peak, want.sample_rate, want.windows, want.file_type), f.name
    assert [isinstance(g, rg.ReplayGainError) for g in got] == [False, True, False, True, False, True, False, True, True]
    assert "Failed to probe format" in str(got[7]) and "Failed to probe format" in str(got[8])
    assert "Failed to open" in str(got[1]) and "Unsupported sample rate: 44000 Hz" in str(got[3]) and "Failed to probe format" in str(got[5])
    # track index > 0: every file reports it
    idx = an.analyze_track_files(files[:1], track_index=1)
    assert isinstance(idx[0], rg.ReplayGainError) and "Track index 1 out of range" in str(idx[0])
    assert an.analyze_track_files([]) == []


@pytest.mark.parametrize("src", ["v1_44k_ms_mixed.mp3", "v2_22k_stereo.mp3", "v1_44k_mono_crc_reservoir.mp3"])
def test_long_streams_carry_the_filterbank_state_across_groups(_ctx, src):
    """The fused kernel's blocks take runs of several groups of six granules when a chunk is large (the overlap and the
    filterbank history then stay in LDS from group to group) and rebuild that state from the two granules before a run that
    starts inside the track: a twelve-minute stream exercises both, and must still be the host decoder's PCM bit for bit."""
    body = (GOLD / src).read_bytes()
    one = mp3dec.scan(body)
    reps = int(12 * 60 * one.sample_rate / one.frames) + 1
    data = body * reps
    want, wi = mp3dec.decode(data)
    assert wi.frames / wi.sample_rate > 11 * 60
    for route in (3, 2):
        _ctx.set_tuning(6, route)
        try:
            got, gi = _ctx.decode_mp3_device(data)
        finally:
            _ctx.set_tuning(6, DEFAULT_ROUTE)
        assert (gi.frames, gi.audio_frames, gi.skipped_frames) == (wi.frames, wi.audio_frames, wi.skipped_frames)
        assert np.array_equal(got, want), (route, int(np.count_nonzero(got != want)))


def test_loader_pipeline_with_tiny_staging_blocks(oracle, tmp_path, monkeypatch):
    """The default route's pipeline (loader threads -> pinned staging blocks -> chunks -> device) with staging blocks of
    48 KB: 60 small files become dozens of chunks, so blocks are refilled while earlier chunks are still in flight, a file
    larger than a block makes its block grow, and the arena of a fresh context grows chunk by chunk with its contents
    kept.  Results must be those of the host-indexed route, file by file, and the album's."""
    import mp3rgain_amd as rg

    monkeypatch.setenv("RG_MP3_STAGE_BYTES", "49152")
    srcs = [p for p in STREAMS if p.stat().st_size < 60000]
    big = (GOLD / "v1_44k_ms_mixed.mp3").read_bytes()
    one = mp3dec.scan(big)
    long_file = tmp_path / "longer_than_a_block.mp3"
    long_file.write_bytes(big * (int(20 * one.sample_rate / one.frames) + 1))  # 20 s: several blocks' worth
    files = []
    for k in range(60):
        f = tmp_path / f"f{k:02d}.mp3"
        f.write_bytes(srcs[k % len(srcs)].read_bytes())
        files.append(f)
    files.insert(17, long_file)
    with rg.Analyzer(0) as an:  # fresh context: nothing is allocated yet
        an.set_kernel(0)
        got = an.analyze_album_files(files)
        again = an.analyze_album_files(files)
        an.set_tuning(6, 2)
        want = an.analyze_album_files(files)
    for a, b, c, f in zip(got.tracks, want.tracks, again.tracks, files):
        assert (a.loudness_db, a.gain_db, a.peak, a.sample_rate, a.windows) == (b.loudness_db, b.gain_db, b.peak, b.sample_rate, b.windows), f.name
        assert (c.loudness_db, c.peak, c.windows) == (b.loudness_db, b.peak, b.windows), f.name
    assert (got.album_loudness_db, got.album_gain_db, got.album_peak) == (want.album_loudness_db, want.album_gain_db, want.album_peak)
    assert (again.album_loudness_db, again.album_peak) == (want.album_loudness_db, want.album_peak)
    # and against the oracle on the host decoder's PCM for a few of them
    for f, a in list(zip(files, got.tracks))[:6] + [(long_file, got.tracks[17])]:
        pcm, info = mp3dec.decode(f.read_bytes())
        ref, _ = oracle.analyze_pcm(pcm[0], pcm[1] if info.channels == 2 else None, info.sample_rate)
        assert (a.loudness_db, a.peak) == (ref["loudness_db"], ref["peak"]), f.name


def test_album_parts_give_the_plain_route_results(tmp_path, monkeypatch):
    """Album parts (rg_files.hip: PartsRun): the tracks of a decoded chunk are analysed while later chunks are copied and decoded,
    and the album is the fold of the parts' packs.  With tiny staging blocks (dozens of chunks) and every chunk made a part, only
    copy-bound chunks (the default rule: the rest waits and joins a later part -- unless the device had to wait for it), and no
    parts at all, an album of 45 files --
    mono and stereo, 8 to 48 kHz, one file longer than a block -- must come out the same, field by field; a WAV file among
    them (not the pipeline's) sends the whole album down the plain route."""
    import mp3rgain_amd as rg
    import wave

    monkeypatch.setenv("RG_MP3_STAGE_BYTES", "65536")
    srcs = [p for p in STREAMS if p.stat().st_size < 60000]
    big = (GOLD / "v1_44k_ms_mixed.mp3").read_bytes()
    one = mp3dec.scan(big)
    files = []
    for k in range(44):
        f = tmp_path / f"p{k:02d}.mp3"
        f.write_bytes(srcs[(5 * k) % len(srcs)].read_bytes())
        files.append(f)
    long_file = tmp_path / "long.mp3"
    long_file.write_bytes(big * (int(25 * one.sample_rate / one.frames) + 1))
    files.insert(9, long_file)
    wav = tmp_path / "tone.wav"
    with wave.open(str(wav), "wb") as w:
        w.setnchannels(2); w.setsampwidth(2); w.setframerate(44100)
        t = np.arange(44100 * 2)
        x = (8000 * np.sin(2 * np.pi * 997 * t / 44100)).astype("<i2")
        w.writeframes(np.stack([x, x], axis=1).tobytes())

    def key(r):
        return [(t.loudness_db, t.gain_db, t.peak, t.sample_rate, t.windows, t.file_type) for t in r.tracks] + [(r.album_loudness_db, r.album_gain_db, r.album_peak)]

    with rg.Analyzer(0) as an:
        an.set_kernel(0)
        # (tuning keys 10 / 11: album parts never / on, and threshold + 1 of the copy-bound rule; the environment's
        # RG_ALBUM_PARTS / RG_PARTS_MIN_BYTES_PER_UNIT are only the defaults a context reads when it is created)
        an.set_tuning(10, 1)
        plain = key(an.analyze_album_files(files))
        plain_wav = key(an.analyze_album_files(files[:20] + [wav] + files[20:]))
        an.set_tuning(10, 2)
        for rule in (0, 120, 400, 10**9):
            an.set_tuning(11, rule + 1)
            assert key(an.analyze_album_files(files)) == plain, rule
            assert key(an.analyze_album_files(files)) == plain, rule  # (buffers in place now)
        an.set_tuning(11, 1)
        assert key(an.analyze_album_files(files[:20] + [wav] + files[20:])) == plain_wav
        assert key(an.analyze_album_files(files[:1])) == key(an.analyze_album_files(files[:1]))
        with pytest.raises(rg.ReplayGainError):
            an.analyze_album_files(files[:7] + [tmp_path / "missing.mp3"] + files[7:])
        assert key(an.analyze_album_files(files)) == plain  # and the context is fine afterwards
        # track mode (rg_analyze_tracks): the same parts without the album packs; a missing file is that file's error only
        def tkey(rs):
            return [(r.code, str(r)) if isinstance(r, rg.ReplayGainError) else (r.loudness_db, r.gain_db, r.peak, r.sample_rate, r.windows, r.file_type) for r in rs]
        with_missing = files[:7] + [tmp_path / "missing.mp3"] + files[7:]
        an.set_tuning(10, 1)
        t_plain, t_plain_missing = tkey(an.analyze_track_files(files)), tkey(an.analyze_track_files(with_missing))
        assert t_plain == [k for k in plain[:-1]]
        an.set_tuning(10, 2)
        for rule in (0, 120, 10**9):
            an.set_tuning(11, rule + 1)
            assert tkey(an.analyze_track_files(files)) == t_plain, rule
            assert tkey(an.analyze_track_files(with_missing)) == t_plain_missing, rule
        # A device that waits for the host (one loader thread and dozens of small chunks, none of them copy-bound by the rule):
        # the chunks it waited for become parts as well and the call's last chunks shrink (key 10 = 2, the default); key 10 = 3
        # keeps to the copy-bound rule, i.e. the plain route here.  Which chunks are which depends on timing; the results do not.
        an.set_tuning(7, 1)
        an.set_tuning(11, 10**9 + 1)
        for mode in (2, 3, 0):
            an.set_tuning(10, mode)
            for _ in range(3):
                assert key(an.analyze_album_files(files)) == plain, mode
            assert tkey(an.analyze_track_files(with_missing)) == t_plain_missing, mode
        an.set_tuning(7, 0)
        with pytest.raises(rg.ReplayGainError):
            an.set_tuning(10, 4)


def test_analyze_tracks_in_groups_bounded_by_memory(_ctx, tmp_path, monkeypatch):
    """`-r` over a whole library is taken in groups whose PCM fits the device (rg_analyze_tracks): with the group size forced
    down to a few files, results and per-file errors are those of the single batch."""
    import mp3rgain_amd as rg

    an = _ctx
    an.set_kernel(0)
    files = []
    srcs = [p for p in STREAMS if p.stat().st_size < 60000]
    for k in range(23):
        f = tmp_path / f"g{k:02d}.mp3"
        f.write_bytes(srcs[k % len(srcs)].read_bytes())
        files.append(f)
    files.insert(5, tmp_path / "nope.mp3")
    whole = an.analyze_track_files(files)
    an.set_tuning(13, 24 * 70000)  # two or three files per group (RG_TRACKS_GROUP_BYTES is the default a context reads at its creation)
    try:
        grouped = an.analyze_track_files(files)
    finally:
        an.set_tuning(13, 0)
    assert len(whole) == len(grouped) == len(files)
    for f, a, b in zip(files, whole, grouped):
        if isinstance(a, rg.ReplayGainError):
            assert isinstance(b, rg.ReplayGainError) and (a.code, str(a)) == (b.code, str(b)), f.name
        else:
            assert (a.loudness_db, a.gain_db, a.peak, a.sample_rate, a.windows) == (b.loudness_db, b.gain_db, b.peak, b.sample_rate, b.windows), f.name
    assert isinstance(whole[5], rg.ReplayGainError) and sum(isinstance(x, rg.ReplayGainError) for x in whole) == 1


def test_album_larger_than_the_device_is_analysed_in_parts(_ctx, oracle, tmp_path, monkeypatch):
    """rg_analyze_album on a file list whose PCM would not fit the device at once: the list is cut into parts, each part is
    an album enqueue of its own, and the parts' histograms and peaks are folded.  With the part size forced down to a few
    files the album and every track are those of the single-batch album (and of the oracle's merge)."""
    an = _ctx
    an.set_kernel(0)
    srcs = [p for p in STREAMS if p.stat().st_size < 60000]
    files = []
    for k in range(19):
        f = tmp_path / f"a{k:02d}.mp3"
        f.write_bytes(srcs[(3 * k) % len(srcs)].read_bytes())
        files.append(f)
    whole = an.analyze_album_files(files)
    an.set_tuning(13, 24 * 70000)
    try:
        parts = an.analyze_album_files(files)
    finally:
        an.set_tuning(13, 0)
    assert (parts.album_loudness_db, parts.album_gain_db, parts.album_peak) == (whole.album_loudness_db, whole.album_gain_db, whole.album_peak)
    for a, b in zip(whole.tracks, parts.tracks):
        assert (a.loudness_db, a.gain_db, a.peak, a.sample_rate, a.windows, a.file_type) == (b.loudness_db, b.gain_db, b.peak, b.sample_rate, b.windows, b.file_type)
    per = []
    for f in files:
        pcm, info = mp3dec.decode(f.read_bytes())
        per.append(oracle.analyze_pcm(pcm[0], pcm[1] if info.channels == 2 else None, info.sample_rate))
    want, _ = oracle.album_from_hists([h for _, h in per], [r["peak"] for r, _ in per])
    assert (parts.album_loudness_db, parts.album_gain_db, parts.album_peak) == (want["album_loudness_db"], want["album_gain_db"], want["album_peak"])


@pytest.mark.parametrize("case", sorted((ROOT / "tests" / "golden" / "mp3_cases").glob("*.mp3")), ids=lambda p: p.stem)
def test_a_frame_decodes_as_a_whole_or_not_at_all(_ctx, case):
    """Regression fixtures found by tools/fuzz_mp3_routes.py: a frame whose second granule's lengths do not add up.  The
    one-shot host decoder used to decode the first granule (moving the overlap and the filterbank history) before it
    dropped the frame; the device routes decide per frame before decoding anything.  Every route, and the host decoder,
    now produce the same PCM."""
    data = case.read_bytes()
    want, wi = mp3dec.decode(data)
    assert wi.skipped_frames >= 1
    for route in (3, 2, 1):
        _ctx.set_tuning(6, route)
        try:
            got, gi = _ctx.decode_mp3_device(data)
        finally:
            _ctx.set_tuning(6, DEFAULT_ROUTE)
        assert (gi.frames, gi.audio_frames, gi.skipped_frames) == (wi.frames, wi.audio_frames, wi.skipped_frames), route
        assert np.array_equal(got, want), route


def test_a_stream_without_a_whole_frame_decodes_to_nothing(_ctx):
    """63 bytes: a header whose frame is cut short.  The host decoder returns zero frames; so must the device routes (the
    frame parser never runs for such a stream: its count must not be whatever the buffer held before)."""
    data = (GOLD / "v1_44k_stereo_long.mp3").read_bytes()
    first = mp3dec.scan(data)
    cut = data[int(first.first_frame_offset):int(first.first_frame_offset) + 63]
    want, wi = mp3dec.decode(cut)
    assert wi.frames == 0
    for route in (3, 2, 1):
        _ctx.set_tuning(6, route)
        try:
            got, gi = _ctx.decode_mp3_device(cut)
        finally:
            _ctx.set_tuning(6, DEFAULT_ROUTE)
        assert gi.frames == 0 and got.shape[1] == 0


def test_the_decode_bench_hook_counts_what_it_decodes(_ctx):
    """rg_mp3_decode_bench (bench.py's mp3_end_to_end.roofline, tools/mp3_chain.py): units, PCM frames and compressed bytes are
    those of `copies` copies of the stream, every stage has a duration and the chain is at least their sum's largest part.
    The PCM it leaves in the arena is not read back here: test_device_half_reproduces_the_host_decoder holds the kernels."""
    data = (GOLD / "dense_44k_joint_128.mp3").read_bytes()
    si = mp3dec.scan(data)
    copies = 6
    r = _ctx.decode_mp3_bench(data, copies, reps=3)
    assert r["units"] == si.audio_frames * 2 * si.channels * copies
    assert r["frames"] == si.frames * copies
    assert 0.8 * len(data) * copies < r["compressed_bytes"] < 1.1 * len(data) * copies  # main data + a 40-byte slot per frame
    ms = r["ms"]
    assert set(ms) == {"frames", "huffman", "backhalf", "chain", "chain_pipelined"}
    assert 0.0 < ms["chain_pipelined"] <= 1.25 * ms["chain"] + 0.05
    assert all(v > 0.0 for v in ms.values())
    assert ms["chain"] >= max(ms["frames"], ms["huffman"], ms["backhalf"])
    assert ms["chain"] <= 1.5 * (ms["frames"] + ms["huffman"] + ms["backhalf"]) + 0.05
