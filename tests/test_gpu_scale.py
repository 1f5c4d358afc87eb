"""GPU tests at BASELINE.json's sizes: direct comparison where the oracle finishes in seconds, and
size-independent properties (album histogram == sum of track histograms, window conservation,
idempotence across pipeline slots) for the large batches."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RATE = 44100
W = 2205


def _device_batch(analyzer, seeds, frames_list, rate=RATE, channels=2):
    import torch

    from mp3rgain_amd import _capi

    total = sum(f * channels for f in frames_list)
    buf = torch.empty(total + 4, dtype=torch.float32, device="cuda:0")
    descs = (_capi.TrackDesc * len(seeds))()
    off = 0
    for t, (s, f) in enumerate(zip(seeds, frames_list)):
        for c in range(channels):
            analyzer.synth_fill_device(buf.data_ptr() + 4 * (off + c * f), s, c, rate, 0, f)
        descs[t].offset_bytes = 4 * off
        descs[t].frames = f
        descs[t].sample_rate = rate
        descs[t].channels = channels
        descs[t].format = _capi.FMT_F32_PLANAR
        off += channels * f
    return buf, descs


def test_config2_ten_minute_track_full_size(analyzer, oracle):
    """BASELINE configs[1] at full size (26 460 000 frames): every one of the 12 000 windows in the oracle's bin."""
    frames = 600 * RATE
    buf, descs = _device_batch(analyzer, [0x5EED0000], [frames])
    analyzer.enqueue_device(descs, 1, buf.data_ptr(), buf.numel() * 4)
    got, h = analyzer.collect(1, want_hist=True)
    l, r = oracle.synth_f32(0x5EED0000, 0, RATE, frames), oracle.synth_f32(0x5EED0000, 1, RATE, frames)
    want, wh = oracle.analyze_pcm(l, r, RATE)
    diff = np.nonzero(h[0] != wh)[0]
    assert diff.size == 0, f"{diff.size} bins differ, first {diff[:5]}"
    assert got[0].loudness_db == want["loudness_db"] and got[0].peak == want["peak"]
    assert abs(got[0].loudness_db - want["loudness_db"]) <= 0.1  # the north_star tolerance, for the record
    # the 1 s digital-silence gap: its windows are dropped, not clamped (replaygain.rs:757); the first one still
    # carries the filter's ringing and counts
    assert got[0].windows == int(wh.sum()) == 12000 - 19


def test_config3_shape_batch_properties(analyzer, oracle):
    """A slice of BASELINE configs[2]/[3] (3-minute tracks): album histogram is the sum of the track histograms,
    every track conserves its windows, per-track gains differ, and a sample of tracks matches the oracle."""
    n, frames = 24, 180 * RATE
    seeds = [0x5EED1000 + t for t in range(n)]
    buf, descs = _device_batch(analyzer, seeds, [frames] * n)
    analyzer.enqueue_device(descs, n, buf.data_ptr(), buf.numel() * 4, album=True)
    got, h = analyzer.collect(n, want_hist=True)
    alb, ah = analyzer.album_finish(want_hist=True)
    assert np.array_equal(ah, h.sum(axis=0, dtype=np.uint64).astype(np.uint32))  # accumulate (replaygain.rs:658-662)
    assert alb.album_peak == max(g.peak for g in got)
    assert alb.album_loudness_db == oracle.hist_loudness(ah)
    for g, hh in zip(got, h):
        assert g.windows == int(hh.sum()) and 3600 - 21 <= g.windows <= 3600 - 18  # the silent second is dropped
        assert g.loudness_db == oracle.hist_loudness(hh)
    assert len({g.gain_db for g in got}) > 4  # per-track levels differ
    for t in (0, 7, 23):
        l, r = oracle.synth_f32(seeds[t], 0, RATE, frames), oracle.synth_f32(seeds[t], 1, RATE, frames)
        want, wh = oracle.analyze_pcm(l, r, RATE)
        assert np.array_equal(h[t], wh) and got[t].peak == want["peak"]


@pytest.mark.parametrize("rate,n,seconds", [(96000, 20, 150), (64000, 72, 170)])
def test_high_rate_batches_large_enough_for_the_long_segments(analyzer, oracle, rate, n, seconds):
    """Batches at 96 / 64 kHz large enough that the segment chooser takes what it takes for production-size batches there --
    the longest segment whose response tables still fit the LDS (96 kHz: L = 2400, one wave per SIMD) and several windows per
    lane (64 kHz) -- against the oracle on a sample of tracks, and the album as the sum of its tracks."""
    frames = seconds * rate + 977  # not a whole number of windows
    seeds = [0x5EED9000 + 16 * (rate // 1000) + t for t in range(n)]
    buf, descs = _device_batch(analyzer, seeds, [frames] * n, rate=rate)
    analyzer.enqueue_device(descs, n, buf.data_ptr(), buf.numel() * 4, album=True)
    got, h = analyzer.collect(n, want_hist=True)
    alb, ah = analyzer.album_finish(want_hist=True)
    assert np.array_equal(ah, h.sum(axis=0, dtype=np.uint64).astype(np.uint32))
    assert alb.album_loudness_db == oracle.hist_loudness(ah)
    for t in (0, n // 2, n - 1):
        l, r = oracle.synth_f32(seeds[t], 0, rate, frames), oracle.synth_f32(seeds[t], 1, rate, frames)
        want, wh = oracle.analyze_pcm(l, r, rate)
        diff = np.nonzero(h[t] != wh)[0]
        assert diff.size == 0, f"track {t}: {diff.size} bins differ, first {diff[:5]}"
        assert got[t].loudness_db == want["loudness_db"] and got[t].peak == want["peak"]


def test_pipelined_enqueues_are_idempotent(analyzer, oracle):
    """Back-to-back enqueues rotate through the pipeline slots and overlap on the GPU; every one of them
    must produce the same bits (no cross-slot interference), including the in-kernel clearing of accumulators."""
    lens = [RATE * 20 + 31, RATE * 7, 2205 * 13, 5000]
    seeds = [0x5EED2000 + t for t in range(len(lens))]
    buf, descs = _device_batch(analyzer, seeds, lens)
    ref = None
    for rep in range(9):
        analyzer.enqueue_device(descs, len(lens), buf.data_ptr(), buf.numel() * 4, album=True)
        if rep % 2 == 0:
            got, h = analyzer.collect(len(lens), want_hist=True)
            alb, ah = analyzer.album_finish(want_hist=True)
            cur = (h.copy(), ah.copy(), [(g.loudness_db, g.peak, g.windows) for g in got], alb.album_loudness_db)
            if ref is None:
                ref = cur
                for t, (s, f) in enumerate(zip(seeds, lens)):
                    l, r = oracle.synth_f32(s, 0, RATE, f), oracle.synth_f32(s, 1, RATE, f)
                    _, wh = oracle.analyze_pcm(l, r, RATE)
                    assert np.array_equal(h[t], wh)
            else:
                assert np.array_equal(cur[0], ref[0]) and np.array_equal(cur[1], ref[1])
                assert cur[2] == ref[2] and cur[3] == ref[3]


def test_mixed_48k_mono_and_hot_tracks(analyzer, oracle):
    """BASELINE configs[4] ingredients: 44.1/48 kHz mixed, mono tracks, full-scale ("hot") tracks whose peak
    reaches 1.0, and the -k rule applied to the GPU's numbers (src/main.rs:2033-2058)."""
    import mp3rgain_amd as rg

    specs = [(44100, 2, 0x5EED3000), (48000, 2, 0x5EED3001), (48000, 1, 0x5EED3002), (44100, 2, 0x5EED3003 | (1 << 40)),
             (48000, 2, 0x5EED3004 | (1 << 40)), (44100, 1, 0x5EED3005)]
    tracks, wants = [], []
    for rate, ch, seed in specs:
        n = rate * 6 + 99
        chans = [oracle.synth_f32(seed, c, rate, n) for c in range(ch)]
        tracks.append(rg.PcmTrack(chans, rate))
        wants.append(oracle.analyze_pcm(chans[0], chans[1] if ch == 2 else None, rate))
    got, h = analyzer.analyze_tracks(tracks, return_histograms=True)
    L = oracle.lib()
    for g, hh, (w, wh) in zip(got, h, wants):
        assert np.array_equal(hh, wh) and g.peak == w["peak"] and g.loudness_db == w["loudness_db"]
        for k in (0, 1):
            assert rg.replaygain.clip_limit_steps(g.gain_steps(), g.gain_db, g.peak, bool(k)) == \
                L.rgo_clip_limit_steps(w["gain_steps"], w["gain_db"], w["peak"], k, 0)
    assert got[3].peak == 1.0 and got[4].peak == 1.0


def test_attached_stream_orders_the_album_tail(analyzer, oracle):
    """rg_set_stream(default stream): work the caller submits to its stream right after an album enqueue (here a
    torch copy of the [histogram | peak] pack, in production the RCCL collective) sees the merged histogram, and
    the fold + album percentile that follow on that stream see the caller's result."""
    import torch

    from mp3rgain_amd import album

    lens = [RATE * 15, RATE * 9 + 7]
    seeds = [0x5EED4000, 0x5EED4001]
    buf, descs = _device_batch(analyzer, seeds, lens)
    analyzer.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        want_h = np.zeros(12000, dtype=np.uint32)
        peaks = []
        for s, f in zip(seeds, lens):
            r, h = oracle.analyze_pcm(oracle.synth_f32(s, 0, RATE, f), oracle.synth_f32(s, 1, RATE, f), RATE)
            want_h += h
            peaks.append(r["peak"])

        class _View:
            def __init__(self, ptr, n):
                self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i4", "data": (ptr, False), "version": 3}

        for rep in range(6):  # every pipeline slot at least once
            analyzer.enqueue_device(descs, 2, buf.data_ptr(), buf.numel() * 4, album=True)
            view = analyzer.device_view()
            pack = torch.as_tensor(_View(view.d_album_hist, album.ALBUM_PACK_WORDS), device="cuda")
            gathered = torch.cat([pack, pack])  # stands in for the all-gather of a 2-rank job, on the caller's stream
            analyzer.album_reduce_gathered(gathered.data_ptr(), 2)
            analyzer.album_result_enqueue()
            alb, ah = analyzer.album_finish(want_hist=True)
            assert np.array_equal(ah, 2 * want_h)  # two identical "ranks"
            assert alb.album_peak == max(peaks) and alb.album_loudness_db == oracle.hist_loudness(2 * want_h)
    finally:
        analyzer.set_stream(None)


def test_fold_of_two_different_ranks(analyzer, oracle):
    """rg_album_reduce_gathered with a genuinely different second pack: this context's album pack next to the pack a
    second rank would have sent (its tracks analysed by the oracle), folded on the device: bins add, peaks take the
    maximum, the album percentile reads the sum -- analyze_album's merge (src/replaygain.rs:1056-1066) across ranks."""
    import torch

    from mp3rgain_amd import album

    lens = [RATE * 11, RATE * 4 + 100]
    seeds = [0x5EED6000, 0x5EED6001]
    buf, descs = _device_batch(analyzer, seeds, lens)
    mine_h = np.zeros(12000, dtype=np.uint32)
    peaks = []
    for s, f in zip(seeds, lens):
        r, h = oracle.analyze_pcm(oracle.synth_f32(s, 0, RATE, f), oracle.synth_f32(s, 1, RATE, f), RATE)
        mine_h += h
        peaks.append(r["peak"])
    # the other rank: three other tracks, one of them hot (peak 1.0) and one at another level
    other_h = np.zeros(12000, dtype=np.uint32)
    other_peak = 0.0
    for s, f in ((0x5EED7000 | (1 << 40), RATE * 6), (0x5EED7001, RATE * 9 + 1), (0x5EED7002, RATE * 2)):
        r, h = oracle.analyze_pcm(oracle.synth_f32(s, 0, RATE, f), oracle.synth_f32(s, 1, RATE, f), RATE)
        other_h += h
        other_peak = max(other_peak, r["peak"])
    assert other_peak > max(peaks) and not np.array_equal(other_h, mine_h)
    pack2 = np.zeros(album.ALBUM_PACK_WORDS, dtype=np.uint32)
    pack2[:12000] = other_h
    pack2[12000:] = np.array([other_peak], dtype=np.float64).view(np.uint32)
    pack2_t = torch.from_numpy(pack2.view(np.int32).copy()).cuda()

    class _View:
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i4", "data": (ptr, False), "version": 3}

    analyzer.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        for order in (0, 1):  # this rank's pack first, then second: the fold does not care who is rank 0
            analyzer.enqueue_device(descs, 2, buf.data_ptr(), buf.numel() * 4, album=True)
            view = analyzer.device_view()
            pack = torch.as_tensor(_View(view.d_album_hist, album.ALBUM_PACK_WORDS), device="cuda")
            gathered = torch.cat([pack, pack2_t] if order == 0 else [pack2_t, pack])
            analyzer.album_reduce_gathered(gathered.data_ptr(), 2)
            analyzer.album_result_enqueue()
            alb, ah = analyzer.album_finish(want_hist=True)
            assert np.array_equal(ah, mine_h + other_h)
            assert alb.album_peak == other_peak
            assert alb.album_loudness_db == oracle.hist_loudness(mine_h + other_h)
            assert alb.windows == int(mine_h.sum()) + int(other_h.sum())
            # host restatement of the fold agrees
            fh, fp = album.fold_gathered(gathered.cpu().numpy().view(np.uint32), 2)
            assert np.array_equal(fh, ah) and fp == other_peak
    finally:
        analyzer.set_stream(None)


def test_library_communicator_exchange_on_the_batch_stream(analyzer, oracle):
    """rg_comm_* + rg_album_exchange: the album exchange as one RCCL all-gather + device fold on the stream of the
    batch (what bench.py runs at N > 1).  One GPU can host only a 1-rank communicator, so the collective is the
    identity here; the bootstrap, the call sequence and the stream ordering are the real ones."""
    import torch.distributed as dist

    lens = [RATE * 12, RATE * 7 + 3, RATE * 3]
    seeds = [0x5EED5000, 0x5EED5001, 0x5EED5002]
    buf, descs = _device_batch(analyzer, seeds, lens)
    want_h = np.zeros(12000, dtype=np.uint32)
    peaks = []
    for s, f in zip(seeds, lens):
        r, h = oracle.analyze_pcm(oracle.synth_f32(s, 0, RATE, f), oracle.synth_f32(s, 1, RATE, f), RATE)
        want_h += h
        peaks.append(r["peak"])
    own_group = not dist.is_initialized()
    if own_group:
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29577", rank=0, world_size=1)
    try:
        analyzer.comm_init_torch()
        for rep in range(10):  # every pipeline slot at least once
            analyzer.enqueue_device(descs, 3, buf.data_ptr(), buf.numel() * 4, album=True)
            analyzer.album_exchange()
            analyzer.album_result_enqueue()
            if rep % 3 == 0:
                alb, ah = analyzer.album_finish(want_hist=True)
                assert np.array_equal(ah, want_h)
                assert alb.album_peak == max(peaks) and alb.album_loudness_db == oracle.hist_loudness(want_h)
        res = analyzer.collect(3)
        assert [r.peak for r in res] == peaks
    finally:
        analyzer.comm_destroy()
        if own_group:
            dist.destroy_process_group()
    # without a communicator the exchange is a no-op (single GPU)
    analyzer.enqueue_device(descs, 3, buf.data_ptr(), buf.numel() * 4, album=True)
    analyzer.album_exchange()
    analyzer.album_result_enqueue()
    alb, ah = analyzer.album_finish(want_hist=True)
    assert np.array_equal(ah, want_h)


def test_finisher_handoff_under_uneven_load(_ctx):
    """The fix-up kernel publishes through device-scope atomics without a release fence, and the last block of a
    track to arrive reads the histogram after one acquire.  Stale reads, if the ordering were
    wrong, show up under uneven load with warm caches: 240 tracks from 0.3 s to 90 s in one batch, re-enqueued 60
    times across all pipeline slots while other batches are in flight; every result of every repetition must
    equal the first one bit for bit, and the first one must equal the order-faithful variant-1 kernel's bins."""
    an = _ctx
    an.set_kernel(2)
    an.set_tuning(1, 0)
    an.set_tuning(2, 0)
    an.set_tuning(3, 0)
    rng = np.random.default_rng(424242)
    lens = [int(x) for x in np.concatenate([rng.integers(RATE // 3, RATE * 3, 150), rng.integers(RATE * 20, RATE * 90, 30),
                                            rng.integers(1, 5000, 60)])]
    rng.shuffle(lens)
    seeds = [0x5EED7000 + t for t in range(len(lens))]
    buf, descs = _device_batch(an, seeds, lens)
    n = len(lens)
    small, sdescs = _device_batch(an, [1, 2, 3], [RATE * 5, RATE, 777])

    def run():
        an.enqueue_device(descs, n, buf.data_ptr(), buf.numel() * 4, album=True)
        got, h = an.collect(n, want_hist=True)
        alb, ah = an.album_finish(want_hist=True)
        return h.copy(), ah.copy(), [(g.loudness_db, g.gain_db, g.peak, g.windows) for g in got], (alb.album_loudness_db, alb.album_peak)

    ref = run()
    assert np.array_equal(ref[1], ref[0].sum(axis=0, dtype=np.uint64).astype(np.uint32))  # album == sum of tracks
    assert [w for _, _, _, w in ref[2]] == [int(x) for x in ref[0].sum(axis=1)]
    import os

    for rep in range(int(os.environ.get("RG_STRESS_REPS", "60"))):
        for _ in range(rep % 4):  # other batches in flight on the neighbouring pipeline streams
            an.enqueue_device(sdescs, 3, small.data_ptr(), small.numel() * 4)
        cur = run()
        assert np.array_equal(cur[0], ref[0]), f"repetition {rep}: track histograms differ"
        assert np.array_equal(cur[1], ref[1]) and cur[2] == ref[2] and cur[3] == ref[3], f"repetition {rep}"
    an.set_kernel(1)
    an.enqueue_device(descs, n, buf.data_ptr(), buf.numel() * 4, album=True)
    got1, h1 = an.collect(n, want_hist=True)
    an.set_kernel(0)
    assert np.array_equal(h1, ref[0])
    assert [(g.loudness_db, g.peak) for g in got1] == [(l, p) for l, _, p, _ in ref[2]]


def test_config3_full_size_1000_tracks(_ctx, oracle):
    """BASELINE configs[2] at its full size: 1000 tracks x 3 min x 44.1 kHz stereo = 7.9e9 stereo frames, 63.5 GB of
    f32 PCM resident in HBM, one batch.  Size-independent properties over all of it, the oracle on a sample."""
    import torch

    an = _ctx
    an.set_kernel(0)
    for key in (1, 2, 3):
        an.set_tuning(key, 0)
    free, _ = torch.cuda.mem_get_info()
    n, frames = 1000, 180 * RATE
    if free < (n * 2 * frames * 4) * 1.15:
        pytest.skip("not enough free HBM for the full-size batch")
    seeds = [0x5EED0000 + t for t in range(n)]
    buf, descs = _device_batch(an, seeds, [frames] * n)
    an.enqueue_device(descs, n, buf.data_ptr(), buf.numel() * 4, album=True)
    got, h = an.collect(n, want_hist=True)
    alb, ah = an.album_finish(want_hist=True)
    # LoudnessHistogram::accumulate over all tracks, album_peak.max (replaygain.rs:658-662, 1056-1059)
    assert np.array_equal(ah, h.sum(axis=0, dtype=np.uint64).astype(np.uint32))
    assert alb.album_peak == max(g.peak for g in got)
    assert alb.album_loudness_db == oracle.hist_loudness(ah) and alb.album_gain_db == 64.82 - alb.album_loudness_db
    wins = h.sum(axis=1)
    assert [g.windows for g in got] == [int(w) for w in wins]
    assert wins.min() >= 3600 - 21 and wins.max() <= 3600 - 18  # the silent second of every track is dropped
    for t in range(0, n, 37):
        assert got[t].loudness_db == oracle.hist_loudness(h[t])
    assert len({g.gain_steps() for g in got}) >= 5  # per-track levels differ
    for t in (0, 499, 999):
        l, r = oracle.synth_f32(seeds[t], 0, RATE, frames), oracle.synth_f32(seeds[t], 1, RATE, frames)
        want, wh = oracle.analyze_pcm(l, r, RATE)
        assert np.array_equal(h[t], wh) and got[t].peak == want["peak"] and got[t].loudness_db == want["loudness_db"]
    # the same batch again (another pipeline slot): the same bits
    an.enqueue_device(descs, n, buf.data_ptr(), buf.numel() * 4, album=True)
    got2, h2 = an.collect(n, want_hist=True)
    assert np.array_equal(h2, h) and [g.peak for g in got2] == [g.peak for g in got]
    del buf
    torch.cuda.empty_cache()


def test_config5_full_size_mixed_batch(_ctx, oracle):
    """BASELINE configs[4] at its full size: 500 tracks at 44.1 kHz + 500 at 48 kHz, 3 minutes each, 10 % mono, 5 % with
    full-scale peaks, interleaved in one batch (four launch groups).  Properties over all of it, oracle on one of each."""
    import torch

    from mp3rgain_amd import _capi

    an = _ctx
    an.set_kernel(0)
    for key in (1, 2, 3):
        an.set_tuning(key, 0)
    n = 1000
    rates = [44100 if t % 2 == 0 else 48000 for t in range(n)]
    chans = [1 if t % 10 == 3 else 2 for t in range(n)]
    seeds = [(0x5EED8000 + t) | ((1 << 40) if t % 20 == 7 else 0) for t in range(n)]  # bit 40: "hot" track
    frames = [180 * r for r in rates]
    total = sum(f * c for f, c in zip(frames, chans))
    free, _ = torch.cuda.mem_get_info()
    if free < total * 4 * 1.15:
        pytest.skip("not enough free HBM for the full-size batch")
    buf = torch.empty(total + 4, dtype=torch.float32, device="cuda:0")
    descs = (_capi.TrackDesc * n)()
    off = 0
    for t in range(n):
        for c in range(chans[t]):
            an.synth_fill_device(buf.data_ptr() + 4 * (off + c * frames[t]), seeds[t], c, rates[t], 0, frames[t])
        descs[t].offset_bytes, descs[t].frames, descs[t].sample_rate = 4 * off, frames[t], rates[t]
        descs[t].channels, descs[t].format = chans[t], _capi.FMT_F32_PLANAR
        off += chans[t] * frames[t]
    an.enqueue_device(descs, n, buf.data_ptr(), buf.numel() * 4, album=True)
    got, h = an.collect(n, want_hist=True)
    alb, ah = an.album_finish(want_hist=True)
    assert np.array_equal(ah, h.sum(axis=0, dtype=np.uint64).astype(np.uint32))
    assert alb.album_peak == max(g.peak for g in got) == 1.0
    assert [g.sample_rate for g in got] == rates  # results come back in input order (replaygain.rs:1061)
    assert [g.windows for g in got] == [int(w) for w in h.sum(axis=1)]
    hot = [t for t in range(n) if seeds[t] >> 40]
    assert len(hot) == 50 and all(got[t].peak == 1.0 for t in hot)
    assert sum(1 for t in range(n) if got[t].peak < 1.0) >= 900
    for t in (0, 1, 3, 13, 7, 27, 999):  # 44.1k stereo, 48k stereo, 48k mono, 48k mono, 48k hot, 48k hot, last
        ch = [oracle.synth_f32(seeds[t], c, rates[t], frames[t]) for c in range(chans[t])]
        want, wh = oracle.analyze_pcm(ch[0], ch[1] if chans[t] == 2 else None, rates[t])
        assert np.array_equal(h[t], wh), f"track {t}"
        assert got[t].peak == want["peak"] and got[t].loudness_db == want["loudness_db"] and got[t].gain_steps() == want["gain_steps"]
    del buf
    torch.cuda.empty_cache()


def test_streamed_host_ingest_gives_the_one_shot_bits(_ctx, oracle):
    """rg_analyze_pcm_batch / rg_analyze_album_pcm on a host arena larger than the ingest chunk (tuning key 5): the
    batch is cut at track boundaries, two device arenas take turns, copies run under the previous sub-batch's kernels.
    Same bits as the one-shot path -- per-track results in input order, histograms, album -- including a track that
    variant 2 flags (exact repeat inside its sub-batch), mixed rates, formats, a mono and an empty track."""
    import mp3rgain_amd as rg

    an = _ctx
    an.set_kernel(0)
    for key in (1, 2, 3, 4, 5):
        an.set_tuning(key, 0)
    rng = np.random.default_rng(77)
    tracks = []
    for t in range(14):
        rate = [44100, 48000, 32000, 22050][t % 4]
        n = int(rate * (0.7 + 1.9 * rng.random()))
        ch = [oracle.synth_f32(0x5EED9000 + t, c, rate, n) for c in range(1 if t == 5 else 2)]
        if t % 5 == 2:
            ch = [np.round(c * 32767).astype(np.int16) for c in ch]
        tracks.append(rg.PcmTrack(ch, rate))
    dc = [(s * 0.9 + 3e-5 * rng.standard_normal(44100 * 2)).astype(np.float32) for s in (1.0, -1.0)]
    tracks.insert(6, rg.PcmTrack(dc, 44100))                      # every window cancels: flagged by variant 2
    tracks.insert(9, rg.PcmTrack([np.zeros(0, np.float32)] * 2, 44100))  # empty track
    one, one_h = an.analyze_tracks(tracks, return_histograms=True)
    alb1, alb1_h = an.analyze_album(tracks, return_histogram=True)
    total = sum(sum(c.nbytes for c in t.channels) for t in tracks)
    for chunk_kib in (64, 300, 1024):
        assert total > chunk_kib * 1024 * 2
        an.set_tuning(5, chunk_kib)
        try:
            got, got_h = an.analyze_tracks(tracks, return_histograms=True)
            alb, alb_h = an.analyze_album(tracks, return_histogram=True)
        finally:
            an.set_tuning(5, 0)
        assert np.array_equal(got_h, one_h)
        for a, b in zip(got, one):
            assert (a.loudness_db, a.gain_db, a.peak, a.sample_rate, a.windows, a.flags) == (b.loudness_db, b.gain_db, b.peak, b.sample_rate, b.windows, b.flags)
        assert np.array_equal(alb_h, alb1_h)
        assert (alb.album_loudness_db, alb.album_gain_db, alb.album_peak) == (alb1.album_loudness_db, alb1.album_gain_db, alb1.album_peak)
        assert [r.loudness_db for r in alb.tracks] == [r.loudness_db for r in alb1.tracks]
    # and the bits are the oracle's
    for tr, r, h in zip(tracks, one, one_h):
        want, wh = oracle.analyze_pcm(tr.channels[0], tr.channels[1] if len(tr.channels) > 1 else None, tr.sample_rate)
        assert np.array_equal(h, wh) and r.loudness_db == want["loudness_db"] and r.peak == want["peak"] and not r.flags & 2
