"""CPU tests that pin the oracle (oracle/rg_oracle.c) before it is trusted as the parity checker.

Sources of truth, strongest first:
  1. the reference's own tests for this path (src/replaygain.rs:1275-1365): supported rates and two
     loudness ranges -- restated below with the same signals and the same assertions;
  2. an independent pure-Python transcription of the reference's per-sample code (bit-exact match
     required: Python floats are IEEE f64 and never fuse multiply-add);
  3. an independent formulation (scipy.signal.lfilter) that must land every window in the same bin;
  4. known answers derived in SURVEY.md Appendix B (72.97 dB / 58.99 dB, percentile thresholds);
  5. committed golden vectors (tests/golden/oracle_vectors.json, made by tools/make_golden.py).
"""
import hashlib
import json
import math
import re
import struct
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
REF = Path("/root/reference/src/replaygain.rs")
RATES = [96000, 88200, 64000, 48000, 44100, 32000, 24000, 22050, 16000, 12000, 11025, 8000]


def _header_table():
    """Parse include/rg_coeffs.h -> {rate: (ya, yb, ba, bb)}."""
    txt = (ROOT / "include" / "rg_coeffs.h").read_text()
    rows = re.findall(r"\{ (\d+)u,\s*\{([^}]*)\},\s*\{([^}]*)\},\s*\{([^}]*)\},\s*\{([^}]*)\},\s*\}", txt)
    out = {}
    for rate, *arrs in rows:
        out[int(rate)] = tuple([float(v) for v in a.split(",")] for a in arrs)
    return out


def test_coefficient_header_digest():
    tab = _header_table()
    assert list(tab) == RATES
    h = hashlib.sha256()
    for r in RATES:
        h.update(struct.pack("<I", r))
        for arr in tab[r]:
            for v in arr:
                h.update(struct.pack("<d", v))
    assert h.hexdigest() == (ROOT / "tests" / "golden" / "coeffs_sha256.txt").read_text().strip()
    for r in RATES:
        ya, yb, ba, bb = tab[r]
        assert (len(ya), len(yb), len(ba), len(bb)) == (11, 11, 3, 3)  # replaygain.rs:1284-1285
        assert ya[0] == 1.0 and ba[0] == 1.0


@pytest.mark.skipif(not REF.exists(), reason="reference tree only exists in the build container")
def test_coefficient_header_matches_reference_source():
    src = REF.read_text()
    tab = _header_table()
    pat = re.compile(r"pub const (YULE|BUTTER)_([AB])_(\d+): \[f64; \d+\] = \[(.*?)\];", re.S)
    seen = 0
    for kind, ab, rate, body in pat.findall(src):
        vals = [float(v) for v in body.replace("\n", " ").split(",") if v.strip()]
        ya, yb, ba, bb = tab[int(rate)]
        mine = {("YULE", "A"): ya, ("YULE", "B"): yb, ("BUTTER", "A"): ba, ("BUTTER", "B"): bb}[(kind, ab)]
        assert mine == vals, (kind, ab, rate)
        seen += 1
    assert seen == 48


def test_reference_test_filter_creation(oracle):
    """src/replaygain.rs:1275-1294."""
    for r in RATES:
        assert oracle.supported_rate(r), r
    assert not oracle.supported_rate(99999)
    assert not oracle.supported_rate(0)


def test_reference_test_rms_and_loudness_ranges(oracle):
    """src/replaygain.rs:1296-1365: same signals (f64 sine straight into filter.process), same asserts,
    plus the known answers of SURVEY.md Appendix B."""
    loud, hist = oracle.unit_test_sine(44100, 1000.0, 0.5, 44100)
    assert 50.0 < loud < 100.0
    assert loud == 72.97 and hist[9297] == 20 and hist.sum() == 20
    loud, hist = oracle.unit_test_sine(44100, 1000.0, 0.1, 44100)
    assert 50.0 < loud < 80.0
    assert loud == 58.99 and hist[7899] == 20 and hist.sum() == 20


def test_percentile_threshold_table(oracle):
    """ceil(total * (1.0 - 0.95)) with 1.0-0.95 == 0.050000000000000044 (replaygain.rs:671)."""
    table = {1: 1, 19: 1, 20: 2, 21: 2, 40: 3, 100: 6, 3600: 181, 12000: 601, 12001: 601}
    for total, thr in table.items():
        assert oracle.lib().rgo_percentile_threshold(total) == thr
        assert math.ceil(total * (1.0 - 0.95)) == thr


def test_hist_loudness_rules(oracle):
    h = np.zeros(12000, dtype=np.uint32)
    assert oracle.hist_loudness(h) == -20.0  # empty (replaygain.rs:667-669)
    h[0] = 1
    assert oracle.hist_loudness(h) == -20.0  # bin 0 -> (0-2000)/100
    h[:] = 0
    h[11999] = 1
    assert oracle.hist_loudness(h) == 99.99
    h[:] = 0
    h[5000:5020] = 1  # 20 windows, threshold 2 -> second from the top
    assert oracle.hist_loudness(h) == (5018 - 2000) / 100.0


# ---- independent pure-Python transcription of the reference's per-sample code ---------------------
class _PyFilter:
    """EqualLoudnessFilter (replaygain.rs:534-616), transcribed independently of oracle/rg_oracle.c."""

    def __init__(self, ya, yb, ba, bb):
        self.ya, self.yb, self.ba, self.bb = ya, yb, ba, bb
        self.yx, self.yy, self.bx, self.by = [0.0] * 11, [0.0] * 11, [0.0] * 3, [0.0] * 3

    def process(self, s):
        self.yx[1:11] = self.yx[0:10]
        self.yy[1:11] = self.yy[0:10]
        self.yx[0] = s
        acc = 0.0
        for i in range(1, 11):
            acc = acc + (self.yb[i] * self.yx[i] - self.ya[i] * self.yy[i])
        y = 1e-10 + self.yb[0] * self.yx[0] + acc
        self.yy[0] = y
        self.bx[1:3] = self.bx[0:2]
        self.by[1:3] = self.by[0:2]
        self.bx[0] = y
        acc = 0.0
        for i in range(1, 3):
            acc = acc + (self.bb[i] * self.bx[i] - self.ba[i] * self.by[i])
        z = 1e-10 + self.bb[0] * self.bx[0] + acc
        self.by[0] = z
        return z


def _py_analyze(l, r, rate, tab):
    fl, fr = _PyFilter(*tab[rate]), _PyFilter(*tab[rate])
    W = rate * 50 // 1000
    hist = np.zeros(12000, dtype=np.uint32)
    lsum = rsum = 0.0
    n = 0
    peak = 0.0

    def finish():
        nonlocal lsum, rsum, n
        if n == 0:
            return
        ms = (lsum + rsum) / float(n) * 0.5
        val = 100.0 * 10.0 * math.log10(ms + 1e-37)
        iv = 0 if val != val else int(max(-2147483648.0, min(2147483647.0, val)))
        idx = iv + 2000
        if 0 <= idx < 12000:
            hist[idx] += 1
        lsum = rsum = 0.0
        n = 0

    for i in range(len(l)):
        ln = float(l[i])
        peak = max(peak, abs(ln))
        a = fl.process(ln * 32768.0)
        if r is not None:
            rn = float(r[i])
            peak = max(peak, abs(rn))
            b = fr.process(rn * 32768.0)
            lsum += a * a
            rsum += b * b
        else:
            q = a * a
            lsum += q
            rsum += q
        n += 1
        if n >= W:
            finish()
    finish()
    return hist, peak


@pytest.mark.parametrize("rate,stereo", [(44100, True), (48000, True), (8000, False), (96000, True)])
def test_oracle_bit_exact_vs_python_transcription(oracle, rate, stereo):
    tab = _header_table()
    n = rate // 4 + 17  # 5 windows + a partial one
    l = oracle.synth_f32(77, 0, rate, n)
    r = oracle.synth_f32(77, 1, rate, n) if stereo else None
    hist_py, peak_py = _py_analyze(l, r, rate, tab)
    res, hist = oracle.analyze_pcm(l, r, rate)
    assert np.array_equal(hist, hist_py)
    assert res["peak"] == peak_py
    # and the filter outputs themselves, bit for bit
    import ctypes as C

    f = C.create_string_buffer(8 + 28 * 8 + 64)
    assert oracle.lib().rgo_filter_init(f, rate) == 0
    pf = _PyFilter(*tab[rate])
    for v in l[:300]:
        assert oracle.lib().rgo_filter_process(f, float(v) * 32768.0) == pf.process(float(v) * 32768.0)


@pytest.mark.parametrize("rate", [44100, 48000, 32000])
def test_oracle_bins_vs_scipy_lfilter(oracle, rate):
    """Independent formulation: two lfilter calls + reshape; same bins (rounding-level differences
    in z cannot move a window across a 0.01 dB boundary except with negligible probability)."""
    from scipy.signal import lfilter

    tab = _header_table()
    ya, yb, ba, bb = tab[rate]
    W = rate * 50 // 1000
    n = W * 40
    l = oracle.synth_f32(91, 0, rate, n)
    r = oracle.synth_f32(91, 1, rate, n)
    _, hist = oracle.analyze_pcm(l, r, rate)

    def chain(x):
        x = x.astype(np.float64) * 32768.0
        # the +1e-10 per stage is an input offset of 1e-10 to each recursion
        y = lfilter(yb, ya, x) + lfilter([1e-10], ya, np.ones_like(x))
        return lfilter(bb, ba, y) + lfilter([1e-10], ba, np.ones_like(x))

    zl, zr = chain(l), chain(r)
    ms = ((zl ** 2).reshape(-1, W).sum(1) + (zr ** 2).reshape(-1, W).sum(1)) / W * 0.5
    idx = np.trunc(1000.0 * np.log10(ms + 1e-37)).astype(np.int64) + 2000
    h2 = np.bincount(idx[(idx >= 0) & (idx < 12000)], minlength=12000).astype(np.uint32)
    assert np.array_equal(hist, h2)


def test_packetwise_equals_whole_buffer(oracle):
    """State carries across packets (replaygain.rs:866-904): 1152-frame packets == one buffer."""
    rate, n = 44100, 44100 + 500
    l, r = oracle.synth_f32(3, 0, rate, n), oracle.synth_f32(3, 1, rate, n)
    whole, _ = oracle.analyze_pcm(l, r, rate)
    st = oracle.TrackStream(rate, 2)
    for o in range(0, n, 1152):
        st.push(l[o:o + 1152], r[o:o + 1152])
    assert st.finish() == whole


def test_window_rules(oracle):
    rate, W = 44100, 2205
    x = oracle.synth_f32(9, 0, rate, 3 * W + 1)
    _, h = oracle.analyze_pcm(x, x, rate)
    assert h.sum() == 4  # the partial last window is counted (replaygain.rs:907)
    z = np.zeros(W * 3, dtype=np.float32)
    res, h = oracle.analyze_pcm(z, z, rate)
    assert h.sum() == 0 and res["loudness_db"] == -20.0 and res["gain_db"] == 64.82 + 20.0  # dropped, not clamped
    res, h = oracle.analyze_pcm(z[:0], None, rate)
    assert h.sum() == 0 and res["peak"] == 0.0


def test_mono_counts_the_channel_twice(oracle):
    """add_mono_sample adds x^2 to both sums (replaygain.rs:731-740) == stereo with L == R."""
    rate, n = 44100, 22050
    x = oracle.synth_f32(13, 0, rate, n)
    a, ha = oracle.analyze_pcm(x, None, rate)
    b, hb = oracle.analyze_pcm(x, x, rate)
    assert np.array_equal(ha, hb) and a == b


def test_s16_s32_scaling(oracle):
    """S16 samples go in unscaled, F32 are multiplied by 32768 (replaygain.rs:969,990): an int16
    signal and the same signal as float/32768 give identical histograms; S32 << 16 likewise."""
    rate, n = 44100, 22050
    rng = np.random.default_rng(5)
    s16 = rng.integers(-30000, 30000, n).astype(np.int16)
    f32 = (s16.astype(np.float32) / 32768.0).astype(np.float32)
    s32 = s16.astype(np.int32) << 16
    r16, h16 = oracle.analyze_pcm(s16, s16, rate)
    rf, hf = oracle.analyze_pcm(f32, f32, rate)
    r32, h32 = oracle.analyze_pcm(s32, s32, rate)
    assert np.array_equal(h16, hf) and np.array_equal(h16, h32)
    assert r16["peak"] == rf["peak"] == r32["peak"] == np.abs(s16.astype(np.float64)).max() / 32768.0


def test_gain_steps_and_clip_rule(oracle):
    L = oracle.lib()
    assert L.rgo_gain_from_loudness(72.97) == 64.82 - 72.97
    assert [L.rgo_gain_steps(v) for v in (0.74, 0.75, 0.76, -0.75, -0.74, 2.25, -2.25)] == [0, 1, 1, -1, 0, 2, -2]
    # src/main.rs:2033-2058
    assert L.rgo_clip_limit_steps(4, 6.0, 0.9, 1, 0) == 1      # -20log10(0.9)=0.915 dB -> round(0.61) = 1
    assert L.rgo_clip_limit_steps(4, 6.0, 0.9, 0, 0) == 4      # warn only
    assert L.rgo_clip_limit_steps(4, 6.0, 0.9, 1, 1) == 4      # wrap mode: no check
    assert L.rgo_clip_limit_steps(4, 6.0, 0.3, 1, 0) == 4      # no clipping
    assert L.rgo_clip_limit_steps(-2, -3.0, 1.0, 1, 0) == -2   # only positive steps are checked
    assert L.rgo_clip_limit_steps(3, 4.5, 1.0, 1, 0) == 0      # peak 1.0 -> 0 dB headroom


def test_golden_vectors(oracle):
    g = json.loads((ROOT / "tests" / "golden" / "oracle_vectors.json").read_text())
    for case in g["cases"]:
        rate, n, seed, ch = case["rate"], case["frames"], case["seed"], case["channels"]
        l = oracle.synth_f32(seed, 0, rate, n)
        r = oracle.synth_f32(seed, 1, rate, n) if ch == 2 else None
        assert hashlib.sha256(l.tobytes()).hexdigest() == case["sha256_ch0"]  # generator is pinned too
        res, hist = oracle.analyze_pcm(l, r, rate)
        nz = np.nonzero(hist)[0]
        assert res["loudness_db"] == case["loudness_db"]
        assert res["peak"] == case["peak"]
        assert [[int(i), int(hist[i])] for i in nz] == case["hist_nonzero"]
