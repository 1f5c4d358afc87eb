"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/mp3rgain_amd.h
declares, its pure host helpers agree with the oracle, and without a GPU it fails loudly
instead of falling back to anything."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest
from hypothesis import given, settings
from hypothesis import strategies as st

ROOT = Path(__file__).resolve().parent.parent


def _declared_functions():
    txt = (ROOT / "include" / "mp3rgain_amd.h").read_text() + (ROOT / "include" / "mp3rgain_amd_node.h").read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(rg_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(capi):
    from mp3rgain_amd import _capi

    declared = _declared_functions()
    bound = sorted(name for name, _, _ in _capi.SYMBOLS)
    assert declared == bound, "header and ctypes binding disagree"
    raw = C.CDLL(str(_capi.LIB_PATH))
    for name in declared:
        assert hasattr(raw, name), f"{name} not exported"
    import __graft_entry__

    assert capi.rg_abi_version() == __graft_entry__.header_abi_version()
    assert capi.rg_is_available() == 1


def test_struct_layouts_match_the_header():
    from mp3rgain_amd import _capi

    assert C.sizeof(_capi.TrackDesc) == 24
    assert C.sizeof(_capi.TrackResult) == 48
    assert C.sizeof(_capi.AlbumResult) == 32
    assert C.sizeof(_capi.PeakResult) == 24
    assert C.sizeof(_capi.DeviceView) == 40


def test_supported_rates_and_windows(capi):
    rates = [96000, 88200, 64000, 48000, 44100, 32000, 24000, 22050, 16000, 12000, 11025, 8000]
    for r in rates:
        assert capi.rg_supported_rate(r) == 1
        assert capi.rg_window_samples(r) == r * 50 // 1000
    for r in (0, 99999, 44101, 192000):
        assert capi.rg_supported_rate(r) == 0
    assert capi.rg_window_samples(44100) == 2205 and capi.rg_window_samples(11025) == 551


def test_design_info(capi):
    """Every row but 88.2 kHz is stable; the halo grows with the pole radius."""
    st_, halo, dec = C.c_int(), C.c_uint32(), C.c_double()
    halos = {}
    for r in [96000, 64000, 48000, 44100, 32000, 24000, 22050, 16000, 12000, 11025, 8000]:
        assert capi.rg_rate_design_info(r, C.byref(st_), C.byref(halo), C.byref(dec)) == 0
        assert st_.value == 1 and 256 <= halo.value <= 8192 and 0.9 < dec.value < 1.0
        halos[r] = halo.value
    assert halos[96000] > halos[44100] > halos[8000]
    assert capi.rg_rate_design_info(88200, C.byref(st_), C.byref(halo), C.byref(dec)) == 0
    assert st_.value == 0
    assert capi.rg_rate_design_info(12345, None, None, None) == -2


@settings(max_examples=200, deadline=None)
@given(st.lists(st.tuples(st.integers(0, 11999), st.integers(1, 5000)), min_size=0, max_size=40))
def test_host_percentile_matches_oracle(capi, oracle, entries):
    h = np.zeros(12000, dtype=np.uint32)
    for i, c in entries:
        h[i] += c
    assert capi.rg_hist_loudness(h.ctypes.data) == oracle.hist_loudness(h)


def test_host_gain_helpers_match_oracle(capi, oracle):
    L = oracle.lib()
    for v in np.linspace(-30, 30, 241):
        assert capi.rg_gain_steps(float(v)) == L.rgo_gain_steps(float(v))
        assert capi.rg_db_to_steps(float(v)) == L.rgo_gain_steps(float(v))
        assert capi.rg_gain_from_loudness(float(v)) == L.rgo_gain_from_loudness(float(v))
    assert capi.rg_steps_to_db(3) == 4.5 and capi.rg_steps_to_db(-2) == -3.0
    for steps in (-3, 0, 1, 4, 9):
        for gain in (-4.0, 1.4, 6.0, 13.3):
            for peak in (0.05, 0.5, 0.9, 1.0, 1.3):
                for k in (0, 1):
                    for w in (0, 1):
                        assert capi.rg_clip_limit_steps(steps, gain, peak, k, w) == \
                            L.rgo_clip_limit_steps(steps, gain, peak, k, w)


def test_no_gpu_means_a_loud_failure_not_a_fallback(capi):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import mp3rgain_amd as rg

    assert capi.rg_create(0) is None
    msg = capi.rg_last_error(None).decode()
    assert "no CPU path" in msg
    with pytest.raises(rg.ReplayGainError):
        rg.Analyzer(0)
    with pytest.raises(rg.ReplayGainError):
        rg.analyze_track(rg.PcmTrack([np.zeros(10, np.float32)], 44100))


def test_pack_tracks_layout():
    import mp3rgain_amd as rg
    from mp3rgain_amd.replaygain import pack_tracks

    a = np.arange(5, dtype=np.float32)
    b = np.arange(3, dtype=np.int16)
    arena, descs = pack_tracks([rg.PcmTrack([a, a + 10], 44100), rg.PcmTrack([b], 48000)])
    assert descs[0].offset_bytes == 0 and descs[0].frames == 5 and descs[0].channels == 2 and descs[0].format == 0
    assert descs[1].offset_bytes == 48 and descs[1].frames == 3 and descs[1].channels == 1 and descs[1].format == 1
    assert np.array_equal(arena[:20].view(np.float32), a) and np.array_equal(arena[20:40].view(np.float32), a + 10)
    assert np.array_equal(arena[48:54].view(np.int16), b)
    with pytest.raises(ValueError):
        rg.PcmTrack([], 44100)
    with pytest.raises(TypeError):
        rg.PcmTrack([np.zeros(4, np.float64)], 44100)


def test_product_does_not_reference_the_oracle():
    """The shipped path must not import, include or link anything under oracle/."""
    for p in list((ROOT / "mp3rgain_amd").rglob("*.py")) + list((ROOT / "mp3rgain_amd" / "csrc").glob("*")):
        if p.suffix in (".py", ".hip", ".cpp", ".h") or p.name == "Makefile":
            assert "oracle" not in p.read_text().lower().replace("no cpu", ""), p


def test_tm_design_keeps_all_moments_when_the_segment_is_shorter_than_the_fast_decay(capi):
    """rg_tm_design (host): H10 = frames for which all 12 transient moments are accumulated.  A multiple of 4 when it
    ends inside the segment; the whole segment, trailing L & 3 frames included, when the fast block outlives it
    (cutting at L & ~3 was a bug: L = 245 at 44.1 kHz, 150 at 24 kHz)."""
    def h10(rate, L):
        H, r, rf, res = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_double()
        assert capi.rg_tm_design_info(rate, L, C.byref(H), C.byref(r), C.byref(rf), C.byref(res), None, None) == 0
        return H.value, r.value, rf.value, res.value

    # (round 6: the cut is 1e-10 of the responses' maximum -- 184 frames at 44.1 kHz; it was 1e-13, 248 frames, when L = 245 was
    # such a segment)
    assert h10(44100, 2205)[0] == 184 and h10(44100, 735)[0] == 184
    assert h10(44100, 245)[0] == 184
    assert capi.rg_tm_design_info(44100, 147, None, None, None, None, None, None) != 0  # would need more than 16 predecessors
    assert h10(24000, 150)[0] == 150 and h10(24000, 1200)[0] == 248 and h10(24000, 300)[0] == 248
    for rate, L in ((44100, 2205), (44100, 245), (48000, 2400), (8000, 400), (24000, 150)):
        H, rounds, rounds_fast, resid = h10(rate, L)
        assert H <= L and (H % 4 == 0 or H == L) and 1 <= rounds <= 4 and rounds_fast <= rounds and resid < 1e-15


def test_graft_entry_build_succeeds():
    """The driver's "does it build" check: make (a no-op when up to date), import, ABI agreement with the header."""
    import __graft_entry__

    __graft_entry__.build()


def _coeffs(rate):
    import re as _re

    src = (ROOT / "include" / "rg_coeffs.h").read_text()
    rows = _re.findall(r"\{\s*(\d+)u,\s*\{([^}]*)\},\s*\{([^}]*)\},\s*\{([^}]*)\},\s*\{([^}]*)\},", src)
    for r in rows:
        if int(r[0]) == rate:
            return [[float(v) for v in part.split(",")] for part in r[1:]]  # yule a, yule b, butter a, butter b
    raise KeyError(rate)


@pytest.mark.parametrize("rate", [96000, 64000, 48000, 44100, 32000, 22050, 8000])
def test_design_reproduces_the_references_response_to_silence(capi, rate):
    """The affine side of variant 2's design, checked on the host against the reference's own recursion (src/replaygain.rs:586-616,
    restated here in extended precision): with zero input the reference's output -- driven by its "+1e-10" per stage alone -- must
    equal, frame for frame over the first window,
        servo form    d_inf + sigma0 . T[n]                      (linear lanes: nothing else contributes)
        classic form  (the lanes' zero-state output with their injected 1e-10) + sigma0 . T[n]
    which pins the track-start state, the decoupling X, the response table and d_inf = 1e-10 / beta at once."""
    import numpy as np

    ya, yb, ba, bb = _coeffs(rate)
    W = rate * 50 // 1000
    H, r, rf, res = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_double()
    T = np.zeros((W, 12))
    assert capi.rg_tm_design_info(rate, W, C.byref(H), C.byref(r), C.byref(rf), C.byref(res), T.ctypes.data, None) == 0
    servo, alpha, beta, g, dinf = C.c_int(), C.c_double(), C.c_double(), C.c_double(), C.c_double()
    s0 = np.zeros(12)
    assert capi.rg_tm_design_affine(rate, W, C.byref(servo), C.byref(alpha), C.byref(beta), C.byref(g), C.byref(dinf), s0.ctypes.data) == 0
    exact = bb[1] == -2.0 * bb[0] and bb[2] == bb[0]
    assert bool(servo.value) == exact
    ld = np.longdouble
    c = ld(1e-10)
    # the reference, zero input, zero history
    yx, yy, bx, by = [ld(0)] * 11, [ld(0)] * 11, [ld(0)] * 3, [ld(0)] * 3
    aff = []
    for n in range(W):
        yx = [ld(0)] + yx[:10]
        yy = [ld(0)] + yy[:10]
        acc = ld(0)
        for i in range(1, 11):
            acc += ld(yb[i]) * yx[i] - ld(ya[i]) * yy[i]
        y = (c + ld(yb[0]) * yx[0]) + acc
        yy[0] = y
        bx = [y] + bx[:2]
        by = [ld(0)] + by[:2]
        acc2 = ld(0)
        for i in range(1, 3):
            acc2 += ld(bb[i]) * bx[i] - ld(ba[i]) * by[i]
        z = (c + ld(bb[0]) * bx[0]) + acc2
        by[0] = z
        aff.append(z)
    aff = np.array(aff, dtype=ld)
    hom = (T.astype(ld) * s0.astype(ld)).sum(axis=1)
    if servo.value:
        assert alpha.value == 2.0 + ba[1] and abs(beta.value - (1.0 + ba[1] + ba[2])) <= 1e-18 and g.value == bb[0]
        assert abs(dinf.value * beta.value - 1e-10) <= 1e-22
        model = ld(dinf.value) + hom
    else:
        assert dinf.value == 0.0
        # the classic lanes: DF2T with 1e-10 injected at the deepest state of each stage, zero input, zero state
        s, t, zs = [ld(0)] * 10, [ld(0)] * 2, []
        for n in range(W):
            y = s[0]
            s = [s[i + 1] - ld(ya[i + 1]) * y for i in range(9)] + [c - ld(ya[10]) * y]
            z = ld(bb[0]) * y + t[0]
            t = [t[1] + ld(bb[1]) * y - ld(ba[1]) * z, c + ld(bb[2]) * y - ld(ba[2]) * z]
            zs.append(z)
        model = np.array(zs, dtype=ld) + hom
    scale = float(np.max(np.abs(aff)))
    assert float(np.max(np.abs(model - aff))) <= 2e-12 * scale, (float(np.max(np.abs(model - aff))), scale)
