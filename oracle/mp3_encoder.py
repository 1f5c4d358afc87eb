"""A small MPEG Layer III ENCODER -- test infrastructure only (nothing under mp3rgain_amd/ imports it).

oracle/mp3_bitstream.py writes streams from chosen quantised values: syntax coverage, but the "audio" is random spectra.
This module puts a real signal through the standard's encoder chain, so that the decoder is also exercised by dense,
music-like content with the statistics an encoder produces (every region on the table that codes it cheapest, window
switching at real attacks, mid/side frames where the channels agree, a working bit reservoir):

    PCM -> 32-band polyphase analysis (ISO/IEC 11172-3 C.1.3: window C = synthesis window D / 32, matrixing
           cos((2k+1)(i-16)pi/64)) -> per subband MDCT over 36 (long / start / stop windows) or 3 x 12 (short) samples
           (2.4.3.4.10.3 inverted) -> alias butterflies for long blocks (the decoder's rotation, inverted)
        -> mid/side where the side channel is weak -> scalefactors from the band energies (long blocks) -> one quantiser
           step per frame, found by bisection on the frame's bit budget (C.1.5.4.4 without the psychoacoustic model:
           noise is white per frame) -> Huffman tables chosen by bit count -> oracle/mp3_bitstream.write_stream.

There is no psychoacoustic model; the streams sound like a 1990 encoder at a low setting.  They are valid Layer III,
which is all a decoder test needs: the golden PCM comes from ffmpeg's decoder (tools/make_mp3_dense.py), and the
round trip PCM -> encode -> this repo's decoder -> PCM is checked against the input signal as well.
"""
from __future__ import annotations

import math
import random
from typing import List, Tuple

import numpy as np

import mp3_bitstream as B

_win = None


def synthesis_window() -> np.ndarray:
    """The standard's 512-tap synthesis window D, rebuilt from the 257 tabulated values the way rg_mp3dec.cpp does."""
    global _win
    if _win is None:
        import re

        txt = (B.ROOT / "mp3rgain_amd" / "csrc" / "rg_mp3_tables.h").read_text()
        m = re.search(r"kMp3SynthWindowQ16\[257\]\s*=\s*\{(.*?)\};", txt, re.S)
        q = np.array([int(x) for x in re.findall(r"-?\d+", m.group(1))], dtype=np.float64) / 65536.0
        d = np.zeros(512)
        d[:257] = q
        for i in range(1, 256):
            d[512 - i] = -q[i] if (i & 63) else q[i]
        _win = d
    return _win


def polyphase_analysis(x: np.ndarray) -> np.ndarray:
    """x: one channel, length a multiple of 32 -> subband samples S[slot][32] (11172-3 figure C.4)."""
    d = synthesis_window()
    c = d / 32.0  # analysis window C (checked: analysis -> the standard's synthesis reconstructs to 86 dB, delay 481)
    nslots = len(x) // 32
    padded = np.concatenate([np.zeros(480), x])
    # X_t[i] = x[32 t + 31 - i]  (newest sample first)
    idx = (np.arange(nslots)[:, None] * 32 + 31 + 480) - np.arange(512)[None, :]
    z = padded[idx] * c[None, :]
    y = z.reshape(nslots, 8, 64).sum(axis=1)
    k = np.arange(32)[:, None]
    i = np.arange(64)[None, :]
    m = np.cos((2 * k + 1) * (i - 16) * math.pi / 64.0)
    return y @ m.T


def _windows():
    i = np.arange(36)
    normal = np.sin(math.pi / 36 * (i + 0.5))
    start = normal.copy()
    start[18:24] = 1.0
    start[24:30] = np.sin(math.pi / 12 * (np.arange(24, 30) - 18 + 0.5))
    start[30:] = 0.0
    stop = start[::-1].copy()
    short = np.sin(math.pi / 12 * (np.arange(12) + 0.5))
    return normal, start, short, stop


def mdct_granules(sub: np.ndarray, block_types: List[int]) -> np.ndarray:
    """sub: S[slot][32] for one channel (slots = 18 * granules) -> xr[granule][576] in the decoder's line order
    (long: 18 lines per subband; short: per subband 3 windows x 6 lines interleaved as line*3 + window, the order the
    decoder's short IMDCT reads before the bitstream reordering is undone)."""
    ngr = sub.shape[0] // 18
    s = sub.copy()
    s[1::2, 1::2] *= -1.0  # frequency inversion: odd subbands, odd time slots
    prev = np.zeros((18, 32))
    normal, start, short, stop = _windows()
    k18 = np.arange(18)[:, None]
    i36 = np.arange(36)[None, :]
    c36 = np.cos(math.pi / 72 * (2 * i36 + 1 + 18) * (2 * k18 + 1))
    k6 = np.arange(6)[:, None]
    i12 = np.arange(12)[None, :]
    c12 = np.cos(math.pi / 24 * (2 * i12 + 1 + 6) * (2 * k6 + 1))
    out = np.zeros((ngr, 576))
    for g in range(ngr):
        cur = s[18 * g:18 * g + 18]
        z = np.concatenate([prev, cur], axis=0)  # [36][32]
        bt = block_types[g]
        if bt == 2:
            xr = np.zeros((32, 18))
            for w in range(3):
                seg = z[6 + 6 * w:18 + 6 * w] * short[:, None]  # [12][32]
                xw = (c12 @ seg) / 3.0  # [6][32]
                xr[:, w::3] = xw.T  # line = 3*k + w
            out[g] = xr.reshape(576)
        else:
            win = (normal, start, None, stop)[bt]
            xr = (c36 @ (z * win[:, None])) / 9.0  # [18][32]
            out[g] = xr.T.reshape(576)
        prev = cur
    return out


_CS = None


def alias_coeffs():
    global _CS
    if _CS is None:
        ci = np.array([-0.6, -0.535, -0.33, -0.185, -0.095, -0.041, -0.0142, -0.0037])
        _CS = (1.0 / np.sqrt(1.0 + ci * ci), ci / np.sqrt(1.0 + ci * ci))
    return _CS


def alias_encode(xr: np.ndarray):
    """The inverse of the decoder's alias reduction (a rotation per butterfly), in place on one long granule."""
    cs, ca = alias_coeffs()
    for sb in range(1, 32):
        for i in range(8):
            lo, up = 18 * sb - 1 - i, 18 * sb + i
            a, b = xr[lo], xr[up]
            xr[lo] = a * cs[i] + b * ca[i]
            xr[up] = b * cs[i] - a * ca[i]


def short_to_bitstream_order(xr: np.ndarray, rate: int) -> np.ndarray:
    """Decoder line order of a short granule (line = 3*k + w over the whole spectrum) -> [band][window][line]."""
    sfb = B.tables()["sfb_short"][B.RATE_ROW[rate]]
    out = np.zeros(576)
    pos = 0
    for b in range(13):
        lo, hi = sfb[b], sfb[b + 1]
        for w in range(3):
            for k in range(lo, hi):
                out[pos] = xr[3 * k + w]
                pos += 1
    return out


# ---- bit counting ------------------------------------------------------------------------------------------------
_HL = None


def _huff_len_cube():
    """HL[t][x][y]: code length incl. sign bits and linbits of table t for clipped magnitudes x, y <= 15; inf where
    the table cannot code the pair.  max_val[t]: largest codable magnitude."""
    global _HL
    if _HL is None:
        T = B.tables()["huff"]
        hl = np.full((32, 16, 16), np.inf)
        mx = np.zeros(32, dtype=np.int64)
        for t in range(1, 32):
            if t in (4, 14):
                continue
            n, lens, _ = T[t]
            lb = B.LINBITS[t]
            a = np.array(lens, dtype=np.float64).reshape(n, n)
            sign = (np.arange(n) > 0).astype(np.float64)
            a = a + sign[:, None] + sign[None, :]
            if lb:
                esc = (np.arange(n) == 15).astype(np.float64) * lb
                a = a + esc[:, None] + esc[None, :]
            hl[t, :n, :n] = a
            mx[t] = (n - 1) + ((1 << lb) - 1 if lb else 0)
        _HL = (hl, mx)
    return _HL


def best_table(v: np.ndarray) -> Tuple[int, float]:
    """Cheapest table for the pairs in v (absolute values, even length) -> (table, bits)."""
    if len(v) == 0 or v.max() == 0:
        return 0, 0.0
    hl, mx = _huff_len_cube()
    m = int(v.max())
    x = np.minimum(v[0::2], 15)
    y = np.minimum(v[1::2], 15)
    cost = hl[:, x, y].sum(axis=1)
    cost[mx < m] = np.inf
    cost[0] = np.inf
    t = int(np.argmin(cost))
    return t, float(cost[t])


def plan_granule(ix: np.ndarray, rate: int, block_type: int):
    """Regions, tables and the bit count of the Huffman part for quantised magnitudes ix[576] (bitstream order)."""
    T = B.tables()
    row = B.RATE_ROW[rate]
    nz = np.nonzero(ix)[0]
    last = int(nz[-1]) + 1 if len(nz) else 0
    last += last & 1
    c1 = last
    while c1 >= 4 and ix[c1 - 4:c1].max() <= 1:
        c1 -= 4
    c1_end = min(576, c1 + ((last - c1 + 3) // 4) * 4)
    quads = ix[c1:c1_end].reshape(-1, 4) if c1_end > c1 else np.zeros((0, 4), dtype=np.int64)
    ql, _ = T["quadA"]
    vq = (quads[:, 0] > 0) * 8 + (quads[:, 1] > 0) * 4 + (quads[:, 2] > 0) * 2 + (quads[:, 3] > 0)
    signs = float((quads > 0).sum())
    bits_a = float(np.array(ql)[vq].sum()) + signs if len(vq) else 0.0
    bits_b = 4.0 * len(vq) + signs
    count1table = 1 if bits_b < bits_a else 0
    bits = min(bits_a, bits_b)
    if block_type != 0:
        r0 = 3 * T["sfb_short"][row][3] if block_type == 2 else T["sfb_long"][row][8]
        bounds = [min(r0, c1), c1, c1]
        r0c = r1c = None
    else:
        sfb = T["sfb_long"][row]
        nb = next(b for b in range(1, 23) if sfb[b] >= c1) if c1 > 0 else 1
        r0c = max(0, min(15, nb // 3 - 1))
        r1c = max(0, min(7, nb // 4 - 1))
        bounds = [min(sfb[min(22, r0c + 1)], c1), min(sfb[min(22, r0c + r1c + 2)], c1), c1]
    tabs = []
    lo = 0
    for r in range(3 if block_type == 0 else 2):
        t, b = best_table(ix[lo:bounds[r]])
        tabs.append(t)
        bits += b
        lo = bounds[r]
    if block_type != 0:
        tabs.append(0)
    return dict(table_select=tabs, region0_count=r0c, region1_count=r1c, count1table=count1table, bits=bits)


def quantise(xr_scaled: np.ndarray, gg: int) -> np.ndarray:
    """11172-3 C.1.5.4.4.1: ix = nint((|xr| / 2^((gg-210)/4))^(3/4) - 0.0946)."""
    step = 2.0 ** (-(gg - 210) * 0.1875)
    return np.floor(np.abs(xr_scaled) ** 0.75 * step + 0.4054).astype(np.int64)


PRETAB = [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 3, 2, 0]


def encode(pcm: np.ndarray, rate: int, bitrate: int, seed: int = 1, allow_ms: bool = True, allow_short: bool = True,
           scale: float = 1.0) -> bytes:
    """pcm: float [channels][frames] in [-1, 1] -> a CBR Layer III stream (MPEG-1, -2 or -2.5 by the rate)."""
    nch = pcm.shape[0]
    lsf = rate < 32000
    ngr_per_frame = 1 if lsf else 2
    spf = 576 * ngr_per_frame
    n = pcm.shape[1]
    nframes = (n + 576 + spf - 1) // spf  # one extra granule flushes the MDCT overlap
    x = np.zeros((nch, nframes * spf))
    x[:, :n] = pcm * scale
    ngr = nframes * ngr_per_frame
    subs = [polyphase_analysis(x[c]) for c in range(nch)]
    # window switching: an attack = the granule's energy envelope (three 6-slot thirds) jumping by > 8x against the
    # running level of the previous granule
    want_short = [False] * (ngr + 2)
    if allow_short:
        for g in range(1, ngr):
            e_prev = sum(float((s[18 * (g - 1):18 * g] ** 2).sum()) for s in subs) / 3.0 + 1e-3 * (scale / 32768.0) ** 2
            thirds = [sum(float((s[18 * g + 6 * k:18 * g + 6 * k + 6] ** 2).sum()) for s in subs) for k in range(3)]
            if max(thirds) > 8.0 * e_prev and max(thirds) > 1e4 * (scale / 32768.0) ** 2:
                want_short[g] = True
    # the standard's window-switching state machine: normal -> start -> short ... short -> stop -> normal | start
    bts = [0] * ngr
    for g in range(ngr):
        prev = bts[g - 1] if g else 0
        if prev == 1:
            bts[g] = 2
        elif prev == 2:
            bts[g] = 2 if (want_short[g] or want_short[g + 1]) else 3
        else:
            bts[g] = 1 if want_short[g + 1] else 0
    specs = [mdct_granules(subs[c], bts) for c in range(nch)]
    for c in range(nch):
        for g in range(ngr):
            if bts[g] != 2:
                alias_encode(specs[c][g])
            else:
                specs[c][g] = short_to_bitstream_order(specs[c][g], rate)
    T = B.tables()
    row = B.RATE_ROW[rate]
    sfb_long = T["sfb_long"][row]
    table = B.BITRATES_V2 if lsf else B.BITRATES_V1
    assert bitrate in table
    side_bytes = (9 if nch == 1 else 17) if lsf else (17 if nch == 1 else 32)
    max_back = 255 if lsf else 511
    frames: List[B.FrameSpec] = []
    reservoir = 0   # bytes of earlier frames' areas still unused (what main_data_begin may reach back into)
    rest = 0.0      # fractional bytes of the nominal frame length: decides the padding bit
    exact = (72 if lsf else 144) * bitrate * 1000 / rate
    for f in range(nframes):
        base_len = int(exact)
        rest += exact - base_len
        padding = 0
        if rest >= 1.0:
            padding = 1
            rest -= 1.0
        cap = base_len + padding - 4 - side_bytes
        grs = range(f * ngr_per_frame, (f + 1) * ngr_per_frame)
        use_ms = False
        if nch == 2 and allow_ms:
            mid = sum(float(((specs[0][g] + specs[1][g]) ** 2).sum()) for g in grs)
            side = sum(float(((specs[0][g] - specs[1][g]) ** 2).sum()) for g in grs)
            use_ms = side < 0.12 * mid
        chans = []
        for g in grs:
            if use_ms:
                m = (specs[0][g] + specs[1][g]) / math.sqrt(2.0)
                s = (specs[0][g] - specs[1][g]) / math.sqrt(2.0)
                chans.append([m, s])
            else:
                chans.append([specs[c][g] for c in range(nch)])
        # scalefactors (long blocks): bands well below the granule's loudest band get a finer step
        plans = []
        for gi, g in enumerate(grs):
            row_plans = []
            for c in range(nch):
                xr = chans[gi][c]
                sf = None
                amp = np.ones(576)
                if bts[g] != 2 and not lsf:
                    e = np.array([float((xr[sfb_long[b]:sfb_long[b + 1]] ** 2).mean()) + 1e-9 for b in range(21)])
                    emax = e.max()
                    sf = []
                    for b in range(21):
                        lim = 7 if b < 11 else 3
                        v = int(max(0.0, min(lim, math.floor(0.25 * math.log2(emax / e[b])))))
                        sf.append(v)
                        amp[sfb_long[b]:sfb_long[b + 1]] = 2.0 ** (0.5 * v)   # scalefac_scale = 0, no preflag
                row_plans.append(dict(xs=xr * amp, sf=sf))
            plans.append(row_plans)
        # one quantiser step for the whole frame: the smallest global_gain whose bits fit the budget
        avail = cap + reservoir
        budget_bits = 8 * min(avail, cap + int(0.6 * reservoir)) - 8
        part2 = 0 if lsf else 53 * nch * sum(1 for g in grs if bts[g] != 2)   # 11 x 3 + 10 x 2 bits of scalefactors
        def frame_bits(gg):
            total = part2
            out = []
            for gi, g in enumerate(grs):
                for c in range(nch):
                    ix = quantise(plans[gi][c]["xs"], gg)
                    if ix.max() > 8191 + 14:
                        return None, math.inf
                    p = plan_granule(ix, rate, bts[g])
                    if p["bits"] + (53 if (bts[g] != 2 and not lsf) else 0) >= 4095:
                        return None, math.inf
                    total += p["bits"]
                    out.append((ix, p))
            return out, total
        lo_g, hi_g = 90, 255
        best = None
        while lo_g < hi_g:
            mid_g = (lo_g + hi_g) // 2
            out, bits = frame_bits(mid_g)
            if bits <= budget_bits:
                best = (mid_g, out, bits)
                hi_g = mid_g
            else:
                lo_g = mid_g + 1
        if best is None or best[0] != lo_g:
            out, bits = frame_bits(lo_g)
            assert bits <= budget_bits, "a frame does not fit at the coarsest step"
            best = (lo_g, out, bits)
        gg, out, bits = best
        k = 0
        gr_specs = []
        for gi, g in enumerate(grs):
            cs = []
            for c in range(nch):
                ix, p = out[k]
                k += 1
                xs = plans[gi][c]["xs"]
                vals = [int(v) if xs[i] >= 0 else -int(v) for i, v in enumerate(ix)]
                spec = B.GranuleSpec(values=vals, global_gain=gg, block_type=bts[g], count1table=p["count1table"],
                                     table_select=p["table_select"], region0_count=p["region0_count"],
                                     region1_count=p["region1_count"])
                if bts[g] != 2:
                    if lsf:
                        spec.scalefac_compress = 0
                        spec.scalefacs = [0] * len(B.scalefactor_widths(spec, True, False, 0))
                    else:
                        spec.scalefac_compress = 12   # slen1 = 3, slen2 = 2
                        spec.scalefacs = list(plans[gi][c]["sf"])
                else:
                    spec.scalefac_compress = 0
                    spec.scalefacs = [0] * len(B.scalefactor_widths(spec, lsf, False, 0))
                cs.append(spec)
            gr_specs.append(cs)
        frames.append(B.FrameSpec(granules=gr_specs, bitrate_kbps=bitrate, mode=(3 if nch == 1 else (1 if allow_ms else 0)),
                                  mode_ext=(2 if use_ms else 0), padding=padding))
        used = (int(bits) + 7) // 8
        reservoir = min(max_back, avail - used)
    return B.write_stream(frames, rate, random.Random(seed))
