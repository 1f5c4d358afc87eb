"""MPEG-1/2/2.5 Layer III bitstream WRITER -- test infrastructure only (nothing under mp3rgain_amd/ imports it).

The reference ships four MP3 fixtures, all long-block-dominated 44.1 kHz encodes of one sine.  To exercise the decoder's
other paths (short / mixed / start / stop blocks, sub-block gains, intensity and mid/side stereo, scfsi reuse,
preflag, both count1 tables, every Huffman table and linbits width, CRC words, the bit reservoir, the MPEG-2 and
MPEG-2.5 scalefactor syntax and band tables) this module writes syntactically valid streams from chosen side
information, scalefactors and quantised spectral values -- the inverse of the decoder's stage A, with no psychoacoustics
and no analysis filterbank: the "music" is random quantised spectra.  A conformant decoder must turn such a stream
into the same PCM as any other conformant decoder, which is what tools/make_mp3_golden.py uses it for (ffmpeg's
decoder, through the image's kaleido/Chromium, produces the golden PCM committed under tests/golden/mp3/).

Syntax: ISO/IEC 11172-3 2.4.1 (header, side information, main data), 13818-3 2.4.1 for the low-sampling-frequency
extension.  Huffman tables are read from the generated header mp3rgain_amd/csrc/rg_mp3_tables.h.
"""
from __future__ import annotations

import random
import re
from dataclasses import dataclass, field
from pathlib import Path
from typing import List, Optional

ROOT = Path(__file__).resolve().parent.parent

BITRATES_V1 = [0, 32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320]
BITRATES_V2 = [0, 8, 16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128, 144, 160]
RATES = {44100: (3, 0), 48000: (3, 1), 32000: (3, 2), 22050: (2, 0), 24000: (2, 1), 16000: (2, 2),
         11025: (0, 0), 12000: (0, 1), 8000: (0, 2)}  # rate -> (version bits, rate index)
RATE_ROW = {44100: 0, 48000: 1, 32000: 2, 22050: 3, 24000: 4, 16000: 5, 11025: 6, 12000: 7, 8000: 8}
LINBITS = [0] * 16 + [1, 2, 3, 4, 6, 8, 10, 13, 4, 5, 6, 7, 8, 9, 11, 13]
SLEN = [[0, 0, 0, 0, 3, 1, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4], [0, 1, 2, 3, 0, 1, 2, 3, 1, 2, 3, 1, 2, 3, 2, 3]]
LSF_PARTITIONS = [
    [[6, 5, 5, 5], [9, 9, 9, 9], [6, 9, 9, 9]], [[6, 5, 7, 3], [9, 9, 12, 6], [6, 9, 12, 6]],
    [[11, 10, 0, 0], [18, 18, 0, 0], [15, 18, 0, 0]], [[7, 7, 7, 0], [12, 12, 12, 0], [6, 15, 12, 0]],
    [[6, 6, 6, 3], [12, 9, 9, 6], [6, 12, 9, 6]], [[8, 8, 5, 0], [15, 12, 9, 0], [6, 18, 9, 0]]]

_tables = None


def tables():
    """Huffman (len, code) lists, count1 table A and the band partitions, parsed from the generated C header."""
    global _tables
    if _tables is None:
        txt = (ROOT / "mp3rgain_amd" / "csrc" / "rg_mp3_tables.h").read_text()

        def arr(name):
            m = re.search(r"\b%s\b[^=]*=\s*\{(.*?)\};" % name, txt, re.S)
            return [int(x) for x in re.findall(r"-?\d+", m.group(1))]

        huff = {}
        for t in (1, 2, 3, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15, 16, 24):
            lens, codes = arr(f"kMp3HuffLen{t}"), arr(f"kMp3HuffCode{t}")
            n = int(round(len(lens) ** 0.5))
            huff[t] = (n, lens, codes)
        for t in range(17, 24):
            huff[t] = huff[16]
        for t in range(25, 32):
            huff[t] = huff[24]
        bl = arr("kMp3BandLong")
        bs = arr("kMp3BandShort")
        long_rows = [bl[22 * r:22 * r + 22] for r in range(9)]
        short_rows = [bs[13 * r:13 * r + 13] for r in range(9)]
        long_rows[4][17], long_rows[4][18] = 54, 62  # 24 kHz: band 18 starts at 332 (see rg_mp3dec.cpp)
        cum = lambda w: [sum(w[:i]) for i in range(len(w) + 1)]  # noqa: E731
        _tables = {"huff": huff, "quadA": (arr("kMp3QuadLenA"), arr("kMp3QuadCodeA")),
                   "sfb_long": [cum(r) for r in long_rows], "sfb_short": [cum(r) for r in short_rows]}
    return _tables


class BitWriter:
    def __init__(self):
        self.bits: List[int] = []

    def put(self, value: int, n: int):
        for k in range(n - 1, -1, -1):
            self.bits.append((value >> k) & 1)

    def __len__(self):
        return len(self.bits)

    def to_bytes(self) -> bytes:
        b = self.bits + [0] * (-len(self.bits) % 8)
        return bytes(int("".join(map(str, b[i:i + 8])), 2) for i in range(0, len(b), 8))


@dataclass
class GranuleSpec:
    """What one granule of one channel carries.  `values`: 576 quantised lines in BITSTREAM order (short blocks:
    [band][window][line])."""
    values: List[int]
    global_gain: int = 150
    block_type: int = 0            # 0 normal, 1 start, 2 short, 3 stop
    mixed: bool = False
    subblock_gain: List[int] = field(default_factory=lambda: [0, 0, 0])
    scalefac_scale: int = 0
    preflag: int = 0               # MPEG-1 only (LSF derives it from scalefac_compress)
    count1table: int = 0
    scalefac_compress: int = 0     # MPEG-1: 0..15; LSF: 0..511
    scalefacs: Optional[List[int]] = None   # flat, in transmission order; None = random within the slen limits
    scfsi: List[int] = field(default_factory=lambda: [0, 0, 0, 0])  # granule 1 of MPEG-1 long blocks only
    table_select: Optional[List[int]] = None  # None = chosen (at random among the feasible ones)
    region0_count: Optional[int] = None
    region1_count: Optional[int] = None
    # filled by the encoder
    part2_3_length: int = 0
    big_values: int = 0


def _max_for_table(t: int) -> int:
    if t == 0:
        return 0
    n = tables()["huff"][t][0]
    return (n - 1) + ((1 << LINBITS[t]) - 1 if LINBITS[t] else 0)


def lsf_slen(sfc: int, intensity_right: bool):
    if not intensity_right:
        if sfc < 400:
            return [(sfc >> 4) // 5, (sfc >> 4) % 5, (sfc & 15) >> 2, sfc & 3], 0
        if sfc < 500:
            s = sfc - 400
            return [(s >> 2) // 5, (s >> 2) % 5, s & 3, 0], 1
        s = sfc - 500
        return [s // 3, s % 3, 0, 0], 2
    s = sfc >> 1
    if s < 180:
        return [s // 36, (s % 36) // 6, (s % 36) % 6, 0], 3
    if s < 244:
        s -= 180
        return [(s & 0x3F) >> 4, (s & 0xF) >> 2, s & 3, 0], 4
    s -= 244
    return [s // 3, s % 3, 0, 0], 5


def scalefactor_widths(g: GranuleSpec, lsf: bool, intensity_right: bool, gr: int) -> List[int]:
    """Bit width of every transmitted scalefactor of the granule, in transmission order (0-width ones included
    for LSF, where they are 'transmitted' as nothing but still occupy a slot)."""
    if not lsf:
        s1, s2 = SLEN[0][g.scalefac_compress], SLEN[1][g.scalefac_compress]
        if g.block_type == 2:
            if g.mixed:
                return [s1] * 8 + [s1] * 9 + [s2] * 18
            return [s1] * 18 + [s2] * 18
        out = []
        for k, (lo, hi) in enumerate([(0, 6), (6, 11), (11, 16), (16, 21)]):
            if gr == 1 and g.scfsi[k]:
                continue
            out += [s1 if k < 2 else s2] * (hi - lo)
        return out
    slen, sset = lsf_slen(g.scalefac_compress, intensity_right)
    kind = 2 if (g.block_type == 2 and g.mixed) else (1 if g.block_type == 2 else 0)
    out = []
    for k in range(4):
        out += [slen[k]] * LSF_PARTITIONS[sset][kind][k]
    return out


def encode_granule(g: GranuleSpec, rate: int, lsf: bool, intensity_right: bool, gr: int, rng: random.Random) -> BitWriter:
    """Main data of one granule/channel (part 2 + part 3); fills g.part2_3_length, big_values, table_select, regions."""
    T = tables()
    row = RATE_ROW[rate]
    w = BitWriter()
    widths = scalefactor_widths(g, lsf, intensity_right, gr)
    if g.scalefacs is None:
        g.scalefacs = [rng.randrange(1 << b) if b else 0 for b in widths]
    assert len(g.scalefacs) == len(widths), (len(g.scalefacs), len(widths))
    for v, b in zip(g.scalefacs, widths):
        assert 0 <= v < (1 << b) or (b == 0 and v == 0)
        w.put(v, b)
    vals = list(g.values)
    assert len(vals) == 576
    # big_values / count1 split: trailing zeros dropped; quadruples of |v| <= 1 at the top form the count1 region
    last = 576
    while last > 0 and vals[last - 1] == 0:
        last -= 1
    last += last & 1
    c1_start = last
    while c1_start >= 4 and all(abs(v) <= 1 for v in vals[c1_start - 4:c1_start]) and (c1_start - 4) % 2 == 0:
        c1_start -= 4
    # count1 covers [c1_start, c1_end) in quadruples
    c1_end = c1_start + ((last - c1_start + 3) // 4) * 4
    if c1_end > 576:
        c1_end = 576
    g.big_values = c1_start // 2
    # regions
    if g.block_type != 0:
        window_switching = True
        if g.block_type == 2:
            r0 = 3 * T["sfb_short"][row][3]
        else:
            r0 = T["sfb_long"][row][8]
        bounds = [min(r0, c1_start), c1_start, c1_start]
    else:
        window_switching = False
        if g.region0_count is None:
            g.region0_count = rng.randrange(16)
            g.region1_count = rng.randrange(8)
        i0 = min(22, g.region0_count + 1)
        i1 = min(22, g.region0_count + g.region1_count + 2)
        bounds = [min(T["sfb_long"][row][i0], c1_start), min(T["sfb_long"][row][i1], c1_start), c1_start]
    if g.table_select is None:
        g.table_select = []
        lo = 0
        for r in range(3 if not window_switching else 2):
            hi = bounds[r]
            mx = max([abs(v) for v in vals[lo:hi]] or [0])
            cands = [t for t in range(1, 32) if t not in (4, 14) and _max_for_table(t) >= mx]
            if mx == 0 and rng.random() < 0.5:
                cands = [0]
            g.table_select.append(rng.choice(cands))
            lo = hi
        if window_switching:
            g.table_select.append(0)
    lo = 0
    for r in range(3):
        hi = bounds[r]
        t = g.table_select[r]
        if hi > lo:
            assert t != 0 or all(v == 0 for v in vals[lo:hi]), "table 0 region must be all zero"
            if t != 0:
                n, lens, codes = T["huff"][t]
                lb = LINBITS[t]
                for i in range(lo, hi, 2):
                    x, y = vals[i], vals[i + 1]
                    ax, ay = abs(x), abs(y)
                    cx, cy = (15 if (lb and ax >= 15) else ax), (15 if (lb and ay >= 15) else ay)
                    assert cx < n and cy < n, (t, ax, ay)
                    w.put(codes[cx * n + cy], lens[cx * n + cy])
                    if lb and ax >= 15:
                        w.put(ax - 15, lb)
                    if ax:
                        w.put(1 if x < 0 else 0, 1)
                    if lb and ay >= 15:
                        w.put(ay - 15, lb)
                    if ay:
                        w.put(1 if y < 0 else 0, 1)
        lo = hi
    ql, qc = T["quadA"]
    for i in range(c1_start, c1_end, 4):
        q = vals[i:i + 4]
        v = sum((1 if q[k] else 0) << (3 - k) for k in range(4))
        if g.count1table:
            w.put(15 - v, 4)
        else:
            w.put(qc[v], ql[v])
        for k in range(4):
            if q[k]:
                w.put(1 if q[k] < 0 else 0, 1)
    g.part2_3_length = len(w)
    assert g.part2_3_length < 4096
    return w


@dataclass
class FrameSpec:
    granules: List[List[GranuleSpec]]   # [granule][channel]
    bitrate_kbps: int
    mode: int = 0                       # 0 stereo, 1 joint, 2 dual, 3 mono
    mode_ext: int = 0
    crc: bool = False
    padding: int = 0


def crc16(bits: List[int]) -> int:
    crc = 0xFFFF
    for b in bits:
        top = (crc >> 15) & 1
        crc = (crc << 1) & 0xFFFF
        if top ^ b:
            crc ^= 0x8005
    return crc


def write_stream(frames: List[FrameSpec], rate: int, rng: random.Random, stuffing: bool = True) -> bytes:
    """Assemble frames with a shared bit reservoir (main_data_begin back-pointers)."""
    ver_bits, rate_idx = RATES[rate]
    lsf = ver_bits != 3
    out_frames = []
    # first pass: encode main data of every frame
    mains = []
    for f in frames:
        nch = 1 if f.mode == 3 else 2
        w = BitWriter()
        for gr, chans in enumerate(f.granules):
            assert len(chans) == nch
            for ch, g in enumerate(chans):
                gw = encode_granule(g, rate, lsf, ch == 1 and f.mode == 1 and bool(f.mode_ext & 1), gr, rng)
                w.bits += gw.bits
        mains.append(w.to_bytes())
    # second pass: place main data.  S = start of the frame's own data area in the global main-data stream,
    # P = where the frame's main data actually starts (<= S); main_data_begin = S - P.
    max_back = 255 if lsf else 511
    S = 0
    P = 0
    stream = bytearray()
    placements = []
    for i, f in enumerate(frames):
        nch = 1 if f.mode == 3 else 2
        side_bytes = (9 if nch == 1 else 17) if lsf else (17 if nch == 1 else 32)
        table = BITRATES_V2 if lsf else BITRATES_V1
        fb = (72 if lsf else 144) * f.bitrate_kbps * 1000 // rate + f.padding
        cap = fb - 4 - (2 if f.crc else 0) - side_bytes
        assert cap > 0
        if P < S - max_back or not stuffing:
            P = max(P, S - max_back) if stuffing else S
        if P > S:
            raise ValueError(f"frame {i}: previous main data overran into this frame's area (raise the bitrate)")
        mdb = S - P
        if P + len(mains[i]) > S + cap:
            raise ValueError(f"frame {i}: main data ({len(mains[i])} B from -{mdb}) does not fit (capacity {cap})")
        if len(stream) < P + len(mains[i]):
            stream.extend(b"\0" * (P + len(mains[i]) - len(stream)))
        stream[P:P + len(mains[i])] = mains[i]
        placements.append((S, cap, mdb, side_bytes, fb, table.index(f.bitrate_kbps)))
        P += len(mains[i])
        S += cap
    stream.extend(b"\0" * max(0, S - len(stream)))
    # third pass: headers + side info + the frame's slice of the main-data stream
    for f, (S_i, cap, mdb, side_bytes, fb, br_idx) in zip(frames, placements):
        nch = 1 if f.mode == 3 else 2
        h = BitWriter()
        h.put(0x7FF, 11)
        h.put(ver_bits, 2)
        h.put(1, 2)                      # Layer III
        h.put(0 if f.crc else 1, 1)      # protection_bit: 0 = CRC present
        h.put(br_idx, 4)
        h.put(rate_idx, 2)
        h.put(f.padding, 1)
        h.put(0, 1)
        h.put(f.mode, 2)
        h.put(f.mode_ext, 2)
        h.put(0, 1)
        h.put(1, 1)
        h.put(0, 2)
        s = BitWriter()
        if not lsf:
            s.put(mdb, 9)
            s.put(0, 5 if nch == 1 else 3)
            for ch in range(nch):
                for k in range(4):
                    s.put(f.granules[1][ch].scfsi[k], 1)
        else:
            s.put(mdb, 8)
            s.put(0, 1 if nch == 1 else 2)
        for chans in f.granules:
            for g in chans:
                s.put(g.part2_3_length, 12)
                s.put(g.big_values, 9)
                s.put(g.global_gain, 8)
                s.put(g.scalefac_compress, 9 if lsf else 4)
                ws = 1 if g.block_type != 0 else 0
                s.put(ws, 1)
                if ws:
                    s.put(g.block_type, 2)
                    s.put(1 if g.mixed else 0, 1)
                    s.put(g.table_select[0], 5)
                    s.put(g.table_select[1], 5)
                    for k in range(3):
                        s.put(g.subblock_gain[k], 3)
                else:
                    for k in range(3):
                        s.put(g.table_select[k], 5)
                    s.put(g.region0_count, 4)
                    s.put(g.region1_count, 3)
                if not lsf:
                    s.put(g.preflag, 1)
                s.put(g.scalefac_scale, 1)
                s.put(g.count1table, 1)
        assert len(s) == side_bytes * 8, (len(s), side_bytes)
        frame = bytearray(h.to_bytes())
        if f.crc:
            c = crc16(h.bits[16:] + s.bits)
            frame += bytes([c >> 8, c & 0xFF])
        frame += s.to_bytes()
        frame += stream[S_i:S_i + cap]
        assert len(frame) == fb, (len(frame), fb)
        out_frames.append(bytes(frame))
    return b"".join(out_frames)


# ---- content helpers -------------------------------------------------------------------------------------------
def random_spectrum(rng: random.Random, nonzero_lines: int, big: int, tail_ones: int = 24, huge_every: int = 0) -> List[int]:
    """576 quantised values: magnitudes decaying with frequency up to `big`, then a stretch of -1/0/+1, then zeros."""
    v = [0] * 576
    nonzero_lines -= nonzero_lines & 1
    for i in range(nonzero_lines):
        m = max(1, int(big * (1.0 - i / max(1, nonzero_lines)) ** 2))
        v[i] = rng.randint(-m, m)
        if huge_every and i % huge_every == 3:
            v[i] = rng.choice([-1, 1]) * rng.randint(15, 15 + huge_every * 37)
    for i in range(nonzero_lines, min(576, nonzero_lines + tail_ones)):
        v[i] = rng.choice([-1, 0, 0, 1])
    return v
