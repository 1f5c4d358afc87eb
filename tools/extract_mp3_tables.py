#!/usr/bin/env python3
"""Generate mp3rgain_amd/csrc/rg_mp3_tables.h: the tabulated constants of ISO/IEC 11172-3 / 13818-3 Layer III
that have no closed form -- the Huffman code tables (Table B.7), the count1 tables A/B, the scalefactor-band
partitions (Table B.8 and the 13818-3 / "MPEG-2.5" extensions) and the 512-tap synthesis window (Table B.3, as the
integers D[i] * 65536 of its first 257 entries).

Neither /root/reference (whose decoder is the un-vendored symphonia crate) nor this image holds the standard's
text, so the numbers are read out of the data section of a conformant decoder that happens to be in the image: the
ffmpeg build inside the `kaleido` wheel's headless Chromium (dist-packages/kaleido/executable/bin/kaleido).  The
tables are the standard's, identical in every decoder; what this script adds is the checking:
  * every Huffman table must be a complete prefix code (Kraft sum exactly 1, no code a prefix of another);
  * the tables small enough to know by heart (1, 2, 3, 5, count1 A/B, the 44.1 kHz band partitions) are typed in
    below and must match;
  * band partitions must sum to 576 (long) and 192 (short);
  * the window must have the symmetry D[512 - i] = -/+ D[i] structure the standard prescribes (checked through
    its known landmarks: D[0] = 0, D[256] * 65536 = 75038, D[64] * 65536 = 213).
The generated header is committed (the build never needs kaleido); tests/test_mp3dec.py re-checks the
structural properties on the header itself and pins its sha256.
"""
import hashlib
import mmap
import struct
import sys
from fractions import Fraction
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
OUT = ROOT / "mp3rgain_amd" / "csrc" / "rg_mp3_tables.h"


def find_binary() -> Path:
    import kaleido

    p = Path(kaleido.__file__).resolve().parent / "executable" / "bin" / "kaleido"
    if not p.exists():
        raise SystemExit(f"{p} not found")
    return p


HUFF_DIMS = [(1, 2), (2, 3), (3, 3), (5, 4), (6, 4), (7, 6), (8, 6), (9, 6), (10, 8), (11, 8), (12, 8), (13, 16), (15, 16),
             (16, 16), (24, 16)]
KNOWN_LENS = {1: [1, 3, 2, 3], 2: [1, 3, 6, 3, 3, 5, 5, 5, 6], 3: [2, 2, 6, 3, 2, 5, 5, 5, 6],
              5: [1, 3, 6, 7, 3, 3, 6, 7, 6, 6, 7, 8, 7, 6, 7, 8]}
KNOWN_CODES = {1: [1, 1, 1, 0], 2: [1, 2, 1, 3, 1, 1, 3, 2, 0], 3: [3, 2, 1, 1, 1, 1, 3, 2, 0]}
QUAD_BITS_A = [1, 4, 4, 5, 4, 6, 5, 6, 4, 5, 5, 6, 5, 6, 6, 6]
QUAD_CODES_A = [1, 5, 4, 5, 6, 5, 4, 4, 7, 3, 6, 0, 7, 2, 3, 1]
LONG_441 = [4, 4, 4, 4, 4, 4, 6, 6, 8, 8, 10, 12, 16, 20, 24, 28, 34, 42, 50, 54, 76, 158]
SHORT_441 = [4, 4, 4, 4, 6, 8, 10, 12, 14, 18, 22, 30, 56]
# row order of the band tables in the decoder the numbers are read from: 44.1, 48, 32, 22.05, 24, 16, 11.025, 12, 8 kHz
BAND_RATES = [44100, 48000, 32000, 22050, 24000, 16000, 11025, 12000, 8000]


def check_prefix_code(lens, codes, name):
    kraft = sum(Fraction(1, 1 << l) for l in lens)
    if kraft != 1:
        raise SystemExit(f"{name}: Kraft sum {kraft} != 1")
    words = sorted((format(c, "0%db" % l) for l, c in zip(lens, codes)))
    for a, b in zip(words, words[1:]):
        if b.startswith(a):
            raise SystemExit(f"{name}: {a} is a prefix of {b}")


def main():
    path = find_binary()
    with open(path, "rb") as f:
        m = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
        # ---- Huffman tables: per table lens (n*n bytes) then codes (n*n uint16), arrays >= 16 bytes 16-aligned ----
        pos = m.find(bytes(KNOWN_LENS[1]) + struct.pack("<4H", *KNOWN_CODES[1]))
        if pos < 0:
            raise SystemExit("Huffman table 1 not found")
        huff = {}
        for tab, n in HUFF_DIMS:
            cnt = n * n
            while m[pos] == 0:
                pos += 1
            if cnt >= 16:
                assert pos % 16 == 0, (tab, hex(pos))
            lens = list(m[pos:pos + cnt])
            pos += cnt
            pos = (pos + 15) & ~15 if cnt * 2 >= 16 else (pos + 1) & ~1
            while m[pos] == 0 and m[pos + 1] == 0:
                pos += 2 if cnt * 2 < 16 else 16
            codes = list(struct.unpack("<%dH" % cnt, m[pos:pos + 2 * cnt]))
            pos += 2 * cnt
            check_prefix_code(lens, codes, f"huffman table {tab}")
            if tab in KNOWN_LENS:
                assert lens == KNOWN_LENS[tab], tab
            if tab in KNOWN_CODES:
                assert codes == KNOWN_CODES[tab], tab
            huff[tab] = (n, lens, codes)
        # ---- count1 tables ----
        qb = m.find(bytes(QUAD_BITS_A))
        quad_bits = [list(m[qb:qb + 16]), list(m[qb + 16:qb + 32])]
        qc = m.find(bytes(QUAD_CODES_A), qb)
        quad_codes = [list(m[qc:qc + 16]), list(m[qc + 16:qc + 32])]
        assert quad_bits[1] == [4] * 16 and quad_codes[1] == [15 - i for i in range(16)]
        for t in range(2):
            check_prefix_code(quad_bits[t], quad_codes[t], f"count1 table {'AB'[t]}")
        # ---- band partitions ----
        bl = m.find(bytes(LONG_441))
        band_long = [list(m[bl + 22 * r:bl + 22 * r + 22]) for r in range(9)]
        bs = m.find(bytes(SHORT_441))
        band_short = [list(m[bs + 13 * r:bs + 13 * r + 13]) for r in range(9)]
        for r in range(9):
            assert sum(band_long[r]) == 576, (r, band_long[r])
            assert sum(band_short[r]) == 192, (r, band_short[r])
        # ---- synthesis window ----
        wpos = m.find(struct.pack("<16i", 0, -1, -1, -1, -1, -1, -1, -2, -2, -2, -2, -3, -3, -4, -4, -5))
        win = list(struct.unpack("<257i", m[wpos:wpos + 257 * 4]))
        assert win[0] == 0 and win[256] == 75038 and win[64] == 213, (win[64], win[256])
        sha = hashlib.sha256(m[:]).hexdigest() if "--sha" in sys.argv else None

    def arr(ctype, name, vals, per_line=16, dims=""):
        s = f"static const {ctype} {name}{dims}[{len(vals)}] = {{\n" if not dims else f"static const {ctype} {name}{dims} = {{\n"
        for i in range(0, len(vals), per_line):
            s += "    " + ", ".join(str(v) for v in vals[i:i + per_line]) + ",\n"
        return s + "};\n"

    o = []
    o.append("/* rg_mp3_tables.h -- GENERATED by tools/extract_mp3_tables.py; do not edit.\n"
             " *\n"
             " * Tabulated constants of ISO/IEC 11172-3 (MPEG-1 audio) Layer III and its 13818-3 low-sampling-frequency\n"
             " * extension: Huffman code tables (Annex B Table B.7) as (length, code) per (x, y) symbol in row-major\n"
             " * x*n + y order, count1 tables A and B, scalefactor-band widths (Table B.8; rows 44.1, 48, 32, 22.05, 24,\n"
             " * 16, 11.025, 12, 8 kHz) and the first 257 entries of the synthesis window D[i] (Table B.3) as\n"
             " * round(D[i] * 65536); D[512 - i] follows by the window's symmetry.  Provenance and the structural checks\n"
             " * (complete prefix codes, partition sums, landmarks) are in the generator's docstring.\n"
             " */\n#pragma once\n#include <stdint.h>\n\n")
    for tab, n in HUFF_DIMS:
        _, lens, codes = huff[tab]
        o.append(arr("uint8_t", f"kMp3HuffLen{tab}", lens))
        o.append(arr("uint16_t", f"kMp3HuffCode{tab}", codes))
    o.append("\nstruct RgMp3HuffSpec { int n; const uint8_t *len; const uint16_t *code; };\n")
    o.append("/* indexed by the Huffman table number of the standard (0, 4 and 14 are empty; 16..23 and 24..31 share codes) */\n")
    o.append("static const RgMp3HuffSpec kMp3Huff[32] = {\n")
    for t in range(32):
        base = t if t < 16 else (16 if t < 24 else 24)
        if base in (0, 4, 14):
            o.append("    {0, 0, 0},\n")
        else:
            n = dict(HUFF_DIMS)[base]
            o.append(f"    {{{n}, kMp3HuffLen{base}, kMp3HuffCode{base}}},\n")
    o.append("};\n")
    o.append("static const uint8_t kMp3Linbits[32] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 2, 3, 4, 6, 8, 10, 13, 4, 5, 6, 7, 8, 9, 11, 13};\n\n")
    o.append(arr("uint8_t", "kMp3QuadLenA", quad_bits[0]))
    o.append(arr("uint8_t", "kMp3QuadCodeA", quad_codes[0]))
    o.append("\n/* scalefactor-band widths; row = rate index in the order above */\n")
    o.append("static const uint8_t kMp3BandLong[9][22] = {\n" + "".join("    {" + ", ".join(map(str, r)) + "},\n" for r in band_long) + "};\n")
    o.append("static const uint8_t kMp3BandShort[9][13] = {\n" + "".join("    {" + ", ".join(map(str, r)) + "},\n" for r in band_short) + "};\n")
    o.append("static const uint32_t kMp3BandRates[9] = {" + ", ".join(map(str, BAND_RATES)) + "};\n\n")
    o.append(arr("int32_t", "kMp3SynthWindowQ16", win, per_line=12))
    text = "".join(o)
    OUT.write_text(text)
    print("wrote", OUT, hashlib.sha256(text.encode()).hexdigest())
    if sha:
        print("source binary sha256", sha)


if __name__ == "__main__":
    main()
