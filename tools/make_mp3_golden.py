#!/usr/bin/env python3
"""Write the synthetic MP3 streams of tests/golden/mp3/ and their golden PCM.

    tools/make_mp3_golden.py            streams (oracle/mp3_bitstream.py, seeded) + ffmpeg PCM via kaleido
    tools/make_mp3_golden.py --no-pcm   streams only (kaleido not needed)

Each case is a handful of frames of random quantised spectra built to walk one part of the Layer III syntax; the golden
PCM comes from ffmpeg's decoder inside the image's headless Chromium (tools/ffmpeg_golden.py).  Also decodes the
reference's own fixtures (copied under tests/golden/fixtures/).  tests/test_mp3dec.py compares the library's decoder
with these files; nothing there needs kaleido.
"""
import random
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))
sys.path.insert(0, str(ROOT / "tools"))

import mp3_bitstream as B  # noqa: E402

OUT = ROOT / "tests" / "golden" / "mp3"


def legal_is_positions(g, lsf, rng):
    """Right-channel scalefactors = intensity positions that every decoder reads the same way: MPEG-1 any value
    (>= 7 means 'not intensity coded'); LSF below both the standard's illegal value 2^slen - 1 and 16."""
    widths = B.scalefactor_widths(g, lsf, True, 0)
    if not lsf:
        return [rng.randrange(1 << b) if b else 0 for b in widths]
    return [rng.randrange(max(1, min((1 << b) - 1, 16))) if b else 0 for b in widths]


# Block types must follow the standard's window-switching state machine (normal -> start -> short ... -> stop -> normal):
# decoders may, and ffmpeg's does, use that a short block's predecessor leaves zeros in the last third of its overlap.
def build_case(name, rate, mode, mode_ext, nframes, seed, *, block_types=(0,), mixed_prob=0.0, bitrate=None, crc=False,
               gg=(150, 165), big=12, lines=(200, 520), huge_every=0, is_cut=None, sfc_lsf=None, stuffing=True,
               padding_every=0, sbg=False, return_specs=False, spikes=None):
    rng = random.Random(seed)
    lsf = rate < 32000
    nch = 1 if mode == 3 else 2
    ngr = 1 if lsf else 2
    table = B.BITRATES_V2 if lsf else B.BITRATES_V1
    bitrate = bitrate or table[-1]
    frames = []
    bt_i = 0
    for fi in range(nframes):
        grs = []
        for gr in range(ngr):
            chans = []
            bt = block_types[bt_i % len(block_types)]
            bt_i += 1
            mixed = bt == 2 and rng.random() < mixed_prob  # both channels alike, as joint stereo requires
            for ch in range(nch):
                nlines = rng.randint(*lines)
                if is_cut is not None and ch == 1:
                    nlines = min(nlines, is_cut)
                vals = B.random_spectrum(rng, nlines, big, tail_ones=0 if (is_cut is not None and ch == 1) else rng.choice([0, 16, 40]),
                                         huge_every=huge_every)
                if spikes:  # (count, lo, hi): large magnitudes anywhere below the last nonzero line, high lines included
                    for _ in range(spikes[0]):
                        vals[rng.randrange(max(2, nlines - (nlines & 1)))] = rng.choice([-1, 1]) * rng.randint(spikes[1], spikes[2])
                g = B.GranuleSpec(values=vals, global_gain=rng.randint(*gg), block_type=bt,
                                  mixed=mixed,
                                  scalefac_scale=rng.randrange(2), count1table=rng.randrange(2))
                if sbg and bt == 2:
                    g.subblock_gain = [rng.randrange(4) for _ in range(3)]
                if lsf:
                    right_is = ch == 1 and mode == 1 and (mode_ext & 1)
                    if sfc_lsf:
                        g.scalefac_compress = rng.choice(sfc_lsf)
                    elif right_is:
                        g.scalefac_compress = rng.choice([2 * 20, 2 * 100 + 1, 2 * 190, 2 * 230 + 1, 2 * 245, 2 * 250 + 1])
                    else:
                        g.scalefac_compress = rng.choice([0, 77, 250, 399, 400, 433, 499, 500, 505, 511])
                    if right_is:
                        g.scalefacs = legal_is_positions(g, True, rng)
                else:
                    g.scalefac_compress = rng.randrange(16)
                    g.preflag = rng.randrange(2) if bt != 2 else 0
                    if gr == 1 and bt == 0 and grs and grs[0][ch].block_type == 0:
                        g.scfsi = [rng.randrange(2) for _ in range(4)]
                chans.append(g)
            grs.append(chans)
        frames.append(B.FrameSpec(granules=grs, bitrate_kbps=bitrate, mode=mode, mode_ext=mode_ext, crc=crc,
                                  padding=1 if padding_every and fi % padding_every == 0 else 0))
    # keep every frame's main data within its capacity: thin the spectra until the stream assembles
    for attempt in range(40):
        try:
            data = B.write_stream(frames, rate, random.Random(seed + 1), stuffing=stuffing)
            return (data, frames) if return_specs else data
        except ValueError:
            for f in frames:
                for chans in f.granules:
                    for g in chans:
                        nz = max((i for i, v in enumerate(g.values) if v), default=0)
                        cut = int(nz * 0.85)
                        g.values = g.values[:cut] + [0] * (576 - cut)
                        g.table_select = None
                        g.region0_count = None
    raise SystemExit(f"{name}: could not fit the main data")


def build_huffman_sweep(seed):
    """Every (x, y) codeword of every Huffman table once (plus linbits escapes of every width), mono, one table per
    granule with all three regions on that table; count1 table A then B on alternating granules."""
    rng = random.Random(seed)
    tabs = [t for t in range(1, 32) if t not in (4, 14)]
    grs = []
    for k, t in enumerate(tabs):
        n = B.tables()["huff"][t][0]
        lb = B.LINBITS[t]
        pairs = [(x, y) for x in range(n) for y in range(n)]
        rng.shuffle(pairs)
        vals = []
        for x, y in pairs:
            for v in (x, y):
                if lb and v == 15:
                    v += rng.randrange(min(1 << lb, 48))
                vals.append(-v if rng.random() < 0.5 else v)
        ones = [rng.choice([-1, 0, 1]) for _ in range(min(576 - len(vals), 32) // 4 * 4)]
        vals = vals + ones + [0] * (576 - len(vals) - len(ones))
        grs.append(B.GranuleSpec(values=vals, global_gain=138 if n == 16 else 170, table_select=[t, t, t], region0_count=3,
                                 region1_count=3, count1table=k & 1, scalefac_compress=0))
    if len(grs) & 1:
        grs.append(B.GranuleSpec(values=[0] * 576, global_gain=100, table_select=[0, 0, 0], region0_count=0, region1_count=0))
    frames = [B.FrameSpec(granules=[[grs[i]], [grs[i + 1]]], bitrate_kbps=320, mode=3) for i in range(0, len(grs), 2)]
    return B.write_stream(frames, 44100, random.Random(seed + 1))


CASES = [
    # name, rate, mode, mode_ext, frames, seed, options
    ("v1_44k_stereo_long", 44100, 0, 0, 8, 101, dict(bitrate=320, huge_every=0)),
    ("v1_44k_stereo_linbits", 44100, 0, 0, 6, 102, dict(bitrate=320, huge_every=7, gg=(120, 135), lines=(60, 200))),
    # large values (above 127, above the 704 of the back half's LDS power table, up to the escape limit) at every height of the
    # spectrum, above line 256 and 512 too: the device format's second byte plane at its full length
    ("v1_44k_ms_spikes_high", 44100, 1, 2, 8, 110, dict(bitrate=320, gg=(114, 126), big=3, lines=(300, 576), spikes=(14, 128, 2500),
                                                        block_types=(0, 0, 1, 2, 3, 0))),
    ("v1_48k_ms_blocktypes", 48000, 1, 2, 10, 103, dict(bitrate=320, block_types=(0, 1, 2, 2, 3, 0, 0, 1, 2, 3), sbg=True)),
    ("v1_44k_ms_mixed", 44100, 1, 2, 8, 104, dict(bitrate=320, block_types=(1, 2, 2, 3), mixed_prob=1.0, sbg=True)),
    ("v1_32k_intensity", 32000, 1, 1, 8, 105, dict(bitrate=256, is_cut=120)),
    ("v1_44k_intensity_ms_short", 44100, 1, 3, 8, 106, dict(bitrate=320, block_types=(0, 1, 2, 2, 3, 0), is_cut=96, mixed_prob=0.5)),
    ("v1_44k_mono_crc_reservoir", 44100, 3, 0, 12, 107, dict(bitrate=128, crc=True, lines=(80, 420), padding_every=3)),
    ("v1_44k_huffman_sweep", 44100, 3, 0, 15, 109, dict(sweep=True)),
    ("v1_48k_dual_channel", 48000, 2, 0, 6, 108, dict(bitrate=256)),
    ("v2_22k_stereo", 22050, 0, 0, 10, 201, dict(bitrate=160, block_types=(0, 0, 1, 2, 3, 0))),
    ("v2_24k_mono", 24000, 3, 0, 10, 202, dict(bitrate=96, block_types=(0, 1, 2, 3), mixed_prob=0.5, lines=(100, 330))),
    ("v2_16k_intensity", 16000, 1, 1, 10, 203, dict(bitrate=128, is_cut=140, block_types=(0, 0, 1, 2, 3))),
    ("v2_22k_intensity_ms", 22050, 1, 3, 10, 204, dict(bitrate=160, is_cut=100, block_types=(0, 1, 2, 2, 3), mixed_prob=0.3)),
    ("v25_11k_stereo", 11025, 0, 0, 10, 301, dict(bitrate=96, block_types=(0, 1, 2, 3))),
    ("v25_12k_ms", 12000, 1, 2, 10, 302, dict(bitrate=96)),
    ("v25_8k_mono", 8000, 3, 0, 10, 303, dict(bitrate=64, block_types=(0, 0, 1, 2, 3))),
]


def main():
    OUT.mkdir(parents=True, exist_ok=True)
    want_pcm = "--no-pcm" not in sys.argv
    only = [a for a in sys.argv[1:] if not a.startswith("--")]
    files = []
    for name, rate, mode, ext, n, seed, opts in CASES:
        if only and name not in only:
            continue
        data = build_huffman_sweep(seed) if opts.get("sweep") else build_case(name, rate, mode, ext, n, seed, **opts)
        (OUT / f"{name}.mp3").write_bytes(data)
        files.append(OUT / f"{name}.mp3")
        print(name, len(data), "bytes")
    if want_pcm:
        import numpy as np

        import ffmpeg_golden as G

        fixtures = [] if only else sorted(f for f in (ROOT / "tests" / "golden" / "fixtures").glob("*.mp3") if f.name != "test_stereo.mp3")
        if not only:
            # The reference's test_stereo.mp3 carries global_gain 255 in every granule but the last (someone's gain run
            # saturated it before it was committed): it decodes to +-6e8 and a fixed-point decoder overflows on it.  The
            # same bits with every global_gain field lowered by 125 steps (the library's own patcher) are a sane stream.
            from mp3rgain_amd import mp3gain

            patched, _ = mp3gain.apply_gain_data((ROOT / "tests" / "golden" / "fixtures" / "test_stereo.mp3").read_bytes(), -125)
            (OUT / "test_stereo_minus125.mp3").write_bytes(patched)
            files.append(OUT / "test_stereo_minus125.mp3")
        for f in files + fixtures:
            data = f.read_bytes()
            rate = G.mp3_rate(data)
            pcm, info = G.decode(data, rate)
            p64 = pcm.astype(np.float64)
            q = np.round(np.where(p64 > 0, p64 * 32767.0, p64 * 32768.0))  # Chromium's int16 -> float: /32767 above zero, /32768 below
            if np.abs(np.where(q > 0, q / 32767.0, q / 32768.0) - p64).max() >= 1e-6 or np.abs(pcm).max() >= 0.999:
                raise SystemExit(f"{f.name}: peak {float(np.abs(pcm).max())}: not 16-bit quantised or too hot for a fixed-point "
                                 "decoder (lower the case's global_gain range)")
            np.save(OUT / (f.stem + ".ffmpeg.npy"), np.clip(q, -32768, 32767).astype(np.int16))
            print(f.name, info["rate"], info["channels"], info["length"], "peak %.4f" % float(np.abs(pcm).max()))


if __name__ == "__main__":
    main()
