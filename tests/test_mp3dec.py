"""The MP3 decode row: mp3rgain_amd/csrc/rg_mp3dec.cpp behind
include/mp3rgain_amd_dec.h.

The reference decodes with symphonia, whose source is not in its tree and which cannot run in this image, and none of
its tests holds a decoded sample: parity with THAT decoder is unpinned.  What is pinned here:
  * against an independent conformant decoder -- ffmpeg's, run through the image's headless Chromium by
    tools/make_mp3_golden.py, outputs committed as tests/golden/mp3/*.ffmpeg.npy -- on the reference's own fixtures and
    on sixteen synthetic streams (oracle/mp3_bitstream.py) that walk the rest of the syntax: every Huffman codeword of
    every table, all block types and mixed blocks, sub-block gains, MS / intensity stereo in the MPEG-1 and the LSF
    form, scfsi, preflag, both count1 tables, CRC words, the bit reservoir, MPEG-2 and MPEG-2.5 at every rate family.
    That decoder is ffmpeg's fixed-point one (int16 output, itself good to about one step): the bar is
    max |delta| <= 1.5 and RMS <= 0.6 steps of 2^-15, which any wrong table entry, window, sign or scale misses by
    orders of magnitude;
  * the fixtures are ffmpeg encodes of a 440 Hz sine (reference .github/workflows/ci.yml:66-69): spectral peak, SNR
    against the best-fit sine, and the loudness the ReplayGain oracle gives the decoded PCM against an ideal sine;
  * packet semantics of the reference's loop (src/replaygain.rs:881-904) and container handling (ID3v2, Xing/Info);
  * PCM hashes of this build's output, so that an unintended change shows up;
  * damaged input never crashes.
"""
import hashlib
import json
import random
import re
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))

from mp3rgain_amd import mp3dec  # noqa: E402

sys.path.insert(0, str(ROOT / "tests"))
from mp3gold import FIX, GOLD, LAME_DELAY, MAX_STEPS, RMS_STEPS, STREAMS, compare_with_gold, load_gold  # noqa: E402


@pytest.mark.parametrize("path", STREAMS, ids=lambda p: p.stem)
def test_pcm_matches_ffmpeg(path):
    data = path.read_bytes()
    pcm, info = mp3dec.decode(data)
    gold = load_gold(path)
    assert gold.shape[0] == info.channels == pcm.shape[0]
    mx, rms, _, _ = compare_with_gold(pcm, info.info_frame, gold)
    assert mx <= MAX_STEPS, f"max |delta| {mx:.2f} steps of 2^-15"
    assert rms <= RMS_STEPS
    assert info.skipped_frames == 0
    assert np.abs(gold).max() > 150, "the case must carry signal well above the comparison floor"


def test_dense_streams_are_what_the_encoder_makes_and_decode_to_the_music():
    """tests/golden/mp3/dense_*.mp3 (tools/make_mp3_dense.py): tens of seconds of synthetic music through a real encoder
    chain (oracle/mp3_encoder.py).  The shortest is re-encoded here and must be the committed bytes; every one decodes to
    the piece it was made from (SNR >= 18 dB at the chain's delay -- there is no psychoacoustic model), uses window
    switching at its attacks, and the joint-stereo one switches mid/side per frame."""
    sys.path.insert(0, str(ROOT / "tools"))
    import make_mp3_dense as M
    import mp3_encoder as E

    for name, rate, nch, secs, br, seed, ms in M.CASES:
        data = (GOLD / f"{name}.mp3").read_bytes()
        pcm = M.piece(rate, secs, nch, seed)
        if name == "dense_22k_mono_56":
            assert E.encode(pcm, rate, br, seed=seed, allow_ms=ms) == data, "the committed stream is not the generator's output"
        dec, info = mp3dec.decode(data)
        assert info.skipped_frames == 0 and info.frames / rate > secs
        d = 1057  # analysis + synthesis filterbank (481) and one granule of MDCT overlap (576)
        m = min(dec.shape[1] - d, pcm.shape[1])
        err = dec[:, d:d + m] - pcm[:, :m]
        assert 10 * np.log10((pcm[:, :m] ** 2).sum() / (err ** 2).sum()) >= 18.0
        _, units, _ = mp3dec.parse_units(data)
        kinds = {int(u.block_type) for u in units}
        assert kinds == {0, 1, 2, 3}, kinds
        assert np.mean([int(u.nz) for u in units]) > 350  # dense: most of the 576 lines carry something


def test_synthetic_streams_cover_the_syntax():
    """What the golden set exercises, read back from the streams' own side information."""
    import mp3_bitstream as B

    seen_tables, block_types, mixed, sbg, versions, modes, crc, reservoir, c1, preflag, scfsi = set(), set(), 0, 0, set(), set(), 0, 0, set(), 0, 0
    for p in GOLD.glob("v*.mp3"):  # the synthetic ones: bare frames, no tags
        d = p.read_bytes()
        pos = 0
        while pos + 4 <= len(d):
            h = d[pos:pos + 4]
            ver = (h[1] >> 3) & 3
            lsf = ver != 3
            br = (B.BITRATES_V2 if lsf else B.BITRATES_V1)[h[2] >> 4]
            rate = [44100, 48000, 32000][(h[2] >> 2) & 3] >> (0 if ver == 3 else (1 if ver == 2 else 2))
            mode = h[3] >> 6
            nch = 1 if mode == 3 else 2
            has_crc = (h[1] & 1) == 0
            fb = (72 if lsf else 144) * br * 1000 // rate + ((h[2] >> 1) & 1)
            side = d[pos + 4 + (2 if has_crc else 0):]
            bits = "".join(format(b, "08b") for b in side[:32])
            q = [0]

            def g(n):
                v = int(bits[q[0]:q[0] + n], 2)
                q[0] += n
                return v

            versions.add(ver)
            modes.add((mode, (h[3] >> 4) & 3 if mode == 1 else 0))
            crc += has_crc
            reservoir += g(8 if lsf else 9) > 0
            g((1 if nch == 1 else 2) if lsf else (5 if nch == 1 else 3))
            if not lsf:
                scfsi += any(g(1) for _ in range(4 * nch))
            for _ in range((1 if lsf else 2) * nch):
                g(12), g(9), g(8), g(9 if lsf else 4)
                if g(1):
                    bt = g(2)
                    block_types.add(bt)
                    mixed += g(1)
                    seen_tables.update([g(5), g(5)])
                    sbg += any(g(3) for _ in range(3))
                else:
                    block_types.add(0)
                    seen_tables.update([g(5), g(5), g(5)])
                    g(4), g(3)
                if not lsf:
                    preflag += g(1)
                g(1)
                c1.add(g(1))
            pos += fb
    assert seen_tables >= set(range(32)) - {4, 14}
    assert block_types == {0, 1, 2, 3} and mixed and sbg and crc and reservoir and preflag and scfsi
    assert versions == {0, 2, 3} and c1 == {0, 1}
    assert {(0, 0), (2, 0), (3, 0), (1, 1), (1, 2), (1, 3)} <= modes


def test_stage_a_recovers_what_the_bitstream_writer_encoded():
    """rg_mp3_parse_units (frame walk, side information, reservoir, scalefactors, Huffman) against the source of the
    synthetic streams: every quantised value, global_gain, block type and scalefactor the writer put in comes back out,
    and the streams on disk are what the seeded generator produces."""
    sys.path.insert(0, str(ROOT / "tools"))
    import make_mp3_golden as M

    checked = 0
    for name, rate, mode, ext, n, seed, opts in M.CASES:
        if opts.get("sweep"):
            continue
        data, frames = M.build_case(name, rate, mode, ext, n, seed, return_specs=True, **opts)
        assert data == (GOLD / f"{name}.mp3").read_bytes(), f"{name}: the committed stream is not the generator's output"
        is_, units, info = mp3dec.parse_units(data)
        specs = [g for f in frames for chans in f.granules for g in chans]
        assert len(specs) == is_.shape[0] == info.audio_frames * (1 if rate < 32000 else 2) * info.channels
        for k, g in enumerate(specs):
            assert np.array_equal(is_[k], np.asarray(g.values, dtype=np.int16)), f"{name}: unit {k}"
            u = units[k]
            assert (u.global_gain, u.block_type, u.mixed, u.scalefac_scale) == (g.global_gain, g.block_type, int(g.mixed), g.scalefac_scale)
            last = max((i for i, v in enumerate(g.values) if v), default=-1)
            assert last < u.nz <= 576
            if g.block_type == 2 or rate < 32000 or not any(g.scfsi):  # transmission order == the unit's flat layout
                sent = [v for v in g.scalefacs]
                assert list(u.sf[:len(sent)]) == sent, f"{name}: unit {k} scalefactors"
            checked += 1
    assert checked >= 300


def _tables_text():
    return (ROOT / "mp3rgain_amd" / "csrc" / "rg_mp3_tables.h").read_text()


def test_generated_tables_are_pinned_and_well_formed():
    """The tabulated constants of the standard: sha256 of the generated header, and the structural checks of its
    generator repeated on the header itself (complete prefix codes, partition sums, window landmarks)."""
    from fractions import Fraction

    import mp3_bitstream as B

    txt = _tables_text()
    pin = (ROOT / "tests" / "golden" / "mp3_tables_sha256.txt").read_text().split()[0]
    assert hashlib.sha256(txt.encode()).hexdigest() == pin
    T = B.tables()
    for t, (n, lens, codes) in T["huff"].items():
        assert len(lens) == len(codes) == n * n
        assert sum(Fraction(1, 1 << l) for l in lens) == 1, f"table {t} is not a complete code"
        words = sorted(format(c, "0%db" % l) for l, c in zip(lens, codes))
        assert all(not b.startswith(a) for a, b in zip(words, words[1:])), f"table {t} is not prefix free"
    ql, qc = T["quadA"]
    assert sum(Fraction(1, 1 << l) for l in ql) == 1
    for r in range(9):
        assert T["sfb_long"][r][-1] == 576 and T["sfb_short"][r][-1] == 192
        assert all(b > a for a, b in zip(T["sfb_long"][r], T["sfb_long"][r][1:]))
    win = [int(x) for x in re.findall(r"-?\d+", re.search(r"kMp3SynthWindowQ16\[257\] = \{(.*?)\};", txt, re.S).group(1))]
    assert len(win) == 257 and win[0] == 0 and win[64] == 213 and win[256] == 75038
    # with the modulation signs removed the window is one smooth symmetric low-pass prototype: positive main lobe,
    # mirror symmetry about tap 256, DC gain 64 (= 32 subbands x 2)
    full = np.zeros(512)
    full[:257] = np.array(win) / 65536.0
    for i in range(1, 256):
        full[512 - i] = -full[i] if i & 63 else full[i]
    signs = np.array([(-1) ** (i // 64) for i in range(512)])
    proto = full * signs           # the modulation signs removed: one smooth low-pass, positive main lobe
    assert proto[201:312].min() > 0 and abs(proto.sum() - 64.0) < 0.01
    assert np.array_equal(proto[1:], proto[:0:-1])


@pytest.mark.parametrize("name", ["test_joint_stereo", "test_mono", "test_vbr"])
def test_fixtures_decode_to_the_440_hz_sine(name, oracle):
    """.github/workflows/ci.yml:66-69: `sine=frequency=440:duration=1` through libmp3lame.  A wrong Huffman entry,
    window or stereo rule breaks the sine."""
    pcm, info = mp3dec.decode((FIX / f"{name}.mp3").read_bytes())
    assert (info.sample_rate, info.audio_frames, info.info_frame, info.id3v2_bytes) == (44100, 40, 1, 44)
    assert pcm.shape[1] == 40 * 1152  # nothing trimmed (FormatOptions::default()), the Info frame not decoded
    x = pcm[0, LAME_DELAY + 2000:LAME_DELAY + 42000].astype(np.float64)  # steady part
    t = np.arange(len(x)) / 44100.0
    # best-fit sine near 440 Hz (least squares on a fine frequency grid)
    best = None
    for f in np.arange(439.0, 441.0001, 0.05):
        A = np.stack([np.sin(2 * np.pi * f * t), np.cos(2 * np.pi * f * t)], axis=1)
        coef, res, *_ = np.linalg.lstsq(A, x, rcond=None)
        r = float(((x - A @ coef) ** 2).sum())
        if best is None or r < best[0]:
            best = (r, f, coef)
    r, f, coef = best
    amp = float(np.hypot(*coef))
    snr = 10 * np.log10((amp ** 2 / 2 * len(x)) / r)
    assert abs(f - 440.0) <= 1.0
    assert snr >= 50.0, f"SNR {snr:.1f} dB"
    assert 0.05 < amp < 0.13  # lavfi's sine is 1/8 full scale; the stereo encodes carry it 3 dB lower per channel
    if pcm.shape[0] == 2:
        assert np.abs(pcm[0] - pcm[1]).max() < 2e-4  # identical channels
    # ReplayGain loudness of the decoded PCM against an ideal sine of the fitted amplitude and frequency
    n = pcm.shape[1]
    ideal = (amp * np.sin(2 * np.pi * f * (np.arange(n) / 44100.0))).astype(np.float32)
    got, _ = oracle.analyze_pcm(pcm[0], pcm[-1] if pcm.shape[0] == 2 else None, 44100)
    want, _ = oracle.analyze_pcm(ideal, ideal if pcm.shape[0] == 2 else None, 44100)
    assert abs(got["loudness_db"] - want["loudness_db"]) <= 0.1


def test_fixture_loudness_tracks_the_encoded_level(oracle):
    """The encodes do not carry the sine at one level (libmp3lame scales CBR 128k, VBR -q 2 and the mono encode
    differently: peaks 0.0843 / 0.0887 / 0.1189, the same in ffmpeg's decode of them), so their loudness differs --
    by exactly the level difference: loudness minus 20 log10(peak) is one constant across the three, within 0.1 dB."""
    k = []
    for name in ("test_joint_stereo", "test_vbr", "test_mono"):
        pcm, _ = mp3dec.decode((FIX / f"{name}.mp3").read_bytes())
        res, _ = oracle.analyze_pcm(pcm[0], pcm[1] if pcm.shape[0] == 2 else None, 44100)
        steady = pcm[0, LAME_DELAY + 2000:LAME_DELAY + 42000]
        k.append(res["loudness_db"] - 20 * np.log10(float(np.abs(steady).max())))
    assert max(k) - min(k) <= 0.1, k


def test_damaged_reference_fixture_decodes_like_the_reference_would():
    """tests/fixtures/test_stereo.mp3 was committed with global_gain saturated at 255 in all but its last granule and
    its last frame six bytes short.  A float decoder follows the bits: enormous samples, 39 whole frames (the reader
    cannot fill the 40th: UnexpectedEof ends the reference's loop, src/replaygain.rs:884-888)."""
    data = (FIX / "test_stereo.mp3").read_bytes()
    pcm, info = mp3dec.decode(data)
    assert info.audio_frames == 39 and pcm.shape == (2, 39 * 1152)
    assert np.isfinite(pcm).all() and np.abs(pcm).max() > 1e6


def test_pcm_hashes_of_this_build():
    """sha256 of round(pcm * 2^15) per stream; tests/golden/mp3_pcm_sha256.json: pins the decoder's own output (rewritten when its arithmetic changes on purpose, as with the fused multiply-adds of round 2; correctness is what the ffmpeg goldens above hold)."""
    want = json.loads((ROOT / "tests" / "golden" / "mp3_pcm_sha256.json").read_text())
    got = {}
    for p in STREAMS:
        pcm, _ = mp3dec.decode(p.read_bytes())
        got[p.stem] = hashlib.sha256(np.round(pcm.astype(np.float64) * 32768.0).astype(np.int32).tobytes()).hexdigest()
    assert got == want


# ---- container / packet semantics ------------------------------------------------------------------------------------
def _frames(data: bytes):
    """(offset, size) of every frame of a clean constant-version stream"""
    import mp3_bitstream as B

    out, pos = [], 0
    while pos + 4 <= len(data):
        h = data[pos:pos + 4]
        ver = (h[1] >> 3) & 3
        lsf = ver != 3
        br = (B.BITRATES_V2 if lsf else B.BITRATES_V1)[h[2] >> 4]
        rate = [44100, 48000, 32000][(h[2] >> 2) & 3] >> (0 if ver == 3 else (1 if ver == 2 else 2))
        fb = (72 if lsf else 144) * br * 1000 // rate + ((h[2] >> 1) & 1)
        out.append((pos, fb))
        pos += fb
    return out


def test_id3v2_and_junk_are_skipped():
    body = (GOLD / "v1_44k_stereo_long.mp3").read_bytes()
    ref, _ = mp3dec.decode(body)
    tag = b"ID3\x04\x00\x00" + bytes([0, 0, 2, 0]) + b"\xff\xfb\x90\x64" * 64  # 256 bytes that look like sync words
    pcm, info = mp3dec.decode(tag + body)
    assert info.id3v2_bytes == 266 and info.first_frame_offset == 266 and np.array_equal(pcm, ref)
    pcm, info = mp3dec.decode(b"\x00garbage\xff\xe0" * 7 + body)
    assert info.junk_bytes == 70 and np.array_equal(pcm, ref)
    # junk between two frames: the decoder resynchronises and loses nothing
    fr = _frames(body)
    cut = fr[3][0]
    pcm, info = mp3dec.decode(body[:cut] + b"\x00" * 100 + body[cut:])
    assert info.junk_bytes == 100 and np.array_equal(pcm, ref)


def test_xing_info_frame_is_not_decoded():
    data = (FIX / "test_mono.mp3").read_bytes()
    info = mp3dec.scan(data)
    assert info.info_frame == 1 and info.audio_frames == 40 and info.frames == 46080
    fr = _frames(data[44:])
    assert len(fr) == 41  # the Info frame + 40 audio frames
    # without the Info frame the same PCM comes out
    a, _ = mp3dec.decode(data)
    b, ib = mp3dec.decode(data[44 + fr[0][1]:])
    assert ib.info_frame == 0 and np.array_equal(a, b)


def test_truncated_last_frame_ends_the_track():
    body = (GOLD / "v1_48k_dual_channel.mp3").read_bytes()
    ref, info = mp3dec.decode(body)
    pcm, i2 = mp3dec.decode(body[:-5])
    assert i2.audio_frames == info.audio_frames - 1 and np.array_equal(pcm, ref[:, :pcm.shape[1]])


def test_frame_reaching_behind_the_reservoir_is_dropped():
    """DecodeError -> continue (src/replaygain.rs:896-899): with the first frames cut away, frames whose
    main_data_begin points at bytes that never arrived produce nothing; decoding resumes once the reservoir holds
    enough, and from there on the PCM is the original's (after the filterbank's memory of the gap has passed)."""
    body = (GOLD / "v1_44k_mono_crc_reservoir.mp3").read_bytes()
    ref, info = mp3dec.decode(body)
    fr = _frames(body)
    pcm, i2 = mp3dec.decode(body[fr[2][0]:])
    assert i2.skipped_frames >= 1
    assert i2.audio_frames + i2.skipped_frames == info.audio_frames - 2
    assert pcm.shape[1] == i2.audio_frames * 1152
    tail = 3 * 1152
    assert np.abs(pcm[:, -tail:] - ref[:, -tail:]).max() < 1e-6


def test_errors():
    with pytest.raises(mp3dec.Mp3DecodeError) as e:
        mp3dec.decode(b"")
    assert e.value.code == -2
    with pytest.raises(mp3dec.Mp3DecodeError):
        mp3dec.decode(b"RIFF" + bytes(1000))
    with pytest.raises(mp3dec.Mp3DecodeError):  # Layer II header: not this decoder's (nor the reference build's) codec
        mp3dec.decode(bytes([0xFF, 0xFD, 0x90, 0x00]) * 400)
    # capacity
    import ctypes as C

    data = (GOLD / "v25_8k_mono.mp3").read_bytes()
    buf = (C.c_char * len(data)).from_buffer_copy(data)
    out = np.zeros(100, dtype=np.float32)
    info = mp3dec.StreamInfo()
    rc = mp3dec.lib().rg_mp3_decode_f32(C.cast(buf, C.c_void_p), len(data), out.ctypes.data, None, 100, C.byref(info))
    assert rc == -3 and info.frames == 5760 and b"capacity" in mp3dec.lib().rg_mp3dec_last_error()


def test_damaged_streams_never_crash():
    rng = random.Random(20260929)
    srcs = [p.read_bytes() for p in STREAMS]
    for k in range(400):
        d = bytearray(rng.choice(srcs))
        kind = rng.randrange(4)
        if kind == 0:
            for _ in range(rng.randint(1, 30)):
                d[rng.randrange(len(d))] = rng.randrange(256)
        elif kind == 1:
            d = d[:rng.randrange(len(d))]
        elif kind == 2:
            a = rng.randrange(len(d))
            del d[a:a + rng.randint(1, 600)]
        else:
            a = rng.randrange(len(d))
            d[a:a] = bytes(rng.randrange(256) for _ in range(rng.randint(1, 300)))
        try:
            pcm, info = mp3dec.decode(bytes(d))
        except mp3dec.Mp3DecodeError:
            continue
        assert pcm.shape[1] == info.frames <= mp3dec.scan(bytes(d)).frames


def test_frame_index_agrees_with_the_decoders_on_damaged_streams():
    """The device route decides on the host, from headers and side information alone, which frames decode
    (rg_mp3_index_stream); the one-shot decoder and rg_mp3_parse_units decide it while decoding.  They must agree on
    every input: same number of units, same PCM length, same counts of decoded and dropped frames."""
    rng = random.Random(4242)
    srcs = [p.read_bytes() for p in STREAMS]
    compared = 0
    for k in range(1500):
        d = bytearray(rng.choice(srcs))
        kind = rng.randrange(5)
        if kind == 0:
            for _ in range(rng.randint(1, 40)):
                d[rng.randrange(len(d))] = rng.randrange(256)
        elif kind == 1:
            d = d[:rng.randrange(len(d))]
        elif kind == 2:
            a = rng.randrange(len(d))
            del d[a:a + rng.randint(1, 1200)]
        elif kind == 3:
            a = rng.randrange(len(d))
            d[a:a] = bytes(rng.randrange(256) for _ in range(rng.randint(1, 300)))
        else:  # flip bits inside side information: part2_3_length, big_values, block types, table selects
            for _ in range(rng.randint(1, 6)):
                a = rng.randrange(len(d))
                d[a] ^= 1 << rng.randrange(8)
        d = bytes(d)
        # the default device route: the host strips headers and side information, the device (here: the same source on
        # the CPU, rg_mp3_frame.h) decides frame by frame -- the same main data and records as the host's indexer
        sc = mp3dec.index_selfcheck(d)
        try:
            n_idx, ii = mp3dec.index_units(d)
        except mp3dec.Mp3DecodeError:
            assert sc < 0, k
            with pytest.raises(mp3dec.Mp3DecodeError):
                mp3dec.parse_units(d)
            continue
        assert sc == 0, k
        is_, _, pi = mp3dec.parse_units(d)
        assert (n_idx, ii.frames, ii.audio_frames, ii.skipped_frames) == (is_.shape[0], pi.frames, pi.audio_frames, pi.skipped_frames), k
        compared += 1
    assert compared > 1000


def test_library_exports_every_declared_symbol():
    import ctypes as C

    from mp3rgain_amd import _capi

    txt = re.sub(r"/\*.*?\*/", "", (ROOT / "include" / "mp3rgain_amd_dec.h").read_text(), flags=re.S)
    declared = sorted(set(re.findall(r"\b(rg_mp3[a-z0-9_]*)\s*\(", txt)))
    assert declared == sorted(n for n, _, _ in mp3dec.SYMBOLS)
    raw = C.CDLL(str(_capi.LIB_PATH))
    for name in declared:
        assert hasattr(raw, name)
    assert C.sizeof(mp3dec.StreamInfo) == 56
