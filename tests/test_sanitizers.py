"""Host byte parsers under AddressSanitizer + UndefinedBehaviorSanitizer: the container walkers (rg_demux.cpp) and the host MP3
decoder (rg_mp3dec.cpp: scanner, one-shot decoder, unit parser, the two frame indexers) are built with gcc's sanitizers
together with small drivers and fed a few thousand damaged files as exact-size heap buffers -- any read past a buffer, signed
overflow or misaligned access aborts the driver.  (The ctypes tests would not notice an over-read that happens to stay inside
mapped memory.)"""
import random
import shutil
import struct
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "oracle"))
import mp4demux_oracle as M  # noqa: E402


def _build(tmp_path_factory, name, sources, extra=()):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    out = tmp_path_factory.mktemp("san") / name
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", f"-I{ROOT / 'include'}",
           *extra, *[str(s) for s in sources], "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        if "sanitize" in r.stderr or "asan" in r.stderr.lower():
            pytest.skip("this toolchain has no sanitizer runtime")
        raise AssertionError(r.stderr)
    return out


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    return _build(tmp_path_factory, "demux_driver", [ROOT / "tests" / "san" / "demux_driver.cpp", ROOT / "mp3rgain_amd" / "csrc" / "rg_demux.cpp"])


@pytest.fixture(scope="module")
def mp3_driver(tmp_path_factory):
    csrc = ROOT / "mp3rgain_amd" / "csrc"
    import platform

    fma = ["-mfma"] if platform.machine() == "x86_64" else []
    return _build(tmp_path_factory, "mp3dec_driver", [ROOT / "tests" / "san" / "mp3dec_driver.cpp", csrc / "rg_mp3dec.cpp", csrc / "rg_mp3gain.cpp"],
                  extra=fma + ["-ffp-contract=off"])


def test_mp3_decoder_under_asan_ubsan(mp3_driver, tmp_path):
    """2000 mutated, truncated and spliced versions of the golden streams (the dense encodes included) through the host decoder
    and the frame indexers the device routes rely on."""
    rng = random.Random(7)
    srcs = [p.read_bytes() for p in sorted((ROOT / "tests" / "golden" / "mp3").glob("*.mp3")) + sorted((ROOT / "tests" / "golden" / "fixtures").glob("*.mp3"))
            if p.stat().st_size < 80000]
    files = []
    for k in range(2000):
        d = bytearray(rng.choice(srcs))
        kind = rng.randrange(4)
        if kind == 0:
            for _ in range(rng.randint(1, 30)):
                d[rng.randrange(len(d))] = rng.randrange(256)
        elif kind == 1:
            d = d[:rng.randrange(4, len(d))]
        elif kind == 2:
            a = rng.randrange(len(d))
            del d[a:a + rng.randint(1, 1000)]
        else:
            a = rng.randrange(len(d))
            d[a:a] = bytes(rng.randrange(256) for _ in range(rng.randint(1, 200)))
        f = tmp_path / f"m{k:04d}.mp3"
        f.write_bytes(bytes(d))
        files.append(str(f))
    for lo in range(0, len(files), 250):
        r = subprocess.run([str(mp3_driver)] + files[lo:lo + 250], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]


def test_container_walkers_under_asan_ubsan(driver, tmp_path):
    rng = random.Random(2026)

    def samples(n, lo=3, hi=300):
        return [bytes(rng.randrange(256) for _ in range(rng.randint(lo, hi))) for _ in range(n)]

    bases = [
        M.build_mp4([M.Track("video", samples(4)), M.Track("aac", samples(30), per_chunk=(3, 5)), M.Track("mp3", samples(12), co64=True)]),
        M.build_mp4([M.Track("aac", samples(25, 1, 15), stz2=4), M.Track("aac_mpeg2", samples(10), stz2=16, mdhd_v1=True, long_descriptors=True)], moov_first=False),
        M.build_mp4([M.Track("aac", samples(9, 40, 41)[:9], fixed_size=False, entry_version=1), M.Track("mp3_qt", samples(7))], brand=b"mp42"),
        b"".join(M.adts_frame(p, crc=(i % 2 == 0)) for i, p in enumerate(samples(12, 10, 90))),
        b"ID3\x04\x00\x00\x00\x00\x01\x00" + bytes(128) + b"".join(M.adts_frame(p, rate=22050, channels=1) for p in samples(6, 10, 60)),
    ]
    files = []
    for k in range(2500):
        d = bytearray(rng.choice(bases))
        kind = rng.randrange(5)
        if kind == 0:
            for _ in range(rng.randint(1, 10)):
                d[rng.randrange(len(d))] = rng.randrange(256)
        elif kind == 1:
            d = d[:rng.randrange(len(d) + 1)]
        elif kind == 2:
            at = rng.randrange(max(1, len(d) - 4))
            d[at:at + 4] = struct.pack(">I", rng.choice([0, 1, 7, 8, 0xFFFFFFFF, 0x7FFFFFFF, 0x80000000, rng.randrange(1 << 32)]))
        elif kind == 3:
            a = rng.randrange(len(d))
            del d[a:a + rng.randint(1, 300)]
        else:
            at = rng.randrange(max(1, len(d) - 8))
            d[at:at + 8] = struct.pack(">Q", rng.choice([0, 1, (1 << 64) - 1, 1 << 63, rng.randrange(1 << 64)]))
        f = tmp_path / f"c{k:04d}.bin"
        f.write_bytes(bytes(d))
        files.append(str(f))
    files += [str(tmp_path / "missing.bin")]
    (tmp_path / "empty.bin").write_bytes(b"")
    files.append(str(tmp_path / "empty.bin"))
    for lo in range(0, len(files), 500):
        r = subprocess.run([str(driver)] + files[lo:lo + 500], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]


@pytest.fixture(scope="module")
def bytes_driver(tmp_path_factory):
    csrc = ROOT / "mp3rgain_amd" / "csrc"
    return _build(tmp_path_factory, "bytes_driver", [ROOT / "tests" / "san" / "bytes_driver.cpp", csrc / "rg_mp3gain.cpp", csrc / "rg_mp4meta.cpp"])


def test_gain_patcher_ape_and_mp4_tags_under_asan_ubsan(bytes_driver, tmp_path):
    """The byte-level rows either side of the path (global_gain scanner / patcher, APEv2 reader, MP4 ReplayGain tag reader and
    writer) on 3000 damaged MP3 and MP4 files."""
    sys.path.insert(0, str(ROOT / "tests"))
    from test_mp4meta import make_mp4

    rng = random.Random(99)
    srcs = [p.read_bytes() for p in sorted((ROOT / "tests" / "golden" / "fixtures").glob("*.mp3"))]
    srcs += [(ROOT / "tests" / "golden" / "mp3" / n).read_bytes() for n in ("v1_44k_mono_crc_reservoir.mp3", "v2_22k_stereo.mp3", "v25_8k_mono.mp3")]
    # an APEv2 tag behind one of them (the undo record the patcher writes)
    ape_items = b"".join(struct.pack("<II", len(v), 0) + k + b"\0" + v for k, v in ((b"MP3GAIN_UNDO", b"+003,+003,N"), (b"MP3GAIN_MINMAX", b"110,210")))
    footer = b"APETAGEX" + struct.pack("<IIII", 2000, len(ape_items) + 32, 2, 0) + bytes(8)
    srcs.append(srcs[0] + ape_items + footer)
    for layout in ("moov_first", "mdat_first"):
        for udta in ("full", "meta_no_ilst", "empty", "none"):
            srcs.append(make_mp4(layout=layout, udta=udta, wide=(udta == "full"), old_rg={"replaygain_track_gain": "-1.00 dB"} if udta == "full" else None)[0])
    files = []
    for k in range(3000):
        d = bytearray(rng.choice(srcs))
        kind = rng.randrange(4)
        if kind == 0:
            for _ in range(rng.randint(1, 12)):
                d[rng.randrange(len(d))] = rng.randrange(256)
        elif kind == 1:
            d = d[:rng.randrange(len(d) + 1)]
        elif kind == 2:
            at = rng.randrange(max(1, len(d) - 4))
            d[at:at + 4] = struct.pack(rng.choice([">I", "<I"]), rng.choice([0, 1, 7, 8, 0xFFFFFFFF, 0x7FFFFFFF, 0x80000000, rng.randrange(1 << 32)]))
        else:
            a = rng.randrange(len(d))
            del d[a:a + rng.randint(1, 200)]
        f = tmp_path / f"b{k:04d}.bin"
        f.write_bytes(bytes(d))
        files.append(str(f))
    for lo in range(0, len(files), 500):
        r = subprocess.run([str(bytes_driver)] + files[lo:lo + 500], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
