"""File-level layer, CPU side: the WAV container parser of include/mp3rgain_amd.h (rg_wav_parse) --
no GPU, no compute."""
import ctypes as C
import struct
import sys
from pathlib import Path

import numpy as np
import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent))
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

from wavutil import wav_bytes  # noqa: E402

from mp3rgain_amd import _capi  # noqa: E402


def parse(b: bytes):
    w = _capi.WavInfo()
    rc = _capi.load().rg_wav_parse(b, len(b), C.byref(w))
    return rc, w


@pytest.mark.parametrize("kind,bits,tag", [("u8", 8, 1), ("s16", 16, 1), ("s24", 24, 1), ("s32", 32, 1), ("f32", 32, 3)])
@pytest.mark.parametrize("nch", [1, 2, 6])
@pytest.mark.parametrize("extensible", [False, True])
def test_wav_parse_formats(kind, bits, tag, nch, extensible):
    frames = 101
    chans = [np.arange(frames) % 50 for _ in range(nch)]
    b = wav_bytes(chans, 48000, kind, extensible=extensible)
    rc, w = parse(b)
    assert rc == 0
    assert (w.sample_rate, w.channels, w.bits_per_sample, w.sample_format) == (48000, nch, bits, tag)
    assert w.block_align == nch * bits // 8 and w.frames == frames
    assert b[w.data_offset - 8:w.data_offset - 4] == b"data"


def test_wav_parse_streamed_and_truncated():
    chans = [np.arange(1000), np.arange(1000)]
    b = wav_bytes(chans, 44100, "s16", streamed=True)
    rc, w = parse(b)
    assert rc == 0 and w.frames == 1000
    # a data chunk that claims more than is there is cut to whole frames
    full = wav_bytes(chans, 44100, "s16", extra_chunks=False)
    rc, w = parse(full[:-7])
    assert rc == 0 and w.frames == 1000 - 2
    rc, w = parse(full[:44])
    assert rc == 0 and w.frames == 0


@pytest.mark.parametrize("bad", [
    b"", b"RIFF", b"RIFF\0\0\0\0WAVX" + bytes(40), b"ID3\x03" + bytes(60),
    b"RIFF" + struct.pack("<I", 4) + b"WAVE",                                             # no chunks
    b"RIFF" + struct.pack("<I", 20) + b"WAVE" + b"data" + struct.pack("<I", 8) + bytes(8),  # data before fmt
    b"RIFF" + struct.pack("<I", 20) + b"WAVE" + b"fmt " + struct.pack("<I", 8) + bytes(8),  # short fmt
])
def test_wav_parse_rejects(bad):
    rc, _ = parse(bad)
    assert rc == _capi.load().rg_wav_parse(b"x", 1, C.byref(_capi.WavInfo())) != 0


def test_wav_parse_rejects_inconsistent_block_align():
    b = bytearray(wav_bytes([np.arange(10)], 44100, "s16", extra_chunks=False))
    struct.pack_into("<H", b, 32, 3)  # block_align
    rc, _ = parse(bytes(b))
    assert rc != 0
    b = bytearray(wav_bytes([np.arange(10)], 44100, "s16", extra_chunks=False))
    struct.pack_into("<H", b, 22, 0)  # channels
    rc, _ = parse(bytes(b))
    assert rc != 0


def test_wavinfo_layout():
    assert C.sizeof(_capi.WavInfo) == 32


def test_wav_parse_survives_damage():
    """2000 damaged WAV headers: never a crash, and whatever parses describes bytes that exist."""
    import random

    rng = random.Random(99)
    bases = [wav_bytes([np.arange(50) % 7, np.arange(50) % 5], 44100, k, extensible=e) for k in ("u8", "s16", "s24", "f32") for e in (False, True)]
    ok = 0
    for _ in range(2000):
        b = bytearray(rng.choice(bases))
        for _ in range(rng.randint(1, 4)):
            kind = rng.randint(0, 3)
            if len(b) == 0:
                break
            pos = rng.randrange(0, len(b))
            if kind == 0:
                b[pos] = rng.getrandbits(8)
            elif kind == 1 and len(b) >= 4:
                struct.pack_into("<I", b, min(pos, len(b) - 4), rng.choice([0, 1, 0xFFFFFFFF, 0x7FFFFFFF, rng.getrandbits(32), rng.randrange(100)]))
            elif kind == 2:
                del b[pos:]
            else:
                b[pos:pos] = bytes(rng.getrandbits(8) for _ in range(rng.randint(1, 40)))
        if len(b) == 0:
            b = bytearray(b"\0")
        rc, w = parse(bytes(b))
        if rc == 0:
            ok += 1
            assert w.channels > 0 and w.block_align == w.channels * (w.bits_per_sample // 8)
            assert w.data_offset + w.frames * w.block_align <= len(b)
    assert 100 < ok < 2000
