"""Test helper: build RIFF/WAVE byte strings from planar numpy channels (not part of the product)."""
import struct

import numpy as np


def wav_bytes(channels, rate, kind, extensible=False, extra_chunks=True, streamed=False):
    """kind: 'u8' | 's16' | 's24' | 's32' | 'f32'; channels: equal-length 1-D arrays already in that value range
    (u8: 0..255 ints, s16/s24/s32: signed ints, f32: float32)."""
    nch, n = len(channels), len(channels[0])
    bits = {"u8": 8, "s16": 16, "s24": 24, "s32": 32, "f32": 32}[kind]
    tag = 3 if kind == "f32" else 1
    inter = np.stack(channels, axis=1).reshape(-1)
    if kind == "u8":
        body = inter.astype(np.uint8).tobytes()
    elif kind == "s16":
        body = inter.astype("<i2").tobytes()
    elif kind == "s32":
        body = inter.astype("<i4").tobytes()
    elif kind == "f32":
        body = inter.astype("<f4").tobytes()
    else:
        b = inter.astype("<i4").view(np.uint8).reshape(-1, 4)[:, :3]
        body = np.ascontiguousarray(b).tobytes()
    align = nch * bits // 8
    if extensible:
        guid_tail = bytes.fromhex("000000001000800000aa00389b71")
        fmt = struct.pack("<HHIIHHHHIH", 0xFFFE, nch, rate, rate * align, align, bits, 22, bits, 0, tag) + guid_tail
    else:
        fmt = struct.pack("<HHIIHH", tag, nch, rate, rate * align, align, bits)
    chunks = b"fmt " + struct.pack("<I", len(fmt)) + fmt
    if extra_chunks:
        chunks += b"LIST" + struct.pack("<I", 5) + b"abcde" + b"\0"  # odd-sized chunk + pad byte
    size = 0xFFFFFFFF if streamed else len(body)
    chunks += b"data" + struct.pack("<I", size) + body
    if extra_chunks and not streamed:
        chunks += (b"\0" if len(body) & 1 else b"") + b"id3 " + struct.pack("<I", 4) + b"tag!"
    riff_size = 0xFFFFFFFF if streamed else 4 + len(chunks)
    return b"RIFF" + struct.pack("<I", riff_size) + b"WAVE" + chunks


def planar_for_oracle(channels, kind):
    """The planar arrays the library's de-interleave produces for a WAV of this kind."""
    if kind == "u8":
        return [((np.asarray(c).astype(np.int32) - 128) * 256).astype(np.int16) for c in channels]
    if kind == "s16":
        return [np.asarray(c).astype(np.int16) for c in channels]
    if kind == "s24":
        return [(np.asarray(c).astype(np.int64) << 8).astype(np.int32) for c in channels]
    if kind == "s32":
        return [np.asarray(c).astype(np.int32) for c in channels]
    return [np.asarray(c).astype(np.float32) for c in channels]


def test_signal(kind, rate, frames, nch, seed):
    """Deterministic music-like test content in the value range of `kind`."""
    rng = np.random.default_rng(seed)
    t = np.arange(frames) / rate
    out = []
    for c in range(nch):
        x = 0.25 * np.sin(2 * np.pi * (220.0 * (c + 1)) * t) + 0.1 * rng.standard_normal(frames) * (0.2 + np.abs(np.sin(1.3 * t + c)))
        x = np.clip(x, -1.0, 1.0)
        if kind == "f32":
            out.append(x.astype(np.float32))
        elif kind == "u8":
            out.append(np.clip(np.round(x * 127) + 128, 0, 255).astype(np.int32))
        else:
            full = {"s16": 32767, "s24": 8388607, "s32": 2147483647}[kind]
            out.append(np.round(x * full).astype(np.int64))
    return out
