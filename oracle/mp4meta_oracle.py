"""CPU restatement of mp3rgain's MP4/M4A ReplayGain tag module (src/mp4meta.rs, v1.5.0) -- TEST INFRASTRUCTURE ONLY.

Only tests/ may import this file; nothing under mp3rgain_amd/ does.  It is a plain-Python statement of what
the reference does to the bytes of an MP4 file, written from the behaviour of src/mp4meta.rs (each function
cites the lines it follows), and it is what tests/test_mp4meta.py compares the C++ implementation
(mp3rgain_amd/csrc/rg_mp4meta.cpp) against, byte for byte.

Pinning: the reference's own unit tests for this module (src/mp4meta.rs:895-943: freeform round trip, the
"{:+.2} dB" / "{:.6}" strings for (3.5, 0.98765) and (2.0, 0.99999), the ftyp brand check) are restated in
tests/test_mp4meta.py against this oracle and against the C++ code.  The reference holds no MP4 fixture file,
so whole-file behaviour is pinned only by those unit tests plus structural checks (box sizes consistent,
chunk offsets still pointing at the same media bytes): "parity partially unpinned".
"""
from __future__ import annotations

import struct
from typing import List, Optional, Tuple

ITUNES = "com.apple.iTunes"                                   # :32
KEYS = ("replaygain_track_gain", "replaygain_track_peak",      # :26-29
        "replaygain_album_gain", "replaygain_album_peak")


class Tags:
    """ReplayGainTags (:113-177): four optional strings, in the order the writer emits them."""

    def __init__(self, track_gain=None, track_peak=None, album_gain=None, album_peak=None):
        self.v: List[Optional[str]] = [track_gain, track_peak, album_gain, album_peak]

    def set_track(self, gain_db: float, peak: float):          # :126-129
        self.v[0], self.v[1] = "%+.2f dB" % gain_db, "%.6f" % peak

    def set_album(self, gain_db: float, peak: float):          # :131-134
        self.v[2], self.v[3] = "%+.2f dB" % gain_db, "%.6f" % peak

    def is_empty(self) -> bool:                                # :136-141
        return all(x is None for x in self.v)

    def __eq__(self, o):
        return self.v == o.v

    def __repr__(self):
        return "Tags(%r)" % (self.v,)


def header(data: bytes, pos: int) -> Optional[Tuple[int, bytes, int]]:
    """BoxHeader::read (:59-88) at `pos`: (size, fourcc, header_size) or None when it does not fit."""
    if pos < 0 or pos + 8 > len(data):
        return None
    size, typ = struct.unpack_from(">I4s", data, pos)
    if size == 1:
        if pos + 16 > len(data):
            return None
        size = struct.unpack_from(">Q", data, pos + 8)[0]
        # a box smaller than its own header: the reference's `size - header_size` (:90-96) underflows
        # (panic in debug, wrapped length in release); the restatement and the library call it "no box"
        return None if 0 < size < 16 else (size, typ, 16)
    return None if 0 < size < 8 else (size, typ, 8)


def content_size(size: int, hdr: int) -> int:                  # :90-96
    return 0 if size == 0 else size - hdr


def find_box(data: bytes, typ: bytes):                         # :180-203
    pos = 0
    while True:
        h = header(data, pos)
        if h is None:
            return None
        size, t, hdr = h
        if t == typ:
            return pos, size, hdr
        if size == 0:
            return None
        nxt = pos + size
        if nxt >= len(data):
            return None
        pos = nxt


def find_in(data: bytes, start: int, size: int, typ: bytes):   # :206-233
    end = start + size
    pos = start
    while pos + 8 <= end:
        h = header(data, pos)
        if h is None:
            return None
        bsize, t, hdr = h
        if t == typ:
            return pos, bsize, hdr
        if bsize == 0:
            return None
        pos += bsize
    return None


def parse_freeform(data: bytes):                               # :236-291
    ns = name = value = None
    pos = 0
    while True:
        h = header(data, pos)
        if h is None:
            break
        size, t, hdr = h
        cstart = pos + hdr
        csize = content_size(size, hdr)
        if csize < 0 or cstart + csize > len(data):
            break
        cend = cstart + csize
        if t in (b"mean", b"name") and cstart + 4 < cend:
            s = data[cstart + 4:cend].decode("utf-8", "replace")
            if t == b"mean":
                ns = s
            else:
                name = s
        elif t == b"data" and cstart + 8 < cend:
            value = data[cstart + 8:cend].decode("utf-8", "replace")
        pos = cend
    if ns is None or name is None or value is None:
        return None
    return ns, name, value


def serialize_freeform(ns: str, name: str, value: str) -> bytes:  # :294-330
    n, m, v = ns.encode(), name.encode(), value.encode()
    inner = (struct.pack(">I4sI", 12 + len(n), b"mean", 0) + n +
             struct.pack(">I4sI", 12 + len(m), b"name", 0) + m +
             struct.pack(">I4sII", 16 + len(v), b"data", 0, 1) + v)
    return struct.pack(">I4s", 8 + len(inner), b"----") + inner


def key_index(parsed) -> int:
    if parsed is None or parsed[0] != ITUNES:
        return -1
    low = parsed[1].encode("utf-8", "replace").lower()  # bytes.lower() folds ASCII only: eq_ignore_ascii_case
    for i, k in enumerate(KEYS):
        if low == k.encode():
            return i
    return -1


def locate(data: bytes):
    """moov -> udta -> meta (+4) -> ilst, as both the reader (:339-377) and the writer (:554-601) walk it."""
    r = {"moov": None, "udta": None, "meta": None, "ilst": None}
    r["moov"] = find_box(data, b"moov")
    if r["moov"] is None:
        return r
    p, s, h = r["moov"]
    r["udta"] = find_in(data, p + h, content_size(s, h), b"udta")
    if r["udta"] is None:
        return r
    p, s, h = r["udta"]
    r["meta"] = find_in(data, p + h, content_size(s, h), b"meta")
    if r["meta"] is None:
        return r
    p, s, h = r["meta"]
    r["ilst"] = find_in(data, p + h + 4, max(0, content_size(s, h) - 4), b"ilst")
    return r


def read_tags(data: bytes) -> Tags:                            # :333-417
    tags = Tags()
    loc = locate(data)
    if loc["ilst"] is None:
        return tags
    p, s, h = loc["ilst"]
    pos, end = p + h, p + h + content_size(s, h)
    while pos + 8 <= end:
        hd = header(data, pos)
        if hd is None:
            break
        size, t, hdr = hd
        if t == b"----" and hdr <= size <= len(data) - pos:
            parsed = parse_freeform(data[pos + hdr:pos + size])
            k = key_index(parsed)
            if k >= 0:
                tags.v[k] = parsed[2]
        if size == 0:
            break
        pos += size
    return tags


def build_ilst(tags: Tags, existing: bytes) -> bytes:          # :621-675
    content = b""
    pos = 0
    while pos + 8 <= len(existing):
        hd = header(existing, pos)
        if hd is None:
            break
        size, t, hdr = hd
        if size == 0 or pos + size > len(existing):
            break
        is_rg = t == b"----" and size >= hdr and key_index(parse_freeform(existing[pos + hdr:pos + size])) >= 0
        if not is_rg:
            content += existing[pos:pos + size]
        pos += size
    for k, v in zip(KEYS, tags.v):
        if v is not None:
            content += serialize_freeform(ITUNES, k, v)
    return struct.pack(">I4s", 8 + len(content), b"ilst") + content


def build_meta(ilst: bytes) -> bytes:                          # :677-716
    hdlr_body = b"\0" * 8 + b"mdir" + b"appl" + b"\0" * 9
    hdlr = struct.pack(">I4s", 8 + len(hdlr_body), b"hdlr") + hdlr_body
    body = b"\0" * 4 + hdlr + ilst
    return struct.pack(">I4s", 8 + len(body), b"meta") + body


def grow(buf: bytearray, pos: int, diff: int):                 # :728-747
    if pos + 4 > len(buf):
        return
    cur = struct.unpack_from(">I", buf, pos)[0]
    if cur <= 1:
        return
    struct.pack_into(">I", buf, pos, (cur + diff) & 0xFFFFFFFF)


def shift_offsets(buf: bytearray, start: int, end: int, diff: int):  # :772-863
    pos = start
    while pos + 8 <= end and pos + 8 <= len(buf):
        size, t = struct.unpack_from(">I4s", buf, pos)
        if size == 0 or pos + size > end:
            break
        if t in (b"stco", b"co64"):
            cp = pos + 12
            if cp + 4 <= len(buf):
                n = struct.unpack_from(">I", buf, cp)[0]
                w, fmt, mask = (4, ">I", 0xFFFFFFFF) if t == b"stco" else (8, ">Q", 0xFFFFFFFFFFFFFFFF)
                p = cp + 4
                for _ in range(n):
                    if p + w > len(buf):
                        break
                    struct.pack_into(fmt, buf, p, (struct.unpack_from(fmt, buf, p)[0] + diff) & mask)
                    p += w
        elif t in (b"trak", b"mdia", b"minf", b"stbl", b"moov", b"udta"):
            shift_offsets(buf, pos + 8, pos + size, diff)
        pos += size


def update(data: bytes, tags: Tags) -> bytes:                  # :433-531
    loc = locate(data)
    if loc["moov"] is None:
        raise ValueError("No moov box found in MP4 file")
    mp, ms, mh = loc["moov"]
    moov_end = mp + ms
    if loc["ilst"] is not None:
        ip, isz, ih = loc["ilst"]
        new = build_ilst(tags, data[ip + ih:ip + ih + content_size(isz, ih)])
        diff = len(new) - isz
        out = bytearray(data[:ip] + new + data[ip + isz:])
        grow(out, mp, diff)
        grow(out, loc["udta"][0], diff)
        grow(out, loc["meta"][0], diff)
    elif loc["udta"] is not None:
        up, us, _ = loc["udta"]
        meta = build_meta(build_ilst(tags, b""))
        out = bytearray(data[:up + us] + meta + data[up + us:])
        grow(out, mp, len(meta))
        grow(out, up, len(meta))
    else:
        meta = build_meta(build_ilst(tags, b""))
        udta = struct.pack(">I4s", 8 + len(meta), b"udta") + meta
        out = bytearray(data[:moov_end] + udta + data[moov_end:])
        grow(out, mp, len(udta))
    mdat = find_box(data, b"mdat")
    if mdat is not None and mdat[0] > mp:
        diff = len(out) - len(data)
        m2 = find_box(bytes(out), b"moov")
        if diff != 0 and m2 is not None:
            shift_offsets(out, mp + 8, mp + m2[1], diff)
    return bytes(out)


def is_mp4(data: bytes) -> bool:                               # :872-889
    if len(data) < 12:
        return False
    size = struct.unpack_from(">I", data, 0)[0]
    return data[4:8] == b"ftyp" and size >= 12 and data[8:12] in (
        b"M4A ", b"M4B ", b"M4P ", b"M4V ", b"mp41", b"mp42", b"isom", b"iso2")
