"""ISO base media (MP4 / M4A) sample tables and ADTS -- test infrastructure only (nothing under mp3rgain_amd/ imports it).

Two halves, both independent of mp3rgain_amd/csrc/rg_demux.cpp:
  * a WRITER of synthetic containers (ISO/IEC 14496-12: ftyp, moov/mvhd, trak/tkhd, mdia/mdhd/hdlr, minf/stbl with
    stsd [mp4a + esds | .mp3 | alac | avc1], stts, stsc, stsz | stz2, stco | co64; mdat) from lists of access units, and of
    ADTS streams (14496-3 1.A.2) -- the reference ships no M4A fixture and none of its tests touches a container;
  * a READER restating the same clauses in Python, which the C++ demuxer is compared with.
What symphonia (the reference's demuxer, not in its tree) does beyond the standard's text is [unverified]."""
from __future__ import annotations

import struct
from typing import List, Optional, Sequence, Tuple

ASC_RATES = [96000, 88200, 64000, 48000, 44100, 32000, 24000, 22050, 16000, 12000, 11025, 8000, 7350]


def box(typ: bytes, body: bytes, large: bool = False) -> bytes:
    if large:
        return struct.pack(">I4sQ", 1, typ, 16 + len(body)) + body
    return struct.pack(">I4s", 8 + len(body), typ) + body


def full(typ: bytes, version: int, flags: int, body: bytes) -> bytes:
    return box(typ, struct.pack(">I", (version << 24) | flags) + body)


def descriptor(tag: int, body: bytes, long_form: bool = False) -> bytes:
    n = len(body)
    if long_form:  # four 7-bit groups, as many muxers write
        ln = bytes([0x80 | ((n >> 21) & 0x7F), 0x80 | ((n >> 14) & 0x7F), 0x80 | ((n >> 7) & 0x7F), n & 0x7F])
    else:
        assert n < 128
        ln = bytes([n])
    return bytes([tag]) + ln + body


def audio_specific_config(aot: int, rate: int, channels: int) -> bytes:
    bits = ""
    bits += format(aot, "05b") if aot < 31 else "11111" + format(aot - 32, "06b")
    if rate in ASC_RATES:
        bits += format(ASC_RATES.index(rate), "04b")
    else:
        bits += "1111" + format(rate, "024b")
    bits += format(channels, "04b") + "000"
    bits += "0" * (-len(bits) % 8)
    return bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8))


def esds(oti: int, asc: Optional[bytes], long_form: bool = False) -> bytes:
    dsi = descriptor(0x05, asc, long_form) if asc is not None else b""
    dcd = descriptor(0x04, bytes([oti, 0x15]) + b"\0\0\0" + struct.pack(">II", 128000, 128000) + dsi, long_form)
    es = descriptor(0x03, struct.pack(">HB", 1, 0) + dcd + descriptor(0x06, b"\x02"), long_form)
    return full(b"esds", 0, 0, es)


class Track:
    """kind: 'aac' (mp4a, object type 0x40), 'aac_mpeg2' (0x67), 'mp3' (mp4a, 0x6B), 'mp3_qt' (.mp3 entry), 'alac', 'video'"""

    def __init__(self, kind: str, samples: Sequence[bytes], rate: int = 44100, channels: int = 2, per_chunk: Sequence[int] = (4,),
                 co64: bool = False, stz2: int = 0, fixed_size: bool = False, mdhd_v1: bool = False, asc_rate: Optional[int] = None,
                 asc_channels: Optional[int] = None, long_descriptors: bool = False, entry_version: int = 0):
        self.kind, self.samples, self.rate, self.channels = kind, list(samples), rate, channels
        self.per_chunk, self.co64, self.stz2, self.fixed_size, self.mdhd_v1 = list(per_chunk), co64, stz2, fixed_size, mdhd_v1
        self.asc_rate, self.asc_channels, self.long_descriptors, self.entry_version = asc_rate, asc_channels, long_descriptors, entry_version

    def chunks(self) -> List[List[bytes]]:
        out, i, k = [], 0, 0
        while i < len(self.samples):
            n = self.per_chunk[min(k, len(self.per_chunk) - 1)]
            out.append(self.samples[i:i + n])
            i += n
            k += 1
        return out


def _sample_entry(t: Track) -> bytes:
    if t.kind == "video":
        return box(b"avc1", b"\0" * 6 + struct.pack(">H", 1) + b"\0" * 70)
    sound = b"\0" * 6 + struct.pack(">H", 1) + struct.pack(">HHIHHHHI", t.entry_version, 0, 0, t.channels, 16, 0, 0, (t.rate & 0xFFFF) << 16)
    if t.entry_version == 1:
        sound += struct.pack(">IIII", 1024, 0, 0, 2)
    if t.kind == "alac":
        return box(b"alac", sound + full(b"alac", 0, 0, b"\0" * 24))
    if t.kind == "mp3_qt":
        return box(b".mp3", sound)
    oti = {"aac": 0x40, "aac_mpeg2": 0x67, "mp3": 0x6B, "mp3_mpeg2": 0x69}[t.kind]
    asc = audio_specific_config(2, t.asc_rate or t.rate, t.asc_channels or t.channels) if t.kind.startswith("aac") else None
    return box(b"mp4a", sound + esds(oti, asc, t.long_descriptors))


def _stbl(t: Track, chunk_offsets: List[int]) -> bytes:
    stsd = full(b"stsd", 0, 0, struct.pack(">I", 1) + _sample_entry(t))
    stts = full(b"stts", 0, 0, struct.pack(">III", 1, len(t.samples), 1024))
    ch = t.chunks()
    runs = []
    for i, c in enumerate(ch):
        if not runs or runs[-1][1] != len(c):
            runs.append((i + 1, len(c)))
    stsc = full(b"stsc", 0, 0, struct.pack(">I", len(runs)) + b"".join(struct.pack(">III", f, n, 1) for f, n in runs))
    sizes = [len(s) for s in t.samples]
    if t.fixed_size:
        assert len(set(sizes)) <= 1
        stsz = full(b"stsz", 0, 0, struct.pack(">II", sizes[0] if sizes else 0, len(sizes)))
    elif t.stz2:
        if t.stz2 == 16:
            tab = b"".join(struct.pack(">H", s) for s in sizes)
        elif t.stz2 == 8:
            tab = bytes(sizes)
        else:
            nib = sizes + [0] * (len(sizes) & 1)
            tab = bytes((nib[i] << 4) | nib[i + 1] for i in range(0, len(nib), 2))
        stsz = full(b"stz2", 0, 0, struct.pack(">II", t.stz2, len(sizes)) + tab)
    else:
        stsz = full(b"stsz", 0, 0, struct.pack(">II", 0, len(sizes)) + b"".join(struct.pack(">I", s) for s in sizes))
    if t.co64:
        stco = full(b"co64", 0, 0, struct.pack(">I", len(chunk_offsets)) + b"".join(struct.pack(">Q", o) for o in chunk_offsets))
    else:
        stco = full(b"stco", 0, 0, struct.pack(">I", len(chunk_offsets)) + b"".join(struct.pack(">I", o) for o in chunk_offsets))
    return box(b"stbl", stsd + stts + stsc + stsz + stco)


def _trak(t: Track, track_id: int, chunk_offsets: List[int]) -> bytes:
    tkhd = full(b"tkhd", 0, 7, struct.pack(">IIII", 0, 0, track_id, 0) + b"\0" * 64)
    dur = len(t.samples) * 1024
    mdhd = full(b"mdhd", 1, 0, struct.pack(">QQIQ", 0, 0, t.rate, dur) + b"\x55\xc4\0\0") if t.mdhd_v1 else \
        full(b"mdhd", 0, 0, struct.pack(">IIII", 0, 0, t.rate, dur) + b"\x55\xc4\0\0")
    hdlr = full(b"hdlr", 0, 0, struct.pack(">I4s", 0, b"vide" if t.kind == "video" else b"soun") + b"\0" * 12 + b"handler\0")
    mh = full(b"vmhd", 0, 1, b"\0" * 8) if t.kind == "video" else full(b"smhd", 0, 0, b"\0" * 4)
    minf = box(b"minf", mh + box(b"dinf", full(b"dref", 0, 0, struct.pack(">I", 1) + full(b"url ", 0, 1, b""))) + _stbl(t, chunk_offsets))
    return box(b"trak", tkhd + box(b"mdia", mdhd + hdlr + minf))


def build_mp4(tracks: Sequence[Track], moov_first: bool = True, brand: bytes = b"M4A ", interleave: bool = True,
              extra_moov_children: bytes = b"") -> bytes:
    """Chunks of all tracks interleaved round robin in the mdat (as a muxer does) or track after track."""
    ftyp = box(b"ftyp", brand + struct.pack(">I", 0) + b"isomiso2")
    chunks = [t.chunks() for t in tracks]

    def layout(base: int):
        offs = [[] for _ in tracks]
        payload = bytearray()
        order = []
        if interleave:
            for k in range(max((len(c) for c in chunks), default=0)):
                order += [(ti, k) for ti in range(len(tracks)) if k < len(chunks[ti])]
        else:
            order = [(ti, k) for ti in range(len(tracks)) for k in range(len(chunks[ti]))]
        for ti, k in order:
            offs[ti].append(base + len(payload))
            for smp in chunks[ti][k]:
                payload += smp
        return offs, bytes(payload)

    def moov_for(offs):
        mvhd = full(b"mvhd", 0, 0, struct.pack(">IIII", 0, 0, 1000, 0) + b"\0" * 80)
        return box(b"moov", mvhd + b"".join(_trak(t, i + 1, offs[i]) for i, t in enumerate(tracks)) + extra_moov_children)

    if moov_first:
        probe = moov_for(layout(0)[0])
        base = len(ftyp) + len(probe) + 8
        offs, payload = layout(base)
        moov = moov_for(offs)
        assert len(moov) == len(probe)
        return ftyp + moov + box(b"mdat", payload)
    offs, payload = layout(len(ftyp) + 8)
    return ftyp + box(b"mdat", payload) + moov_for(offs)


def adts_frame(payload: bytes, rate: int = 44100, channels: int = 2, profile: int = 1, crc: bool = False, mpeg2: bool = False, blocks: int = 1) -> bytes:
    hdr_len = 9 if crc else 7
    n = hdr_len + len(payload)
    fi = ASC_RATES.index(rate)
    h = bytearray(7)
    h[0] = 0xFF
    h[1] = 0xF0 | (0x08 if mpeg2 else 0) | (0 if crc else 1)
    h[2] = (profile << 6) | (fi << 2) | ((channels >> 2) & 1)
    h[3] = ((channels & 3) << 6) | ((n >> 11) & 3)
    h[4] = (n >> 3) & 0xFF
    h[5] = ((n & 7) << 5) | 0x1F
    h[6] = 0xFC | (blocks - 1)
    return bytes(h) + (b"\0\0" if crc else b"") + payload


# ---- the reader -------------------------------------------------------------------------------------------------------
def _boxes(d: bytes, start: int, end: int):
    pos = start
    while pos + 8 <= end:
        size, typ = struct.unpack(">I4s", d[pos:pos + 8])
        hdr = 8
        if size == 1:
            if pos + 16 > end:
                return
            size = struct.unpack(">Q", d[pos + 8:pos + 16])[0]
            hdr = 16
        elif size == 0:
            size = end - pos
        if size < hdr or pos + size > end:
            return
        yield typ, pos + hdr, pos + size
        pos += size


def _child(d, start, end, typ):
    for t, a, b in _boxes(d, start, end):
        if t == typ:
            return a, b
    return None


def _descr(d, pos, end):
    tag = d[pos]
    pos += 1
    n = 0
    for _ in range(4):
        c = d[pos]
        pos += 1
        n = (n << 7) | (c & 0x7F)
        if not c & 0x80:
            break
    return tag, pos, n


def audio_tracks(d: bytes) -> List[dict]:
    moov = _child(d, 0, len(d), b"moov")
    if moov is None:
        raise ValueError("no moov box")
    out = []
    for typ, a, b in _boxes(d, *moov):
        if typ != b"trak":
            continue
        mdia = _child(d, a, b, b"mdia")
        if not mdia:
            continue
        hd = _child(d, *mdia, b"hdlr")
        if not hd or d[hd[0] + 8:hd[0] + 12] != b"soun":
            continue
        stbl = _child(d, *_child(d, *mdia, b"minf"), b"stbl")
        sd = _child(d, *stbl, b"stsd")
        ent = next(_boxes(d, sd[0] + 8, sd[1]))
        et, ea, eb = ent
        ver = struct.unpack(">H", d[ea + 8:ea + 10])[0]
        info = {"channels": struct.unpack(">H", d[ea + 16:ea + 18])[0], "sample_rate": struct.unpack(">I", d[ea + 24:ea + 28])[0] >> 16,
                "object_type": 0, "asc": b""}
        kids = ea + 28 + (16 if ver == 1 else 0)
        if et == b".mp3":
            info["codec"] = "mp3"
        elif et == b"mp4a":
            es = _child(d, kids, eb, b"esds")
            if es:
                pos = es[0] + 4
                tag, pos, n = _descr(d, pos, es[1])
                flags = d[pos + 2]
                pos += 3 + (2 if flags & 0x80 else 0) + (2 if flags & 0x20 else 0)
                tag, pos, n = _descr(d, pos, es[1])
                info["object_type"] = d[pos]
                dend = pos + n
                pos += 13
                if pos < dend:
                    tag, pos, n = _descr(d, pos, dend)
                    if tag == 5:
                        asc = d[pos:pos + n]
                        info["asc"] = asc
                        bits = "".join(format(x, "08b") for x in asc)
                        p = 5
                        aot = int(bits[:5], 2)
                        if aot == 31:
                            aot = 32 + int(bits[5:11], 2)
                            p = 11
                        fi = int(bits[p:p + 4], 2)
                        p += 4
                        if fi == 15:
                            info["sample_rate"] = int(bits[p:p + 24], 2)
                            p += 24
                        elif fi < 13:
                            info["sample_rate"] = ASC_RATES[fi]
                        cc = int(bits[p:p + 4], 2)
                        if 1 <= cc <= 7:
                            info["channels"] = 8 if cc == 7 else cc
            oti = info["object_type"]
            if oti == 0x40 or 0x66 <= oti <= 0x68:
                info["codec"] = "aac"
            elif oti in (0x69, 0x6B):
                info["codec"] = "mp3"
            else:
                continue
        else:
            continue
        info["stbl"] = stbl
        out.append(info)
    return out


def access_units(d: bytes, track: dict) -> List[Tuple[int, int]]:
    stbl = track["stbl"]
    z = _child(d, *stbl, b"stsz")
    sizes = []
    if z:
        fixed, count = struct.unpack(">II", d[z[0] + 4:z[0] + 12])
        sizes = [fixed] * count if fixed else [struct.unpack(">I", d[z[0] + 12 + 4 * i:z[0] + 16 + 4 * i])[0] for i in range(count)]
    else:
        z = _child(d, *stbl, b"stz2")
        field = d[z[0] + 7]
        count = struct.unpack(">I", d[z[0] + 8:z[0] + 12])[0]
        tab = d[z[0] + 12:z[1]]
        if field == 16:
            sizes = [struct.unpack(">H", tab[2 * i:2 * i + 2])[0] for i in range(count)]
        elif field == 8:
            sizes = list(tab[:count])
        else:
            sizes = [(tab[i // 2] >> 4) if i % 2 == 0 else (tab[i // 2] & 15) for i in range(count)]
    c = _child(d, *stbl, b"stsc")
    nr = struct.unpack(">I", d[c[0] + 4:c[0] + 8])[0]
    runs = [struct.unpack(">III", d[c[0] + 8 + 12 * i:c[0] + 20 + 12 * i]) for i in range(nr)]
    o = _child(d, *stbl, b"stco")
    wide = False
    if not o:
        o = _child(d, *stbl, b"co64")
        wide = True
    nc = struct.unpack(">I", d[o[0] + 4:o[0] + 8])[0]
    offs = [struct.unpack(">Q" if wide else ">I", d[o[0] + 8 + (8 if wide else 4) * i:o[0] + 8 + (8 if wide else 4) * (i + 1)])[0] for i in range(nc)]
    out, s = [], 0
    for ci in range(1, nc + 1):
        per = 0
        for f, n, _ in runs:
            if f <= ci:
                per = n
        off = offs[ci - 1]
        for _ in range(per):
            if s >= len(sizes):
                break
            if off + sizes[s] > len(d):
                return out
            out.append((off, sizes[s]))
            off += sizes[s]
            s += 1
    return out
