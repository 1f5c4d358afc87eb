"""CPU oracle for the lossless MP3 gain path -- TEST INFRASTRUCTURE ONLY (imported by tests/ only).

A pure-Python restatement of mp3rgain v1.5.0 src/lib.rs: frame walk, global_gain locations, the
saturating / wrapping patch and the APEv2 tag round trip.  Each function cites the lines it follows.
Pinned by the reference's own unit tests (src/lib.rs:1340-1444) and integration tests
(tests/integration_tests.rs), both restated in tests/test_mp3gain.py on the reference's fixture files
(tests/golden/fixtures/*.mp3 are byte copies of /root/reference/tests/fixtures/*.mp3)."""
from __future__ import annotations

KBPS_V1 = [0, 32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320]   # lib.rs:152-154
KBPS_V2 = [0, 8, 16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128, 144, 160]       # lib.rs:157-158
RATES = [[44100, 48000, 32000], [22050, 24000, 16000], [11025, 12000, 8000]]    # lib.rs:161-165
APE = b"APETAGEX"


def parse_header(h: bytes):
    """lib.rs:169-252 -> dict or None"""
    if len(h) < 4 or h[0] != 0xFF or (h[1] & 0xE0) != 0xE0:
        return None
    vb = (h[1] >> 3) & 3
    if vb == 1:
        return None
    version = {0: "MPEG2.5", 2: "MPEG2", 3: "MPEG1"}[vb]
    if (h[1] >> 1) & 3 != 1:
        return None
    bi = (h[2] >> 4) & 15
    if bi in (0, 15):
        return None
    si = (h[2] >> 2) & 3
    if si == 3:
        return None
    kbps = KBPS_V1[bi] if version == "MPEG1" else KBPS_V2[bi]
    rate = RATES[{"MPEG1": 0, "MPEG2": 1, "MPEG2.5": 2}[version]][si]
    padding = 1 if h[2] & 2 else 0
    mode = ["Stereo", "Joint Stereo", "Dual Channel", "Mono"][(h[3] >> 6) & 3]
    spf = 1152 if version == "MPEG1" else 576
    return {"version": version, "has_crc": (h[1] & 1) == 0, "bitrate_kbps": kbps, "sample_rate": rate,
            "padding": padding, "mode": mode, "frame_size": (spf * kbps * 125) // rate + padding}


def gain_locations(off: int, h):
    """lib.rs:262-298 -> [(byte, bit)]"""
    nch = 1 if h["mode"] == "Mono" else 2
    ngr = 2 if h["version"] == "MPEG1" else 1
    lead = {("MPEG1", 1): 18, ("MPEG1", 2): 20}.get((h["version"], nch), 9 if nch == 1 else 10)
    rec = 59 if h["version"] == "MPEG1" else 63
    base = off + (6 if h["has_crc"] else 4)
    out = []
    for gr in range(ngr):
        for ch in range(nch):
            bit = lead + (gr * nch + ch) * rec + 21
            out.append((base + bit // 8, bit % 8))
    return out


def read_gain(d, loc):
    """lib.rs:301-317"""
    i, s = loc
    if i >= len(d):
        return 0
    if s == 0:
        return d[i]
    hi = (d[i] << s) & 0xFF
    return hi | (d[i + 1] >> (8 - s)) if i + 1 < len(d) else hi


def write_gain(d: bytearray, loc, v: int):
    """lib.rs:320-340"""
    i, s = loc
    if i >= len(d):
        return
    if s == 0:
        d[i] = v
        return
    mh = (0xFF << (8 - s)) & 0xFF
    d[i] = (d[i] & mh) | (v >> s)
    if i + 1 < len(d):
        d[i + 1] = (d[i + 1] & (0xFF >> s)) | ((v << (8 - s)) & 0xFF)


def skip_id3v2(d) -> int:
    """lib.rs:343-354"""
    if len(d) < 10 or bytes(d[0:3]) != b"ID3":
        return 0
    return 10 + (((d[6] & 0x7F) << 21) | ((d[7] & 0x7F) << 14) | ((d[8] & 0x7F) << 7) | (d[9] & 0x7F))


def _le32(d, o):
    return int.from_bytes(bytes(d[o:o + 4]), "little")


def find_audio_end(d) -> int:
    """lib.rs:358-383"""
    end = len(d)
    if end >= 128 and bytes(d[end - 128:end - 125]) == b"TAG":
        end -= 128
    if end >= 32 and bytes(d[end - 32:end - 24]) == APE:
        f = end - 32
        size = _le32(d, f + 12)
        hdr = 32 if _le32(d, f + 20) & (1 << 31) else 0
        if f + 32 >= size + hdr:
            end = f + 32 - size - hdr
    return end


def is_xing(d, off, h) -> bool:
    """lib.rs:388-408"""
    if h["version"] == "MPEG1":
        side = 17 if h["mode"] == "Mono" else 32
    else:
        side = 9 if h["mode"] == "Mono" else 17
    x = off + (6 if h["has_crc"] else 4) + side
    return x + 4 <= len(d) and bytes(d[x:x + 4]) in (b"Xing", b"Info")


def frames(d):
    """iterate_frames, lib.rs:412-461 -> [(offset, header)]"""
    end, pos, out = find_audio_end(d), skip_id3v2(d), []
    while pos + 4 <= end:
        h = parse_header(bytes(d[pos:pos + 4]))
        if h is None:
            pos += 1
            continue
        nxt = pos + h["frame_size"]
        ok = (d[nxt] == 0xFF and (d[nxt + 1] & 0xE0) == 0xE0) if nxt + 2 <= end else nxt <= end
        if not ok:
            pos += 1
            continue
        if not is_xing(d, pos, h):
            out.append((pos, h))
        pos = nxt
    return out


def analyze(d):
    """lib.rs:470-514"""
    fr = frames(d)
    if not fr:
        raise ValueError("No valid MP3 frames found")
    gains = [read_gain(d, loc) for off, h in fr for loc in gain_locations(off, h)]
    return {"frame_count": len(fr), "mpeg_version": fr[0][1]["version"], "channel_mode": fr[0][1]["mode"],
            "min_gain": min(gains), "max_gain": max(gains), "avg_gain": sum(gains) / len(gains),
            "headroom_steps": 255 - max(gains), "headroom_db": (255 - max(gains)) * 1.5}


def adjust(cur: int, steps: int, wrap: bool) -> int:
    """lib.rs:526-540 (Rust % truncates toward zero)"""
    if wrap:
        v = cur + steps
        r = abs(v) % 256 * (1 if v >= 0 else -1)
        return (r + 256) % 256
    return min(255, cur + min(steps, 255)) if steps > 0 else max(0, cur - min(-steps, 255))


def apply_gain(d: bytearray, steps: int, wrap: bool = False, channel=None) -> int:
    """apply_gain_to_data lib.rs:544-592 / apply_gain_to_channel_data :677-737"""
    fr = frames(d)
    for off, h in fr:
        locs = gain_locations(off, h)
        if channel is None:
            pick = locs
        else:
            nch = 1 if h["mode"] == "Mono" else 2
            ngr = 2 if h["version"] == "MPEG1" else 1
            pick = [locs[g * nch + channel] for g in range(ngr) if g * nch + channel < len(locs)]
        for loc in pick:
            write_gain(d, loc, adjust(read_gain(d, loc), steps, wrap and channel is None))
    return len(fr)


def ape_footer(d):
    """lib.rs:944-966"""
    if len(d) < 32:
        return None
    if bytes(d[len(d) - 32:len(d) - 24]) == APE:
        return len(d) - 32
    if len(d) >= 160 and bytes(d[len(d) - 160:len(d) - 152]) == APE and bytes(d[len(d) - 128:len(d) - 125]) == b"TAG":
        return len(d) - 160
    return None


def ape_read(d):
    """lib.rs:974-1027 -> [(key, value)] or None"""
    f = ape_footer(d)
    if f is None or _le32(d, f + 8) != 2000:
        return None
    size, n = _le32(d, f + 12), _le32(d, f + 16)
    if f + 32 < size:
        return None
    pos, items = f + 32 - size, []
    for _ in range(n):
        if pos + 8 > f:
            break
        vlen = _le32(d, pos)
        pos += 8
        k0 = pos
        while pos < f and d[pos] != 0:
            pos += 1
        if pos >= f:
            break
        key = bytes(d[k0:pos]).decode("utf-8", "replace")
        pos += 1
        if pos + vlen > f:
            break
        items.append((key, bytes(d[pos:pos + vlen]).decode("utf-8", "replace")))
        pos += vlen
    return items


def ape_serialize(items) -> bytes:
    """lib.rs:1037-1085"""
    if not items:
        return b""
    body = b"".join(len(v.encode()).to_bytes(4, "little") + b"\0\0\0\0" + k.encode() + b"\0" + v.encode() for k, v in items)
    size, n = len(body) + 32, len(items)

    def block(flags):
        return APE + (2000).to_bytes(4, "little") + size.to_bytes(4, "little") + n.to_bytes(4, "little") + \
            flags.to_bytes(4, "little") + b"\0" * 8
    return block((1 << 31) | (1 << 29)) + body + block(1 << 31)


def ape_strip(d) -> bytes:
    """remove_ape_tag, lib.rs:1088-1119"""
    f = ape_footer(d)
    if f is None:
        return bytes(d)
    size = _le32(d, f + 12)
    hdr = 32 if _le32(d, f + 20) & (1 << 31) else 0
    audio_end = f + 32 - size - hdr if f + 32 >= size + hdr else 0
    id3 = f + 32
    out = bytes(d[:audio_end])
    if len(d) > id3 + 3 and bytes(d[id3:id3 + 3]) == b"TAG":
        out += bytes(d[id3:])
    return out


def ape_write(d, items) -> bytes:
    """write_ape_tag, lib.rs:1122-1150 on bytes"""
    audio = ape_strip(d)
    tag = ape_serialize(items)
    if len(audio) >= 128 and audio[-128:-125] == b"TAG":
        return audio[:-128] + tag + audio[-128:]
    return audio + tag


def ape_set(items, key, value):
    """ApeTag::set, lib.rs:885-899"""
    items = list(items)
    for i, (k, _) in enumerate(items):
        if k.upper() == key.upper():
            items[i] = (k, value)
            return items
    items.append((key.upper(), value))
    return items


def undo_value(l, r, wrap):
    """set_undo_gain format "{:+04},{:+04},{}", lib.rs:926-930"""
    return f"{l:+04d},{r:+04d},{'W' if wrap else 'N'}"
