"""MP4/M4A ReplayGain tags (second half): the C++ implementation behind
include/mp3rgain_amd_mp4.h against the reference's own unit tests (src/mp4meta.rs:895-943, restated) and,
byte for byte, against the independent restatement oracle/mp4meta_oracle.py on synthetic MP4 files.
The reference ships no MP4 fixture, so the files are built here."""
import struct
import sys
from pathlib import Path

import pytest

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "oracle"))

import mp4meta_oracle as O  # noqa: E402
from mp3rgain_amd import mp4meta as M  # noqa: E402


def box(typ: bytes, body: bytes = b"") -> bytes:
    return struct.pack(">I4s", 8 + len(body), typ) + body


def box64(typ: bytes, body: bytes) -> bytes:
    return struct.pack(">I4sQ", 1, typ, 16 + len(body)) + body


def text_item(typ: bytes, value: str) -> bytes:
    return box(typ, box(b"data", struct.pack(">II", 1, 0) + value.encode()))


def stco(offsets, wide=False) -> bytes:
    body = struct.pack(">II", 0, len(offsets)) + b"".join(struct.pack(">Q" if wide else ">I", o) for o in offsets)
    return box(b"co64" if wide else b"stco", body)


PAYLOAD = bytes(range(256)) * 3


def make_mp4(layout="moov_first", udta="full", wide=False, extra_items=True, old_rg=None, mdat64=False):
    """ftyp, moov[mvhd, trak[mdia[minf[stbl[stco|co64]]]], udta?], mdat -- chunk offsets point into PAYLOAD."""
    ftyp = box(b"ftyp", b"M4A " + struct.pack(">I", 0) + b"M4A mp42isom")
    items = b""
    if extra_items:
        items += text_item(b"\xa9nam", "Song") + O.serialize_freeform("com.apple.iTunes", "iTunNORM", " 0000 0001")
    for k, v in (old_rg or {}).items():
        items += O.serialize_freeform("com.apple.iTunes", k, v)
    if extra_items:
        items += text_item(b"\xa9ART", "Artist")
    hdlr = box(b"hdlr", b"\0" * 8 + b"mdirappl" + b"\0" * 9)
    if udta == "full":
        u = box(b"udta", box(b"meta", b"\0" * 4 + hdlr + box(b"ilst", items)))
    elif udta == "meta_no_ilst":
        u = box(b"udta", box(b"meta", b"\0" * 4 + hdlr))
    elif udta == "empty":
        u = box(b"udta", box(b"cprt", b"\0" * 6 + b"x"))
    else:
        u = b""
    mdat_hdr = 16 if mdat64 else 8

    def moov_with(offs):
        stbl = box(b"stbl", box(b"stsd", b"\0" * 8) + stco(offs, wide))
        trak = box(b"trak", box(b"tkhd", b"\0" * 84) + box(b"mdia", box(b"mdhd", b"\0" * 24) + box(b"minf", box(b"smhd", b"\0" * 8) + stbl)))
        return box(b"moov", box(b"mvhd", b"\0" * 100) + trak + u)

    rel = [0, 100, 300, 700]
    mdat = (box64 if mdat64 else box)(b"mdat", PAYLOAD)
    if layout == "moov_first":
        base = len(ftyp) + len(moov_with([0] * 4)) + mdat_hdr
        data = ftyp + moov_with([base + r for r in rel]) + mdat
    else:
        base = len(ftyp) + mdat_hdr
        data = ftyp + mdat + moov_with([base + r for r in rel])
    return data, rel


def chunk_offsets(data: bytes):
    i = data.find(b"stco")
    w = 4
    if i < 0:
        i, w = data.find(b"co64"), 8
    n = struct.unpack_from(">I", data, i + 8)[0]
    return [struct.unpack_from(">Q" if w == 8 else ">I", data, i + 12 + k * w)[0] for k in range(n)]


def top_level(data: bytes):
    pos, out = 0, []
    while pos < len(data):
        size, typ = struct.unpack_from(">I4s", data, pos)
        if size == 1:
            size = struct.unpack_from(">Q", data, pos + 8)[0]
        assert size >= 8
        out.append((typ, pos, size))
        pos += size
    assert pos == len(data)
    return out


def tags_pair(track=None, album=None):
    t, o = M.ReplayGainTags(), O.Tags()
    if track:
        t.set_track(*track)
        o.set_track(*track)
    if album:
        t.set_album(*album)
        o.set_album(*album)
    return t, o


# ---- the reference's unit tests, restated (src/mp4meta.rs:895-943) ---------------------------------------

def test_freeform_tag_serialization():  # :895-913
    tag = M.FreeformTag("com.apple.iTunes", "replaygain_track_gain", "+3.50 dB")
    ser = M.serialize_freeform_tag(tag)
    assert ser[4:8] == b"----"
    assert M.parse_freeform_tag(ser[8:]) == tag
    assert ser == O.serialize_freeform(tag.namespace, tag.name, tag.value)
    assert O.parse_freeform(ser[8:]) == (tag.namespace, tag.name, tag.value)
    assert struct.unpack_from(">I", ser, 0)[0] == len(ser)


def test_replaygain_tags():  # :915-928
    tags, otags = tags_pair((3.5, 0.98765), (2.0, 0.99999))
    assert tags.track_gain == "+3.50 dB" and tags.track_peak == "0.987650"
    assert tags.album_gain == "+2.00 dB" and tags.album_peak == "0.999990"
    assert otags.v == [tags.track_gain, tags.track_peak, tags.album_gain, tags.album_peak]
    assert not tags.is_empty() and M.ReplayGainTags().is_empty()


@pytest.mark.parametrize("gain,peak", [(-8.15, 1.0), (5.826, 0.5), (0.0, 0.0), (-0.004, 1.234567891), (12.345, 3e-7), (-64.0, 32767.5)])
def test_tag_number_formats(gain, peak):
    tags, otags = tags_pair((gain, peak))
    assert tags.track_gain == "%+.2f dB" % gain == otags.v[0]
    assert tags.track_peak == "%.6f" % peak == otags.v[1]


def test_is_mp4_detection(tmp_path):  # :930-943 + :872-889
    hdr = bytes([0, 0, 0, 0x14]) + b"ftypM4A " + bytes(4) + b"M4A "
    assert M.is_mp4_data(hdr) and O.is_mp4(hdr)
    for brand in (b"M4A ", b"M4B ", b"M4P ", b"M4V ", b"mp41", b"mp42", b"isom", b"iso2", b"qt  ", b"3gp4"):
        d = struct.pack(">I4s4sI", 16, b"ftyp", brand, 0)
        assert M.is_mp4_data(d) == O.is_mp4(d) == (brand not in (b"qt  ", b"3gp4"))
    assert not M.is_mp4_data(hdr[:11])
    assert not M.is_mp4_data(struct.pack(">I4s4sI", 8, b"ftyp", b"M4A ", 0))  # size < 12
    assert not M.is_mp4_data(b"ID3\x03" + bytes(20))
    f = tmp_path / "a.m4a"
    f.write_bytes(hdr)
    assert M.is_mp4_file(f)
    assert not M.is_mp4_file(tmp_path / "missing.m4a")
    mp3 = Path(__file__).parent / "golden" / "fixtures" / "test_stereo.mp3"
    assert not M.is_mp4_file(mp3)


# ---- whole-file behaviour against the restatement ----------------------------------------------------------

LAYOUTS = ["moov_first", "mdat_first"]
UDTAS = ["full", "meta_no_ilst", "empty", "none"]
TAGSETS = [((3.5, 0.98765), None), ((-8.15, 1.0), (-7.9, 1.0)), (None, (2.0, 0.5)), (None, None)]


@pytest.mark.parametrize("layout", LAYOUTS)
@pytest.mark.parametrize("udta", UDTAS)
@pytest.mark.parametrize("tagset", TAGSETS)
@pytest.mark.parametrize("wide", [False, True])
def test_update_matches_restatement(layout, udta, tagset, wide):
    old = {"replaygain_track_gain": "+1.00 dB", "REPLAYGAIN_TRACK_PEAK": "0.100000"} if udta == "full" else None
    data, rel = make_mp4(layout, udta, wide, old_rg=old)
    tags, otags = tags_pair(*tagset)
    got = M.update_mp4_metadata(data, tags)
    assert got == O.update(data, otags)
    # structure: top-level boxes tile the file, the payload is intact and the chunk offsets still point at it
    boxes = top_level(got)
    assert [b[0] for b in boxes] == [b[0] for b in top_level(data)]
    mdat_pos = [p for t, p, _ in boxes if t == b"mdat"][0]
    assert got[mdat_pos + 8:mdat_pos + 8 + len(PAYLOAD)] == PAYLOAD
    assert chunk_offsets(got) == [mdat_pos + 8 + r for r in rel]
    if layout == "mdat_first":
        assert chunk_offsets(got) == chunk_offsets(data)
    if udta == "meta_no_ilst":
        # reference quirk, reproduced: a meta box without ilst gets a SECOND meta box appended to udta
        # (create_or_update_ilst :587-600 -> NeedsMeta), and the reader only ever looks at the first meta
        assert got.count(b"meta") == 2 and M.read_replaygain_tags_data(got).is_empty() and O.read_tags(got).is_empty()
        return
    # the tags read back, old ReplayGain atoms (any case) are gone, foreign items survive
    back = M.read_replaygain_tags_data(got)
    assert [back.track_gain, back.track_peak, back.album_gain, back.album_peak] == otags.v == O.read_tags(got).v
    assert got.count(b"replaygain_track_gain") + got.count(b"REPLAYGAIN_TRACK_GAIN") == (1 if tagset[0] else 0)
    assert b"REPLAYGAIN_TRACK_PEAK" not in got
    if udta == "full":
        for keep in (b"\xa9nam", b"Song", b"iTunNORM", b"\xa9ART", b"Artist"):
            assert keep in got
    # writing the same tags again changes nothing
    assert M.update_mp4_metadata(got, tags) == got


def test_read_existing_tags_case_insensitive():
    data, _ = make_mp4(old_rg={"REPLAYGAIN_TRACK_GAIN": "-3.21 dB", "replaygain_track_peak": "0.912345",
                               "Replaygain_Album_Gain": "-2.00 dB", "replaygain_album_peak": "0.999999"})
    t = M.read_replaygain_tags_data(data)
    assert (t.track_gain, t.track_peak, t.album_gain, t.album_peak) == ("-3.21 dB", "0.912345", "-2.00 dB", "0.999999")
    assert O.read_tags(data).v == [t.track_gain, t.track_peak, t.album_gain, t.album_peak]
    # a freeform atom of another namespace with the same name is not a ReplayGain tag
    other = O.serialize_freeform("org.example", "replaygain_track_gain", "+9.99 dB")
    data2, _ = make_mp4(extra_items=False)
    ilst_at = data2.find(b"ilst") - 4
    assert struct.unpack_from(">I", data2, ilst_at)[0] == 8
    assert M.read_replaygain_tags_data(data2).track_gain is None
    tags, otags = tags_pair((1.0, 0.5))
    with_other = O.update(data2, O.Tags())  # empty ilst stays
    assert M.update_mp4_metadata(data2, M.ReplayGainTags()) == with_other
    assert other not in with_other


def test_no_tags_without_metadata():
    for udta in ("meta_no_ilst", "empty", "none"):
        data, _ = make_mp4(udta=udta)
        assert M.read_replaygain_tags_data(data).is_empty()
        assert O.read_tags(data).is_empty()
    assert M.read_replaygain_tags_data(b"").is_empty()
    assert M.read_replaygain_tags_data(box(b"ftyp", b"M4A \0\0\0\0")).is_empty()


def test_extended_size_mdat_and_size_fields():
    data, rel = make_mp4("moov_first", "none", mdat64=True)
    tags, otags = tags_pair((3.5, 0.9), (2.5, 0.95))
    got = M.update_mp4_metadata(data, tags)
    assert got == O.update(data, otags)
    grew = len(got) - len(data)
    assert grew > 0
    assert chunk_offsets(got) == [o + grew for o in chunk_offsets(data)]
    mdat_pos = got.find(b"mdat") - 4
    assert got[mdat_pos + 16:mdat_pos + 16 + 16] == PAYLOAD[:16]
    # moov grew by exactly the inserted udta box
    m0 = struct.unpack_from(">I", data, data.find(b"moov") - 4)[0]
    m1 = struct.unpack_from(">I", got, got.find(b"moov") - 4)[0]
    assert m1 - m0 == grew


def test_file_level_round_trip_and_errors(tmp_path):
    data, _ = make_mp4(old_rg={"replaygain_track_gain": "+0.50 dB"})
    f = tmp_path / "song.m4a"
    f.write_bytes(data)
    assert M.is_mp4_file(f)
    assert M.read_replaygain_tags(f).track_gain == "+0.50 dB"
    tags, otags = tags_pair((-6.38, 0.988), (-5.5, 1.0))
    M.write_replaygain_tags(f, tags)
    assert f.read_bytes() == O.update(data, otags)
    assert M.read_replaygain_tags(f) == tags
    M.delete_replaygain_tags(f)
    assert f.read_bytes() == O.update(O.update(data, otags), O.Tags())
    assert M.read_replaygain_tags(f).is_empty()
    assert b"iTunNORM" in f.read_bytes()
    with pytest.raises(M.Mp4MetaError, match="Failed to read"):
        M.read_replaygain_tags(tmp_path / "nope.m4a")
    bad = tmp_path / "nomoov.m4a"
    bad.write_bytes(box(b"ftyp", b"M4A \0\0\0\0") + box(b"mdat", b"abc"))
    with pytest.raises(M.Mp4MetaError, match="No moov box found in MP4 file"):
        M.write_replaygain_tags(bad, tags)
    with pytest.raises(ValueError, match="No moov box found in MP4 file"):
        O.update(bad.read_bytes(), otags)
    assert bad.read_bytes().endswith(b"abc")  # untouched


def test_damaged_files_do_not_crash():
    """Truncations and corrupted size fields: never a crash; where the restatement has an answer, the same answer."""
    import random

    rng = random.Random(20260928)
    base, _ = make_mp4(old_rg={"replaygain_track_gain": "+1.00 dB"})
    tags, otags = tags_pair((1.5, 0.75))
    cases = [base[:n] for n in range(0, len(base), 37)]
    for _ in range(300):
        b = bytearray(base)
        for _ in range(rng.randint(1, 3)):
            pos = rng.randrange(0, len(b) - 4)
            struct.pack_into(">I", b, pos, rng.choice([0, 1, 7, 8, 9, 16, rng.randrange(1 << 32), rng.randrange(2000)]))
        cases.append(bytes(b))
    agreed = 0
    for d in cases:
        try:
            got = M.update_mp4_metadata(d, tags)
        except M.Mp4MetaError:
            got = None
        M.read_replaygain_tags_data(d)
        try:
            want = O.update(d, otags)
        except Exception:
            continue
        if got is not None and got == want:
            agreed += 1
    assert agreed > len(cases) // 3


@pytest.mark.parametrize("container", [b"moov", b"udta", b"meta", b"ilst"])
@pytest.mark.parametrize("size", [2, 3, 7])
def test_container_box_smaller_than_its_header(container, size):
    """A 32-bit size of 2..7 on moov / udta / meta / ilst made `size - header` wrap to ~2^64 and the ilst walk
    read past the buffer (round-1 advisor finding, reproduced under ASan).  Now such a box does not exist:
    the library answers like the restatement, and never copies foreign bytes into the output."""
    base, _ = make_mp4(old_rg={"replaygain_track_gain": "+1.00 dB"})
    i = base.find(container) - 4
    d = bytearray(base)
    struct.pack_into(">I", d, i, size)
    d = bytes(d)
    tags, otags = tags_pair((1.5, 0.75), (-2.0, 0.5))
    try:
        want = O.update(d, otags)
    except ValueError:
        want = None
    try:
        got = M.update_mp4_metadata(d, tags)
    except M.Mp4MetaError:
        got = None
    assert got == want
    M.read_replaygain_tags_data(d)  # must simply not crash
    # extended-size form: size field 1, 64-bit size below 16
    d2 = bytearray(base)
    d2[i:i + 8] = struct.pack(">I4s", 1, container)
    d2[i + 8:i + 16] = struct.pack(">Q", size + 8)
    try:
        want2 = O.update(bytes(d2), otags)
    except ValueError:
        want2 = None
    try:
        got2 = M.update_mp4_metadata(bytes(d2), tags)
    except M.Mp4MetaError:
        got2 = None
    assert got2 == want2


def test_library_exports_every_declared_symbol():
    """include/mp3rgain_amd_mp4.h, the ctypes binding and the shared library agree on the entry points."""
    import ctypes as C
    import re

    from mp3rgain_amd import _capi

    txt = re.sub(r"/\*.*?\*/", "", (Path(__file__).resolve().parents[1] / "include" / "mp3rgain_amd_mp4.h").read_text(), flags=re.S)
    declared = sorted(set(re.findall(r"\b(rg_mp4_[a-z0-9_]+)\s*\(", txt)))
    assert declared == sorted(n for n, _, _ in M.SYMBOLS)
    raw = C.CDLL(str(_capi.LIB_PATH))
    for name in declared:
        assert hasattr(raw, name), f"{name} not exported"
    assert C.sizeof(M._Tags) == 8 + 4 * 64
