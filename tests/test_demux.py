"""include/mp3rgain_amd_demux.h (mp3rgain_amd/csrc/rg_demux.cpp): ISO base media sample tables and ADTS.

The reference's demuxer is symphonia (not in its tree) and none of its tests holds a container; what is pinned here is
the standard's text: synthetic files from an independent writer (oracle/mp4demux_oracle.py) that walks the variants --
moov before / after mdat, stco / co64, stsz fixed / table, stz2 with 4 / 8 / 16-bit fields, several stsc runs, 64-bit box
sizes, mdhd version 1, long-form descriptor lengths, AudioSpecificConfig overriding the sample entry, several audio tracks
beside a video track and an audio codec the reference's build cannot decode -- the C++ walker must return exactly the
access units that were put in, and agree with the oracle's own reader; damaged files never crash."""
import random
import struct
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "oracle"))
import mp4demux_oracle as M  # noqa: E402

from mp3rgain_amd import demux  # noqa: E402


def _samples(rng, n, lo=5, hi=400, fixed=None):
    return [bytes(rng.randrange(256) for _ in range(fixed or rng.randint(lo, hi))) for _ in range(n)]


def _check_file(data, tracks):
    """tracks: the writer's audio tracks the reference's build would decode, in file order"""
    got = demux.mp4_audio_tracks(data)
    ref = M.audio_tracks(data)
    assert len(got) == len(tracks) == len(ref)
    for i, (g, t, r) in enumerate(zip(got, tracks, ref)):
        want_codec = demux.CODEC_MP3 if t.kind.startswith("mp3") else demux.CODEC_AAC
        assert g.codec == want_codec and (r["codec"] == "mp3") == (want_codec == demux.CODEC_MP3)
        assert g.sample_rate == (t.asc_rate or t.rate) == r["sample_rate"]
        assert g.channels == (t.asc_channels or t.channels) == r["channels"]
        assert g.n_samples == len(t.samples) and g.timescale == t.rate and g.duration == 1024 * len(t.samples)
        assert bytes(g.asc[:g.asc_len]) == r["asc"]
        if t.kind.startswith("aac"):
            assert g.audio_object_type == 2 and g.object_type in (0x40, 0x67)
        au = demux.mp4_access_units(data, i)
        assert au == M.access_units(data, r)
        assert [data[o:o + s] for o, s in au] == t.samples


def test_variants_of_the_sample_tables():
    rng = random.Random(1)
    cases = [
        dict(tracks=[M.Track("aac", _samples(rng, 37))]),
        dict(tracks=[M.Track("aac", _samples(rng, 50), per_chunk=(7, 7, 3, 5, 1), co64=True)], moov_first=False),
        dict(tracks=[M.Track("aac", _samples(rng, 20, fixed=96), fixed_size=True, per_chunk=(20,))]),
        dict(tracks=[M.Track("aac", _samples(rng, 33, lo=1, hi=15), stz2=4), M.Track("aac", _samples(rng, 21, lo=1, hi=255), stz2=8, rate=48000)]),
        dict(tracks=[M.Track("aac_mpeg2", _samples(rng, 12), stz2=16, mdhd_v1=True, long_descriptors=True, rate=32000, channels=1)]),
        dict(tracks=[M.Track("aac", _samples(rng, 9), rate=22050, asc_rate=44100, asc_channels=2, channels=1)]),  # HE-AAC style: ASC wins
        dict(tracks=[M.Track("aac", _samples(rng, 9), rate=44100, entry_version=1)], brand=b"mp42"),
        dict(tracks=[M.Track("mp3", _samples(rng, 30)), M.Track("mp3_qt", _samples(rng, 11), rate=24000)], interleave=False),
    ]
    for c in cases:
        data = M.build_mp4(c["tracks"], moov_first=c.get("moov_first", True), brand=c.get("brand", b"M4A "), interleave=c.get("interleave", True))
        _check_file(data, c["tracks"])


def test_only_decodable_audio_tracks_count():
    """The reference counts the tracks whose codec its build can decode (`codec != CODEC_TYPE_NULL`, src/replaygain.rs:827-836;
    features mp3 / aac / isomp4): a video track and an ALAC track are not among them."""
    rng = random.Random(2)
    v, alac = M.Track("video", _samples(rng, 5)), M.Track("alac", _samples(rng, 5))
    a0, a1, m = M.Track("aac", _samples(rng, 14)), M.Track("aac", _samples(rng, 8), rate=48000, channels=6), M.Track("mp3", _samples(rng, 6))
    data = M.build_mp4([v, a0, alac, a1, m])
    _check_file(data, [a0, a1, m])
    assert demux.mp4_audio_tracks(M.build_mp4([v, alac])) == []
    with pytest.raises(demux.DemuxError):
        demux.mp4_access_units(data, 3)


def test_large_boxes_and_trailing_boxes():
    rng = random.Random(3)
    t = M.Track("aac", _samples(rng, 10))
    data = M.build_mp4([t], extra_moov_children=M.box(b"udta", M.box(b"meta", b"\0" * 40)))
    _check_file(data, [t])
    # the same mdat written with a 64-bit size: offsets move by 8
    ftyp_len = struct.unpack(">I", data[:4])[0]
    moov_len = struct.unpack(">I", data[ftyp_len:ftyp_len + 4])[0]
    mdat = data[ftyp_len + moov_len:]
    assert mdat[4:8] == b"mdat"
    t2 = M.Track("aac", t.samples)
    d2 = M.build_mp4([t2], moov_first=False)
    head = d2[:struct.unpack(">I", d2[:4])[0]]
    body = d2[len(head):]
    msz = struct.unpack(">I", body[:4])[0]
    wide = head + M.box(b"free", b"") + body  # a free box in front: same offsets + 8 is what the tables must say, so rebuild them
    # (the writer has no knob for it: patch the chunk offsets by +8)
    moov_at = len(head) + 8 + msz
    assert wide[moov_at + 4:moov_at + 8] == b"moov"
    w = bytearray(wide)
    at = wide.index(b"stco", moov_at) + 8
    n = struct.unpack(">I", w[at:at + 4])[0]
    for i in range(n):
        o = struct.unpack(">I", w[at + 4 + 4 * i:at + 8 + 4 * i])[0]
        w[at + 4 + 4 * i:at + 8 + 4 * i] = struct.pack(">I", o + 8)
    _check_file(bytes(w), [t2])


def test_truncated_file_ends_the_track():
    """Samples that reach past the end of the file are not returned (the reference's packet loop ends at UnexpectedEof)."""
    rng = random.Random(4)
    t = M.Track("aac", _samples(rng, 40))
    data = M.build_mp4([t])
    au = demux.mp4_access_units(data, 0)
    cut = au[25][0] + 3
    part = demux.mp4_access_units(data[:cut], 0)
    assert part == au[:25]


def test_adts():
    rng = random.Random(5)
    pay = _samples(rng, 25, lo=20, hi=600)
    stream = b"".join(M.adts_frame(p, rate=48000, channels=2, crc=(i % 3 == 0)) for i, p in enumerate(pay))
    info = demux.adts_scan(stream)
    assert (info.sample_rate, info.channels, info.profile, info.frames, info.raw_blocks, info.first_frame_offset, info.junk_bytes) == (48000, 2, 2, 25, 25, 0, 0)
    assert [stream[o:o + s] for o, s in demux.adts_access_units(stream)] == pay
    # an ID3v2 tag in front, junk between two frames, MPEG-2 signalling, 7.1
    tag = b"ID3\x04\x00\x00" + bytes([0, 0, 1, 0]) + b"\xff\xf1" * 64
    f = [M.adts_frame(p, rate=22050, channels=7, mpeg2=True) for p in pay[:6]]
    s2 = tag + b"".join(f[:3]) + b"\x00" * 13 + b"".join(f[3:])
    i2 = demux.adts_scan(s2)
    assert (i2.sample_rate, i2.channels, i2.mpeg_version, i2.frames, i2.first_frame_offset, i2.junk_bytes) == (22050, 8, 1, 6, len(tag), 13)
    with pytest.raises(demux.DemuxError):
        demux.adts_scan(b"\x00" * 100)
    with pytest.raises(demux.DemuxError):
        demux.adts_scan(b"")


def test_damaged_containers_never_crash():
    rng = random.Random(6)
    base = M.build_mp4([M.Track("video", _samples(rng, 4)), M.Track("aac", _samples(rng, 30), per_chunk=(3, 5)), M.Track("mp3", _samples(rng, 12), co64=True)])
    ok = 0
    for k in range(1500):
        d = bytearray(base)
        kind = rng.randrange(4)
        if kind == 0:
            for _ in range(rng.randint(1, 8)):
                d[rng.randrange(len(d))] = rng.randrange(256)
        elif kind == 1:
            d = d[:rng.randrange(len(d))]
        elif kind == 2:
            at = rng.randrange(len(d) - 4)
            d[at:at + 4] = struct.pack(">I", rng.choice([0, 1, 7, 0xFFFFFFFF, 0x7FFFFFFF, rng.randrange(1 << 32)]))
        else:
            a = rng.randrange(len(d))
            del d[a:a + rng.randint(1, 300)]
        d = bytes(d)
        try:
            tr = demux.mp4_audio_tracks(d)
        except demux.DemuxError:
            continue
        for i in range(len(tr)):
            try:
                for o, s in demux.mp4_access_units(d, i):
                    assert o + s <= len(d)
                ok += 1
            except demux.DemuxError:
                pass
    assert ok > 300
    for k in range(300):  # ADTS
        d = bytearray(b"".join(M.adts_frame(p) for p in _samples(rng, 8, lo=10, hi=80)))
        for _ in range(rng.randint(1, 6)):
            d[rng.randrange(len(d))] = rng.randrange(256)
        try:
            for o, s in demux.adts_access_units(bytes(d)):
                assert o + s <= len(d)
        except demux.DemuxError:
            pass


def test_sample_table_that_describes_more_bytes_than_the_file_is_refused():
    """A crafted table (fixed sample size 1, count 2^32 - 1, every chunk at the same offset) must come back as a format
    error at once -- not as four billion access units for the caller to allocate (rg_files.hip walked such a table twice)."""
    rng = random.Random(7)
    data = bytearray(M.build_mp4([M.Track("mp3", _samples(rng, 20, fixed=96), fixed_size=True, per_chunk=(1,) * 20)]))
    z = data.index(b"stsz")
    struct.pack_into(">II", data, z + 8, 1, 0xFFFFFFFF)          # sample_size = 1, sample_count = 2^32 - 1
    c = data.index(b"stsc")
    (runs,) = struct.unpack_from(">I", data, c + 8)
    for r in range(runs):
        struct.pack_into(">I", data, c + 12 + 12 * r + 4, 1000)  # samples_per_chunk of every run: a chunk stays inside the file
    o = data.index(b"stco")
    (chunks,) = struct.unpack_from(">I", data, o + 8)
    (first,) = struct.unpack_from(">I", data, o + 12)
    for k in range(chunks):
        struct.pack_into(">I", data, o + 12 + 4 * k, first)      # every chunk at the same place: the walk never runs off the file
    assert chunks == 20
    with pytest.raises(demux.DemuxError):
        demux.mp4_access_units(bytes(data), 0)
