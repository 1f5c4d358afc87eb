"""MP4 / M4A files through the file-level entry points (rg_files.hip + rg_demux.cpp): track discovery and selection with the
reference's error texts (src/replaygain.rs:827-858), MPEG Layer III inside MP4 decoded by the library from the sample table
(bit-identical to the bare stream), AAC tracks handed to the decoder command with the selected track's index."""
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "oracle"))
sys.path.insert(0, str(ROOT / "tests"))
import mp3_bitstream as B  # noqa: E402
import mp4demux_oracle as M  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = ROOT / "tests" / "golden" / "mp3"


def _mp3_frames(data: bytes):
    out, pos = [], 0
    while pos + 4 <= len(data):
        h = data[pos:pos + 4]
        ver = (h[1] >> 3) & 3
        lsf = ver != 3
        br = (B.BITRATES_V2 if lsf else B.BITRATES_V1)[h[2] >> 4]
        rate = [44100, 48000, 32000][(h[2] >> 2) & 3] >> (0 if ver == 3 else (1 if ver == 2 else 2))
        fb = (72 if lsf else 144) * br * 1000 // rate + ((h[2] >> 1) & 1)
        out.append(data[pos:pos + fb])
        pos += fb
    return out


def _same(a, b):
    assert (a.loudness_db, a.gain_db, a.peak, a.sample_rate, a.windows) == (b.loudness_db, b.gain_db, b.peak, b.sample_rate, b.windows)


def test_mp3_in_mp4_and_track_selection(_ctx, tmp_path):
    import mp3rgain_amd as rg
    from wavutil import test_signal, wav_bytes

    an = _ctx
    an.set_kernel(0)
    a = (GOLD / "dense_48k_stereo_192.mp3").read_bytes()
    b = (GOLD / "v2_22k_stereo.mp3").read_bytes()
    (tmp_path / "a.mp3").write_bytes(a)
    (tmp_path / "b.mp3").write_bytes(b)
    ra, rb = an.analyze_track_file(tmp_path / "a.mp3"), an.analyze_track_file(tmp_path / "b.mp3")
    rng = np.random.default_rng(1)
    vid = M.Track("video", [bytes(rng.integers(0, 256, 300, dtype=np.uint8)) for _ in range(9)])
    alac = M.Track("alac", [b"\0" * 50] * 4)
    ta = M.Track("mp3", _mp3_frames(a), rate=48000, per_chunk=(5, 9, 2), co64=True)
    tb = M.Track("mp3_qt", _mp3_frames(b), rate=22050)
    f = tmp_path / "two.m4a"
    f.write_bytes(M.build_mp4([vid, ta, alac, tb], moov_first=False))
    for route in (3, 2, 1, 0):
        an.set_tuning(6, route)
        try:
            r0, r1 = an.analyze_track_file(f), an.analyze_track_file(f, track_index=1)
            r00 = an.analyze_track_file(f, track_index=0)
        finally:
            an.set_tuning(6, 3)
        _same(r0, ra)
        _same(r00, ra)
        _same(r1, rb)
        assert r0.file_type == rg.AudioFileType.Aac and ra.file_type == rg.AudioFileType.Mp3  # detect_file_type looks at the container
    with pytest.raises(rg.ReplayGainError, match=r"Track index 2 out of range \(file has 2 audio track\(s\)\)"):
        an.analyze_track_file(f, track_index=2)
    with pytest.raises(rg.ReplayGainError, match=r"Track index 1 out of range \(file has 1 audio track\(s\)\)"):
        an.analyze_track_file(tmp_path / "a.mp3", track_index=1)
    none = tmp_path / "video_only.mp4"
    none.write_bytes(M.build_mp4([vid, alac], brand=b"isom"))
    with pytest.raises(rg.ReplayGainError, match="No audio track found"):
        an.analyze_track_file(none)
    # batch and album entry points see the same thing, per file
    got = an.analyze_track_files([f, none, tmp_path / "b.mp3"], track_index=1)
    _same(got[0], rb)
    assert isinstance(got[1], rg.ReplayGainError) and "No audio track found" in str(got[1])
    assert isinstance(got[2], rg.ReplayGainError) and "Track index 1 out of range (file has 1 audio track(s))" in str(got[2])
    alb = an.analyze_album_files([f, tmp_path / "b.mp3"])
    want = an.analyze_album_files([tmp_path / "a.mp3", tmp_path / "b.mp3"])
    assert (alb.album_loudness_db, alb.album_peak) == (want.album_loudness_db, want.album_peak)
    with rg.Node([0, 0]) as node:
        nb = node.analyze_album_files([f, tmp_path / "b.mp3", tmp_path / "a.mp3"])
        assert [t.loudness_db for t in nb.tracks] == [ra.loudness_db, rb.loudness_db, ra.loudness_db]
    # peak scan goes through the same loader
    assert an.find_peak_amplitude_file(f).peak == an.find_peak_amplitude_file(tmp_path / "a.mp3").peak


def test_aac_tracks_go_to_the_decoder_command(_ctx, tmp_path):
    """No AAC decoder is built: an AAC track is the reference's "Failed to create decoder" (src/replaygain.rs:861-863)
    unless a decoder command is set; the command gets the path and, for `{track}`, the index of the selected audio track."""
    import mp3rgain_amd as rg
    from wavutil import test_signal, wav_bytes

    an = _ctx
    an.set_kernel(0)
    rng = np.random.default_rng(2)
    aac = [M.Track("aac", [bytes(rng.integers(0, 256, 200, dtype=np.uint8)) for _ in range(20)], rate=r, channels=2) for r in (44100, 48000)]
    f = tmp_path / "song.m4a"
    f.write_bytes(M.build_mp4(aac))
    with pytest.raises(rg.ReplayGainError, match="Failed to create decoder") as ei:
        an.analyze_track_file(f)
    assert ei.value.code == -9
    wavs = []
    for k, rate in enumerate((44100, 48000)):
        chans = test_signal("s16", rate, rate * 2 + 100 * k, 2, seed=10 + k)
        w = tmp_path / f"t{k}.wav"
        w.write_bytes(wav_bytes(chans, rate, "s16"))
        wavs.append(w)
    try:
        an.set_decoder_command(f"cat {tmp_path}/t{{track}}.wav; true {{}}")
        want = [an.analyze_track_file(w) for w in wavs]
        r0, r1 = an.analyze_track_file(f), an.analyze_track_file(f, track_index=1)
        _same(r0, want[0])
        _same(r1, want[1])
        assert r0.file_type == rg.AudioFileType.Aac
        with pytest.raises(rg.ReplayGainError, match=r"Track index 5 out of range \(file has 2 audio track\(s\)\)"):
            an.analyze_track_file(f, track_index=5)
    finally:
        an.set_decoder_command(None)
