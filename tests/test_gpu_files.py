"""GPU parity of the file-level layer (SURVEY.md 8b last row): analyze_track / analyze_album /
find_peak_amplitude on WAV files and decoder pipes, through the C ABI, against the CPU oracle run on the
planar samples the WAV holds.  Loudness is held to exact equality with the oracle here (the +-0.1 dB
north_star tolerance is DB_TOL); peaks are bit-equal."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent))
from wavutil import planar_for_oracle, test_signal, wav_bytes  # noqa: E402

test_signal.__test__ = False
ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu
DB_TOL = 0.1


@pytest.fixture()
def an(_ctx):
    _ctx.set_kernel(0)
    _ctx.set_tuning(1, 0)
    _ctx.set_tuning(2, 0)
    _ctx.set_decoder_command(None)
    return _ctx


def oracle_result(oracle, chans, kind, rate):
    pl = planar_for_oracle(chans, kind)
    return oracle.analyze_pcm(pl[0], pl[1] if len(pl) > 1 else None, rate)


@pytest.mark.parametrize("kind", ["u8", "s16", "s24", "s32", "f32"])
@pytest.mark.parametrize("nch,frames", [(1, 44100 + 333), (2, 2 * 44100 + 8), (2, 44100 + 1001), (6, 30011)])
def test_analyze_track_wav(an, oracle, tmp_path, kind, nch, frames):
    import mp3rgain_amd as rg

    rate = 44100
    chans = test_signal(kind, rate, frames, nch, seed=frames + nch)
    f = tmp_path / f"t_{kind}_{nch}.wav"
    f.write_bytes(wav_bytes(chans, rate, kind, extensible=(nch > 2)))
    got = an.analyze_track_file(f)
    want, wh = oracle_result(oracle, chans, kind, rate)
    assert abs(got.loudness_db - want["loudness_db"]) <= DB_TOL
    assert got.loudness_db == want["loudness_db"] and got.gain_db == want["gain_db"]
    assert got.gain_steps() == want["gain_steps"]
    assert got.peak == want["peak"]
    assert got.sample_rate == rate and got.windows == int(wh.sum())
    assert got.file_type == rg.AudioFileType.Mp3  # detect_file_type: everything that is not MP4 (:777-783)
    # the same samples through the planar PCM entry point give the same result, bit for bit
    same = an.analyze_track(rg.PcmTrack(planar_for_oracle(chans, kind), rate))
    assert (same.loudness_db, same.gain_db, same.peak, same.windows) == (got.loudness_db, got.gain_db, got.peak, got.windows)
    # find_peak_amplitude looks at every channel (:1212-1238)
    pk = an.find_peak_amplitude_file(f)
    pl = planar_for_oracle(chans, kind)
    assert pk.peak == oracle.find_peak(pl, pl[0].dtype)
    assert pk.peak_pcm == pk.peak * 32768.0 and pk.sample_rate == rate


def test_analyze_album_files(an, oracle, tmp_path):
    import mp3rgain_amd as rg

    specs = [("f32", 48000, 2, 48000 * 2 + 4), ("s16", 44100, 2, 44100 * 3), ("s24", 44100, 1, 50000), ("f32", 44100, 2, 44100 + 17)]
    files, hists, peaks, wants = [], [], [], []
    for i, (kind, rate, nch, frames) in enumerate(specs):
        chans = test_signal(kind, rate, frames, nch, seed=100 + i)
        f = tmp_path / f"a{i}.wav"
        f.write_bytes(wav_bytes(chans, rate, kind, streamed=(i == 3)))
        files.append(f)
        want, wh = oracle_result(oracle, chans, kind, rate)
        wants.append(want)
        hists.append(wh)
        peaks.append(want["peak"])
    album = an.analyze_album_files(files)
    ref, _ = oracle.album_from_hists(hists, peaks)
    assert [t.loudness_db for t in album.tracks] == [w["loudness_db"] for w in wants]
    assert [t.peak for t in album.tracks] == peaks
    assert [t.sample_rate for t in album.tracks] == [s[1] for s in specs]
    assert album.album_loudness_db == ref["album_loudness_db"] and album.album_gain_db == ref["album_gain_db"]
    assert album.album_peak == max(peaks)
    # module-level mirrors of the reference's function names take paths too
    one = rg.analyze_track(files[1])
    assert one.loudness_db == wants[1]["loudness_db"]
    assert rg.analyze_album([str(f) for f in files]).album_gain_db == album.album_gain_db
    assert rg.find_peak_amplitude(files[0]).peak == peaks[0]


def test_decoder_command_and_file_type(an, oracle, tmp_path):
    import mp3rgain_amd as rg

    rate, frames = 44100, 44100 + 99
    chans = test_signal("f32", rate, frames, 2, seed=7)
    wav = wav_bytes(chans, rate, "f32", streamed=True)
    want, _ = oracle_result(oracle, chans, "f32", rate)
    # "compressed" stand-ins: the WAV stream behind a 4-byte prefix; the decoder strips it
    mp3 = tmp_path / "it's a song.mp3"  # the path is shell-quoted
    mp3.write_bytes(b"JUNK" + wav)
    with pytest.raises(rg.ReplayGainError, match="Failed to probe format"):
        an.analyze_track_file(mp3)
    an.set_decoder_command("tail -c +5 {}")
    got = an.analyze_track_file(mp3)
    assert got.loudness_db == want["loudness_db"] and got.peak == want["peak"] and got.file_type == rg.AudioFileType.Mp3
    m4a = tmp_path / "song.m4a"
    m4a.write_bytes(b"\0\0\0\x14ftypM4A \0\0\0\0M4A " + wav)  # is_mp4_file looks at the ftyp brand (mp4meta.rs:872-889)
    an.set_decoder_command("tail -c +21")  # no {}: the quoted path is appended
    got = an.analyze_track_file(m4a)
    assert got.loudness_db == want["loudness_db"] and got.file_type == rg.AudioFileType.Aac
    assert an.find_peak_amplitude_file(m4a).peak == want["peak"]
    an.set_decoder_command("false {}")
    with pytest.raises(rg.ReplayGainError, match="Failed to probe format"):
        an.analyze_track_file(mp3)
    an.set_decoder_command(None)


def test_file_level_errors(an, tmp_path):
    import mp3rgain_amd as rg

    with pytest.raises(rg.ReplayGainError, match="Failed to open: "):
        an.analyze_track_file(tmp_path / "missing.wav")
    chans = test_signal("s16", 44000, 5000, 2, seed=1)
    odd = tmp_path / "odd_rate.wav"
    odd.write_bytes(wav_bytes(chans, 44000, "s16"))
    with pytest.raises(rg.ReplayGainError, match=r"Unsupported sample rate: 44000 Hz\. Supported rates: 96000, 88200"):
        an.analyze_track_file(odd)
    ok = tmp_path / "ok.wav"
    ok.write_bytes(wav_bytes(test_signal("s16", 44100, 5000, 2, seed=2), 44100, "s16"))
    assert an.analyze_track_file(ok, 0).sample_rate == 44100
    with pytest.raises(rg.ReplayGainError, match=r"Track index 1 out of range \(file has 1 audio track\(s\)\)"):
        an.analyze_track_file(ok, 1)
    # album: the first failing file aborts the whole call (src/replaygain.rs:1055)
    with pytest.raises(rg.ReplayGainError, match="Failed to open: "):
        an.analyze_album_files([ok, tmp_path / "missing.wav", ok])
    junk = tmp_path / "junk.mp3"
    junk.write_bytes(b"ID3" + bytes(500))
    with pytest.raises(rg.ReplayGainError, match="Failed to probe format: .*junk.mp3"):
        an.analyze_album_files([ok, junk])
    with pytest.raises(rg.ReplayGainError, match="Failed to probe format: .*junk.mp3"):
        an.find_peak_amplitude_file(junk)
    with pytest.raises(rg.ReplayGainError, match="input 1 is not a RIFF/WAVE stream"):
        an.analyze_wav_bytes([ok.read_bytes(), b"nonsense"])
    f64 = bytearray(wav_bytes([np.zeros(10)], 44100, "f32", extra_chunks=False))
    f64[34:36] = (64).to_bytes(2, "little")
    f64[32:34] = (8).to_bytes(2, "little")
    bad = tmp_path / "f64.wav"
    bad.write_bytes(bytes(f64))
    with pytest.raises(rg.ReplayGainError, match="Failed to probe format"):
        an.analyze_track_file(bad)
    # an empty data chunk is an empty track: no windows -> -20 dB fall-through of get_loudness (:670-672)
    empty = tmp_path / "empty.wav"
    empty.write_bytes(wav_bytes([np.zeros(0), np.zeros(0)], 44100, "s16", extra_chunks=False))
    r = an.analyze_track_file(empty)
    assert r.windows == 0 and r.loudness_db == -20.0 and r.peak == 0.0


def test_large_stereo_files_fast_paths(an, oracle, tmp_path):
    """16-byte de-interleave paths (frames % 4 == 0 / % 8 == 0) and the scalar path give the same planes."""
    import mp3rgain_amd as rg

    rate = 48000
    for kind, frames in (("f32", 48000 * 20), ("f32", 48000 * 20 + 3), ("s16", 48000 * 20), ("s16", 48000 * 20 + 4), ("s32", 48000 * 5)):
        chans = test_signal(kind, rate, frames, 2, seed=frames % 1000)
        got = an.analyze_wav_bytes([wav_bytes(chans, rate, kind, extra_chunks=False)])[0]
        same = an.analyze_track(rg.PcmTrack(planar_for_oracle(chans, kind), rate))
        assert (got.loudness_db, got.peak, got.windows) == (same.loudness_db, same.peak, same.windows)


def test_rg_create_reports_pipeline_streams_that_share_a_hardware_queue(tmp_path):
    """INTEGRATION.md 'Hardware queues': a context times a spinning kernel on one of its pipeline streams against one on each;
    with GPU_MAX_HW_QUEUES=1 the four streams run one after the other and the context says so (stderr, and rg_last_error until a
    real error replaces it) -- results are unaffected; with the default four queues it is silent."""
    import subprocess
    import sys

    code = ("import sys; sys.path.insert(0, %r)\n"
            "import mp3rgain_amd as rg\n"
            "from mp3rgain_amd import _capi\n"
            "an = rg.Analyzer(0)\n"
            "print('LAST_ERROR=' + _capi.load().rg_last_error(an._ctx).decode())\n") % str(ROOT)
    env = dict(os.environ)
    env.pop("GPU_MAX_HW_QUEUES", None)
    quiet = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert quiet.returncode == 0, quiet.stderr
    assert "share hardware queues" not in quiet.stderr and "LAST_ERROR=\n" in quiet.stdout + "\n", (quiet.stdout, quiet.stderr)
    env["GPU_MAX_HW_QUEUES"] = "1"
    one = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert one.returncode == 0, one.stderr
    assert "share hardware queues" in one.stderr and "share hardware queues" in one.stdout, (one.stdout, one.stderr)
