"""Command-line façade on the GPU box: -r, -a, -e, -x and the beets TSV line run the analysis through the C ABI.
The MP3 files are the reference's fixtures (byte copies, some with their level moved by the lossless gain patcher so
that quiet / loud / clipping cases exist); the library decodes them itself (rg_mp3dec.cpp), and every printed number
is checked against the CPU oracle run on the same decoder's PCM, the applied gain in the file's global_gain fields.
Only the M4A case still goes through a stand-in decoder command: no AAC decoder is built."""
import io
import json
import math
import shutil
import sys
from pathlib import Path

import numpy as np
import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent))
from wavutil import test_signal, wav_bytes  # noqa: E402

test_signal.__test__ = False
pytestmark = pytest.mark.gpu
FIX = Path(__file__).parent / "golden" / "fixtures"

DECODER = """\
import sys, zlib
sys.path.insert(0, {tests!r})
from wavutil import test_signal, wav_bytes
name = sys.argv[1].rsplit('/', 1)[-1]
seed = zlib.crc32(name.encode()) % 1000
amp = {{'loud.mp3': 3.0, 'quiet.mp3': 0.05}}.get(name, 1.0)
ch = [(c * amp).clip(-1, 1).astype('float32') for c in test_signal('f32', 44100, 44100 * 2 + seed, 2, seed)]
sys.stdout.buffer.write(wav_bytes(ch, 44100, 'f32', streamed=True))
"""


# level shifts, in 1.5 dB steps, that turn the fixture (a 440 Hz sine at about -21 dBFS) into the named case
SHIFT = {"quiet.mp3": -12, "loud.mp3": 22}


def decoded_mp3(path):
    from mp3rgain_amd import mp3dec

    pcm, _ = mp3dec.decode(Path(path).read_bytes())
    return [pcm[c] for c in range(pcm.shape[0])]


def decoded(name: str):
    import zlib

    seed = zlib.crc32(name.encode()) % 1000
    amp = {"loud.mp3": 3.0, "quiet.mp3": 0.05}.get(name, 1.0)
    return [(c * amp).clip(-1, 1).astype(np.float32) for c in test_signal("f32", 44100, 44100 * 2 + seed, 2, seed)]


@pytest.fixture()
def box(tmp_path, _ctx):
    from mp3rgain_amd import cli

    (tmp_path / "dec.py").write_text(DECODER.format(tests=str(Path(__file__).resolve().parent)))
    dec = f"{sys.executable} {tmp_path / 'dec.py'} {{}}"

    def run(*args):
        out, err = io.StringIO(), io.StringIO()
        rc = cli.main(["--decoder", dec] + [str(a) for a in args], out, err)
        return rc, out.getvalue(), err.getvalue()

    def mp3(name, src="test_joint_stereo.mp3"):
        from mp3rgain_amd import mp3gain

        p = tmp_path / name
        shutil.copyfile(FIX / src, p)
        if name in SHIFT:
            mp3gain.apply_gain(p, SHIFT[name])
        return p

    return run, mp3, tmp_path


def want_for(oracle, f):
    """oracle result for a file: MP3s through the library's decoder, the M4A stand-in through its decoder script"""
    ch = decoded_mp3(f) if str(f).endswith(".mp3") else decoded(Path(f).name)
    res, hist = oracle.analyze_pcm(ch[0], ch[1] if len(ch) > 1 else None, 44100)
    return res, hist


def test_track_gain_applies_to_mp3(box, oracle):
    from mp3rgain_amd import mp3gain

    run, mp3, _ = box
    f = mp3("quiet.mp3")
    w, _ = want_for(oracle, f)
    steps = w["gain_steps"]
    assert steps > 0
    b = mp3gain.analyze(f)
    rc, out, err = run("-r", f)
    assert (rc, err) == (0, "")
    assert out == ("mp3rgain Analyzing and applying track gain to 1 file(s)\n  Target: 89 dB (ReplayGain 1.0)\n\n"
                   "  -> Analyzing quiet.mp3...\n"
                   f"      Loudness: {w['loudness_db']:.1f} dB, Gain: {w['gain_db']:+.1f} dB ({steps} steps), Peak: {w['peak']:.4f}\n"
                   f"  v quiet.mp3 (40 frames, {steps * 1.5:+.1f} dB)\n")  # src/main.rs:1223-1238, 1956-1969, 2139-2147
    a = mp3gain.analyze(f)
    assert (a.min_gain, a.max_gain) == (min(b.min_gain + steps, 255), min(b.max_gain + steps, 255))
    assert mp3gain.read_ape_tag_value(f, "MP3GAIN_UNDO") == f"{steps:+04d},{steps:+04d},N"
    # the analysis of the patched file moves by the applied gain: a global_gain step is a factor 2^(1/4) in amplitude,
    # 1.505 dB (the "1.5 dB" of the tool's messages is the nominal figure)
    w2, _ = want_for(oracle, f)
    assert abs((w2["loudness_db"] - w["loudness_db"]) - steps * 20 * math.log10(2 ** 0.25)) <= 0.02
    # undo brings the audio bytes back
    rc, out, _ = run("-u", f)
    assert "(40 frames restored)" in out and mp3gain.analyze(f).max_gain == b.max_gain


def test_track_gain_modifier_dry_run_json_and_clipping(box, oracle):
    from mp3rgain_amd import mp3gain

    run, mp3, _ = box
    f = mp3("quiet.mp3")
    w, _ = want_for(oracle, f)
    b = mp3gain.analyze(f)
    rc, out, _ = run("-n", "-e", "-m", "-2", f)  # -e = track gain only (src/main.rs:527-530)
    s = w["gain_steps"]
    assert f"({s} steps + -2 = {s - 2}), Peak:" in out and "  Gain modifier: -2 steps\n" in out
    assert f"  ~ [DRY RUN] quiet.mp3 (would apply {(s - 2) * 1.5:+.1f} dB, {s - 2} steps)\n" in out
    assert mp3gain.analyze(f).max_gain == b.max_gain
    rc, out, err = run("-o", "json", "-r", "-n", f)
    d = json.loads(out)
    r = d["files"][0]
    assert r["status"] == "dry_run" and r["loudness_db"] == w["loudness_db"] and r["peak"] == w["peak"]
    assert r["gain_applied_steps"] == s and r["gain_applied_db"] == s * 1.5 and d["summary"]["dry_run"] is True
    # a loud, clipped track gets negative gain: no clipping logic; a quiet one with peak*gain > 1 trips -k (:2033-2058)
    new_peak = w["peak"] * 10 ** (w["gain_db"] / 20)
    rc, out, err = run("-n", "-r", "-k", f)
    if new_peak > 1.0:
        safe = max(round(-20 * math.log10(w["peak"]) / 1.5), 0)
        assert f"gain reduced from {s} to {safe} steps to prevent clipping (peak: {w['peak']:.4f})" in err
    else:
        assert err == ""
    loud = mp3("loud.mp3")
    wl, _ = want_for(oracle, loud)
    assert wl["gain_steps"] < 0 and wl["peak"] > 1.0  # a float decode is not clipped: the peak says by how much it would be
    bl = mp3gain.analyze(loud)
    rc, out, err = run("-r", loud)
    assert err == "" and mp3gain.analyze(loud).max_gain == bl.max_gain + wl["gain_steps"]


def test_album_gain(box, oracle):
    from mp3rgain_amd import mp3gain

    run, mp3, _ = box
    names = ["one.mp3", "quiet.mp3", "three.mp3"]
    files = [mp3(n, s) for n, s in zip(names, ("test_joint_stereo.mp3", "test_vbr.mp3", "test_joint_stereo.mp3"))]
    per = [want_for(oracle, f) for f in files]
    alb, _ = oracle.album_from_hists([h for _, h in per], [w["peak"] for w, _ in per])
    steps = round(alb["album_gain_db"] / 1.5)
    before = [mp3gain.analyze(f).max_gain for f in files]
    rc, out, err = run("-a", "-c", *files)
    assert (rc, err) == (0, "")
    assert out.startswith("mp3rgain Analyzing album gain for 3 file(s)\n  Target: 89 dB (ReplayGain 1.0)\n\n  -> Analyzing tracks...\n\n"
                          f"  Album loudness: {alb['album_loudness_db']:.1f} dB\n"
                          f"  Album gain:     {alb['album_gain_db']:+.1f} dB ({steps} steps)\n"
                          f"  Album peak:     {alb['album_peak']:.4f}\n\n")  # src/main.rs:1297-1334
    if steps != 0:
        for f, b in zip(files, before):
            assert mp3gain.analyze(f).max_gain == max(0, min(255, b + steps))
            assert f"  v {f.name} (" in out
    # the files now carry the album gain in their global_gain fields, and the decoder reads those: the second analysis
    # sees every track `steps` global_gain steps (2^(1/4) in amplitude each) louder
    per2 = [want_for(oracle, f) for f in files]
    alb2, _ = oracle.album_from_hists([h for _, h in per2], [w["peak"] for w, _ in per2])
    if steps != 0:
        assert abs((alb2["album_loudness_db"] - alb["album_loudness_db"]) - steps * 20 * math.log10(2 ** 0.25)) <= 0.03
    per, alb, steps = per2, alb2, round(alb2["album_gain_db"] / 1.5)
    rc, out, _ = run("-o", "json", "-n", "-a", *files)
    d = json.loads(out)
    assert d["album"] == {"loudness_db": alb["album_loudness_db"], "gain_db": alb["album_gain_db"], "gain_steps": steps, "peak": alb["album_peak"]}
    assert [r["loudness_db"] for r in d["files"]] == [w["loudness_db"] for w, _ in per]
    # a file that cannot be decoded aborts the album (:1436-1452)
    rc, out, err = run("-a", files[0], files[0].parent / "missing.mp3")
    assert rc == 1 and err.startswith("error: Failed to analyze album: Failed to open: ")


def test_beets_tsv_and_max_amplitude(box, oracle):
    run, mp3, tmp = box
    f = mp3("one.mp3")
    w, _ = want_for(oracle, f)
    rc, out, err = run("-o", "-s", "s", "-k", "-d", "0", f)  # what beets runs
    assert (rc, err) == (0, "")
    lines = out.splitlines()
    assert lines[0] == "File\tMP3 gain\tdB gain\tMax Amplitude\tMax global_gain\tMin global_gain"  # src/main.rs:1123
    assert lines[1] == f"one.mp3\t{w['gain_steps']}\t{w['gain_db']:.6f}\t{w['peak'] * 32768.0:.6f}\t210\t110"  # :1719-1722
    rc, out, _ = run("-o", "tsv", "-d", "3.0", f)  # -d shifts the suggested gain (:1711-1713)
    g = w["gain_db"] + 3.0
    assert out.splitlines()[1].split("\t")[1:3] == [str(round(g / 1.5)), f"{g:.6f}"]
    # -x, src/main.rs:583-689
    head = -20.0 * math.log10(w["peak"])
    rc, out, _ = run("-x", f)
    assert out == ("mp3rgain Finding maximum amplitude for 1 file(s)\n\none.mp3\n"
                   f"  Max PCM sample: {w['peak'] * 32768.0:.6f}\n" + ("    (may be clipped - actual peak could be higher)\n" if w["peak"] >= 0.9999 else "") +
                   f"  Headroom:       {head:+.2f} dB\n  Max global_gain: 210\n  Min global_gain: 110\n\n")
    rc, out, _ = run("-x", "-q", f)
    assert out == f"one.mp3\t{w['peak'] * 32768.0:.6f}\t{head:.2f}\n"
    rc, out, _ = run("-x", "-o", "json", f)
    r = json.loads(out)["files"][0]
    assert r["max_amplitude"] == w["peak"] * 32768.0 and r["max_gain"] == 210 and r["min_gain"] == 110
    # a WAV file needs no decoder; the global_gain columns are whatever the MP3 frame scanner makes of PCM bytes
    # (false syncs, or the 255/0 fall-back of :1707-1708), as they would be in the reference
    wav = tmp / "plain.wav"
    ch = test_signal("s16", 44100, 50000, 2, 5)
    wav.write_bytes(wav_bytes(ch, 44100, "s16"))
    ww, _ = oracle.analyze_pcm(np.asarray(ch[0], np.int16), np.asarray(ch[1], np.int16), 44100)
    rc, out, _ = run("-o", "tsv", wav)
    assert out.splitlines()[1].split("\t")[:4] == ["plain.wav", str(ww["gain_steps"]), f"{ww['gain_db']:.6f}", f"{ww['peak'] * 32768.0:.6f}"]


def test_m4a_gets_tags_only(box, oracle):
    from mp3rgain_amd import mp4meta

    sys.path.insert(0, str(Path(__file__).resolve().parent))
    from test_mp4meta import make_mp4

    run, _, tmp = box
    f = tmp / "quiet.m4a"
    data, _ = make_mp4()
    f.write_bytes(data)
    w, _ = want_for(oracle, f)
    rc, out, err = run("-r", "-c", f)
    assert (rc, err) == (0, "")
    assert out.endswith(f"  v quiet.m4a (tags written, {w['gain_db']:+.1f} dB)\n")  # src/main.rs:2204-2212
    t = mp4meta.read_replaygain_tags(f)
    assert (t.track_gain, t.track_peak, t.album_gain) == ("%+.2f dB" % w["gain_db"], "%.6f" % w["peak"], None)
    rc, out, _ = run("-n", "-r", f)
    assert "(tags only)" in out or "no adjustment needed" in out


def test_track_gain_over_several_files_is_one_batch_with_the_same_output(box, oracle):
    """-r on a list of files analyses them as one GPU batch (rg_analyze_tracks); what is printed, applied and reported
    is what -r prints file by file -- including a file that cannot be opened in the middle of the list."""
    from mp3rgain_amd import mp3gain

    run, mp3, tmp = box
    a, b, c = mp3("one.mp3"), mp3("quiet.mp3", "test_vbr.mp3"), mp3("three.mp3", "test_mono.mp3")
    missing = tmp / "missing.mp3"
    rc, out, err = run("-n", "-r", a, missing, b, c)          # dry run: nothing is written
    assert rc == 0
    singles = [run("-n", "-r", f) for f in (a, missing, b, c)]
    head = "[DRY RUN] mp3rgain Analyzing and would apply track gain to {} file(s)\n  Target: 89 dB (ReplayGain 1.0)\n\n"
    tail = "\nNo files were modified.\n"
    body = lambda text, n: text[len(head.format(n)):-len(tail)]  # noqa: E731
    assert out.startswith(head.format(4)) and out.endswith(tail) and all(o.startswith(head.format(1)) and o.endswith(tail) for _, o, _ in singles)
    assert body(out, 4) == "".join(body(o, 1) for _, o, _ in singles)
    assert err == "".join(e for _, _, e in singles) and "Failed to open" in err
    # json: the same per-file records
    rc, out, _ = run("-n", "-r", "-o", "json", a, missing, b, c)
    recs = json.loads(out)["files"]
    one = [json.loads(run("-n", "-r", "-o", "json", f)[1])["files"][0] for f in (a, missing, b, c)]
    assert recs == one and recs[1]["status"] == "error"
    # and for real: every file gets its own gain
    before = [mp3gain.analyze(f).max_gain for f in (a, b, c)]
    wants = [want_for(oracle, f)[0]["gain_steps"] for f in (a, b, c)]
    rc, out, err = run("-r", "-c", a, b, c)
    assert rc == 0
    after = [mp3gain.analyze(f).max_gain for f in (a, b, c)]
    assert after == [max(0, min(255, x + w)) for x, w in zip(before, wants)]


def test_the_module_runs_as_a_process_of_its_own_without_pytorch(box, tmp_path):
    """`python -m mp3rgain_amd` sets MP3RGAIN_AMD_STANDALONE: the library is loaded on the system's HIP runtime, PyTorch is
    never imported (the process starts in 0.4 s instead of 1.4 s), and the output is that of the in-process call."""
    import os
    import subprocess

    run, mp3, tmp = box
    a, b = mp3("one.mp3", "test_vbr.mp3"), mp3("two.mp3", "test_mono.mp3")
    rc, want, _ = run("-r", "--dry-run", a, b)
    assert rc == 0
    code = ("import sys, runpy\n"
            "sys.argv = ['mp3rgain_amd'] + sys.argv[1:]\n"
            "try:\n    runpy.run_module('mp3rgain_amd', run_name='__main__')\n"
            "except SystemExit as ex:\n    rc = ex.code or 0\n"
            "print('torch imported:', 'torch' in sys.modules)\nsys.exit(rc)\n")
    env = {k: v for k, v in os.environ.items() if k != "MP3RGAIN_AMD_STANDALONE"}
    env["PYTHONPATH"] = str(Path(__file__).resolve().parent.parent) + os.pathsep + env.get("PYTHONPATH", "")
    p = subprocess.run([sys.executable, "-c", code, "-r", "--dry-run", str(a), str(b)], capture_output=True, text=True, env=env, timeout=120)
    assert p.returncode == 0, p.stderr
    assert "torch imported: False" in p.stdout
    assert p.stdout.replace("torch imported: False\n", "") == want
