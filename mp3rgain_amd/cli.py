"""Command-line façade: mp3rgain's option set and output lines over this
library -- `python -m mp3rgain_amd [OPTIONS] <FILES>...`.

Behaviour follows mp3rgain v1.5.0 src/main.rs (function by function, cited below); colours and the progress
bar are left out (the reference drops both when stdout is not a terminal).  ReplayGain analysis (-r, -a, -e,
-x, TSV info) runs on the GPU through the C ABI; it needs decoded audio, so an analysed file is a WAV file, or
any file when a decoder command is given (`--decoder "ffmpeg -v error -i {} -f wav -c:a pcm_f32le -"` or the
MP3RGAIN_AMD_DECODER environment variable).  Gain changes, undo and tags are the lossless byte work of
`mp3gain` / `mp4meta` and need no decoder.
"""
from __future__ import annotations

import json
import math
import os
import shutil
import sys
from dataclasses import dataclass, field
from pathlib import Path
from typing import List, Optional, Tuple

from . import mp3gain, mp4meta
from . import replaygain as rgmod

VERSION = "0.1.0"
GAIN_STEP_DB = 1.5
REFERENCE_DB = 89.0


class CliError(Exception):
    """An anyhow error that leaves main(): printed as `Error: ...`, exit status 1."""


class Exit(Exception):
    def __init__(self, code: int):
        self.code = code


@dataclass
class Options:  # src/main.rs:65-96
    gain_steps: Optional[int] = None
    gain_modifier_db: float = 0.0
    channel_gain: Optional[Tuple[int, int]] = None
    gain_modifier: int = 0
    undo: bool = False
    stored_tag_mode: str = "none"  # none | check | delete | skip | recalc | id3v2 | apev2
    track_gain: bool = False
    album_gain: bool = False
    skip_album: bool = False
    max_amplitude_only: bool = False
    track_index: Optional[int] = None
    preserve_timestamp: bool = False
    ignore_clipping: bool = False
    prevent_clipping: bool = False
    quiet: bool = False
    recursive: bool = False
    dry_run: bool = False
    output_format: str = "text"  # text | json | tsv
    wrap_gain: bool = False
    use_temp_file: bool = False
    assume_mpeg2: bool = False
    decoder: Optional[str] = None  # this façade only
    files: List[Path] = field(default_factory=list)


def _rust_float(x: float) -> str:
    """`{}` of an f64: 89.0 -> "89", 1.5 -> "1.5"."""
    return str(int(x)) if x == int(x) and abs(x) < 1e15 else repr(x)


def _parse_int(s: str, what: str, bits: int = 32, signed: bool = True) -> int:
    try:
        if s.strip() != s or s.startswith("+-") or "_" in s:
            raise ValueError
        v = int(s, 10)
    except ValueError:
        raise CliError(f"{what}: {s}")
    lo, hi = (-(1 << (bits - 1)), (1 << (bits - 1)) - 1) if signed else (0, (1 << bits) - 1)
    if not lo <= v <= hi:
        raise CliError(f"{what}: {s}")
    return v


def _parse_float(s: str, what: str) -> float:
    try:
        if s.strip() != s or "_" in s:
            raise ValueError
        return float(s)
    except ValueError:
        raise CliError(f"{what}: {s}")


def parse_args(args: List[str], out, err) -> Options:
    """parse_args, src/main.rs:183-434."""
    o = Options()
    i = 0

    def need(flag: str, msg: str) -> str:
        nonlocal i
        i += 1
        if i >= len(args):
            print(f"error: {msg}", file=err)
            raise Exit(1)
        return args[i]

    while i < len(args):
        arg = args[i]
        if arg == "--dry-run":
            o.dry_run = True
        elif arg == "--help":
            print_usage(out)
            raise Exit(0)
        elif arg == "--version":
            print_version(out)
            raise Exit(0)
        elif arg == "--decoder":  # not in the reference: see the module docstring
            o.decoder = need("--decoder", "--decoder requires an argument")
        elif arg.startswith("-") and len(arg) > 1 and not arg.startswith("--"):
            flag = arg[1:]
            if flag == "g":
                o.gain_steps = _parse_int(need("g", "-g requires an argument"), "invalid gain value")
            elif flag == "d":
                o.gain_modifier_db = _parse_float(need("d", "-d requires an argument"), "invalid dB value")
            elif flag == "m":
                o.gain_modifier = _parse_int(need("m", "-m requires an argument"), "invalid modifier value")
            elif flag == "s":
                mode = need("s", "-s requires an argument")
                table = {"c": "check", "d": "delete", "s": "skip", "r": "recalc", "i": "id3v2", "a": "apev2"}
                if mode not in table:
                    print(f"error: unknown -s mode '{mode}', use c/d/s/r/i/a", file=err)
                    raise Exit(1)
                o.stored_tag_mode = table[mode]
                if mode == "i":
                    print("warning: -s i (ID3v2 tags) not fully supported, using APEv2", file=err)
            elif flag == "o":
                # mp3gain compatibility: a bare -o means TSV
                if i + 1 < len(args) and args[i + 1].lower() in ("json", "text", "tsv", "db"):
                    i += 1
                    o.output_format = {"json": "json", "text": "text", "tsv": "tsv", "db": "tsv"}[args[i].lower()]
                else:
                    o.output_format = "tsv"
            elif flag == "l":
                two = "-l requires two arguments: <channel> <gain>"
                a = need("l", two)
                try:
                    ch = _parse_int(a, "", 64, signed=False)
                except CliError:
                    raise CliError(f"invalid channel number: {a} (use 0 for left, 1 for right)")
                if ch not in (0, 1):
                    raise CliError(f"invalid channel: {ch} (use 0 for left, 1 for right)")
                o.channel_gain = (ch, _parse_int(need("l", two), "invalid gain value"))
            elif flag == "r":
                o.track_gain = True
            elif flag == "a":
                o.album_gain = True
            elif flag == "e":
                o.skip_album = True
            elif flag == "x":
                o.max_amplitude_only = True
            elif flag == "i":
                o.track_index = _parse_int(need("i", "-i requires an argument"), "invalid track index", 32, signed=False)
            elif flag == "u":
                o.undo = True
            elif flag == "p":
                o.preserve_timestamp = True
            elif flag == "c":
                o.ignore_clipping = True
            elif flag == "k":
                o.prevent_clipping = True
            elif flag == "q":
                o.quiet = True
            elif flag == "R":
                o.recursive = True
            elif flag == "n":
                o.dry_run = True
            elif flag == "w":
                o.wrap_gain = True
            elif flag == "t":
                o.use_temp_file = True
            elif flag == "f":
                o.assume_mpeg2 = True
            elif flag in ("v", "-version"):
                print_version(out)
                raise Exit(0)
            elif flag in ("h", "-help"):
                print_usage(out)
                raise Exit(0)
            elif all(c in "pqckuranRewxtf" for c in flag):  # combined short flags: -qp, -kc ...
                names = {"p": "preserve_timestamp", "q": "quiet", "c": "ignore_clipping", "k": "prevent_clipping",
                         "u": "undo", "r": "track_gain", "a": "album_gain", "n": "dry_run", "R": "recursive",
                         "e": "skip_album", "w": "wrap_gain", "x": "max_amplitude_only", "t": "use_temp_file",
                         "f": "assume_mpeg2"}
                for c in flag:
                    setattr(o, names[c], True)
            elif flag.startswith("g"):  # attached values: -g2, -d4.5, -m2, -i1
                o.gain_steps = _parse_int(flag[1:], "invalid gain value")
            elif flag.startswith("d"):
                o.gain_modifier_db = _parse_float(flag[1:], "invalid dB value")
            elif flag.startswith("m"):
                o.gain_modifier = _parse_int(flag[1:], "invalid modifier value")
            elif flag.startswith("i"):
                o.track_index = _parse_int(flag[1:], "invalid track index", 32, signed=False)
            else:
                print(f"warning: unknown option: -{flag}", file=err)
        elif not arg.startswith("--"):
            o.files.append(Path(arg))
        i += 1
    return o


# ---- JSON shapes (src/main.rs:101-165): field order of the structs, None fields skipped -----------------------
_FILE_KEYS = ("file", "status", "frames", "mpeg_version", "channel_mode", "min_gain", "max_gain", "avg_gain",
              "headroom_steps", "headroom_db", "gain_applied_steps", "gain_applied_db", "loudness_db", "peak",
              "max_amplitude", "error", "warning", "dry_run")


def _is_riff_wave(file) -> bool:
    try:
        with open(file, "rb") as f:
            head = f.read(12)
    except OSError:
        return False
    return len(head) == 12 and head[:4] == b"RIFF" and head[8:] == b"WAVE"


def _file_result(file: Path, **kw) -> dict:
    d = {"file": str(file)}
    for k in _FILE_KEYS[1:]:
        if kw.get(k) is not None:
            d[k] = kw[k]
    return d


def _summary(total: int, ok: int, failed: int, dry: bool) -> dict:
    d = {"total_files": total, "successful": ok, "failed": failed}
    if dry:
        d["dry_run"] = True
    return d


def _finite(x):
    """serde_json writes non-finite floats as null."""
    if isinstance(x, float) and not math.isfinite(x):
        return None
    if isinstance(x, dict):
        return {k: _finite(v) for k, v in x.items()}
    if isinstance(x, list):
        return [_finite(v) for v in x]
    return x


def _print_json(out, files=None, album=None, summary=None):
    files, album = _finite(files), _finite(album)
    d = {}
    if files is not None:
        d["files"] = files
    if album is not None:
        d["album"] = album
    if summary is not None:
        d["summary"] = summary
    print(json.dumps(d, indent=2), file=out)


def _count(result: dict, tally: List[int]):
    if result.get("status") == "success":
        tally[0] += 1
    elif result.get("status") == "error":
        tally[1] += 1


def _name(p: Path) -> str:
    return p.name or "unknown"


def _file_identity(f):
    """What makes two command-line arguments the same file: device + inode (a symlink, ./x and x, a relative and an
    absolute path all name one file); the resolved path when it cannot be stat'ed."""
    try:
        st = os.stat(f)
        return (st.st_dev, st.st_ino)
    except OSError:
        return os.path.realpath(os.fspath(f))


class Cli:
    def __init__(self, opts: Options, out, err):
        self.o, self.out, self.err = opts, out, err
        self._an: Optional[rgmod.Node] = None
        self._batch: Optional[dict] = None  # results of the one batched analysis of all files (analyze_track)

    # The GPU contexts are created on first use: byte-level commands never touch a device.  The analysis runs on a node
    # (include/mp3rgain_amd_node.h): every visible GPU of the machine, one context each -- `-a` and `-r` over many files
    # deal the files out over all of them, a single file runs on the first.  MP3RGAIN_AMD_DEVICES="0,2" restricts it.
    def analyzer(self) -> rgmod.Node:
        if self._an is None:
            devs = os.environ.get("MP3RGAIN_AMD_DEVICES")
            if devs:
                devices = [int(d) for d in devs.split(",")]
            elif len(self.o.files) <= 1:
                devices = [0]  # one file runs on one GPU: no contexts (streams, tables, pinned buffers) on the others, and a
                               # GPU that cannot be opened elsewhere on the machine does not fail a single-file command
            else:
                devices = None  # every visible GPU
            self._an = rgmod.Node(devices)
            dec = self.o.decoder or os.environ.get("MP3RGAIN_AMD_DECODER")
            if dec:
                self._an.set_decoder_command(dec)
        return self._an

    def analyze_track(self, file):
        """analyze_track for one file of the command line.  With several files the whole list is analysed once, as one GPU
        batch (rg_analyze_tracks: the files load on all host cores, decode and analysis run on the device), and every file
        then finds its result -- or the error it would have raised -- here; the reference analyses file by file
        (src/main.rs:1937-2001), the results are the same."""
        files = self.o.files
        # (a file named twice is analysed again after its first occurrence has been patched, as in the reference: no batch)
        if len(files) > 1 and len({_file_identity(f) for f in files}) == len(files):
            if self._batch is None:
                res = self.analyzer().analyze_track_files(files, self.o.track_index)
                self._batch = {os.fspath(f): r for f, r in zip(files, res)}
            r = self._batch.get(os.fspath(file))
            if r is not None:
                if isinstance(r, rgmod.ReplayGainError):
                    raise r
                return r
        return self.analyzer().analyze_track_file(file, self.o.track_index)

    def p(self, *a):
        print(*a, file=self.out)

    def e(self, *a):
        print(*a, file=self.err)

    @property
    def text(self) -> bool:
        return self.o.output_format == "text"

    @property
    def talk(self) -> bool:
        return self.text and not self.o.quiet

    # ---- run, src/main.rs:472-540 ------------------------------------------------------------------------
    def run(self) -> int:
        o = self.o
        if not o.files:
            self.e("error: no files specified")
            return 1
        if o.recursive:
            o.files = expand_files_recursive(o.files)
            if not o.files:
                self.e("error: no audio files found (MP3/M4A)")
                return 1
        if o.assume_mpeg2 and self.talk:
            self.e("note: -f (assume MPEG2) is accepted for compatibility but has no effect")
        if o.max_amplitude_only:
            return self.cmd_max_amplitude()
        if o.stored_tag_mode == "delete":
            return self.cmd_delete_tags()
        if o.stored_tag_mode == "check":
            return self.cmd_check_tags()
        if o.undo:
            return self.cmd_undo()
        if o.album_gain and not o.skip_album:
            return self.cmd_album_gain()
        if o.track_gain or o.skip_album:
            return self.cmd_track_gain()
        if o.channel_gain is not None:
            return self.cmd_apply_channel(*o.channel_gain)
        if o.gain_steps is not None:
            return self.cmd_apply(o.gain_steps)
        return self.cmd_info()

    # ---- find_max_amplitude, src/lib.rs:1174-1199 ------------------------------------------------------------
    def find_max_amplitude(self, file: Path):
        info = mp3gain.analyze(file)  # "No valid MP3 frames found" for anything that is not MP3
        peak = self.analyzer().find_peak_amplitude_file(file).peak
        return peak, info.max_gain, info.min_gain

    # ---- -x, src/main.rs:583-689 -------------------------------------------------------------------------------
    def cmd_max_amplitude(self) -> int:
        o = self.o
        if self.talk:
            self.p(f"mp3rgain Finding maximum amplitude for {len(o.files)} file(s)")
            self.p()
        results = []
        for file in o.files:
            name = _name(file)
            try:
                max_amp, max_gain, min_gain = self.find_max_amplitude(file)
            except (mp3gain.Mp3GainError, rgmod.ReplayGainError) as ex:
                if o.output_format == "json":
                    results.append(_file_result(file, status="error", error=str(ex)))
                elif not o.quiet:
                    self.e(f"{name} - {ex}")
                continue
            pcm = max_amp * 32768.0
            headroom = -20.0 * math.log10(max_amp) if max_amp > 0.0 else math.inf
            may_clip = file.suffix.lower() == ".mp3" and max_amp >= 0.9999
            if o.output_format == "text":
                if not o.quiet:
                    self.p(name)
                    self.p(f"  Max PCM sample: {pcm:.6f}")
                    if may_clip:
                        self.p("    (may be clipped - actual peak could be higher)")
                    self.p(f"  Headroom:       {_signed(headroom, 2)} dB")
                    self.p(f"  Max global_gain: {max_gain}")
                    self.p(f"  Min global_gain: {min_gain}")
                    self.p()
                else:
                    self.p(f"{name}\t{pcm:.6f}\t{_plain(headroom, 2)}")
            elif o.output_format == "tsv":
                self.p(f"{name}\t{pcm:.6f}\t{_plain(headroom, 2)}\t{max_gain}\t{min_gain}")
            else:
                results.append(_file_result(file, max_amplitude=pcm, headroom_db=headroom, max_gain=max_gain, min_gain=min_gain,
                                            warning="peak may be clipped - actual value could be higher" if may_clip else None))
        if o.output_format == "json":
            _print_json(self.out, files=results)
        return 0

    # ---- -s d, src/main.rs:691-794 -----------------------------------------------------------------------------
    def cmd_delete_tags(self) -> int:
        o = self.o
        pre = "[DRY RUN] " if o.dry_run else ""
        if self.talk:
            self.p(f"{pre}mp3rgain {'Would delete' if o.dry_run else 'Deleting'} ReplayGain tags from {len(o.files)} file(s)")
            self.p()
        results, tally = [], [0, 0]
        for file in o.files:
            name = _name(file)
            if o.dry_run:
                if self.talk:
                    self.p(f"  ~ [DRY RUN] {name} (would delete tags)")
                results.append(_file_result(file, status="dry_run", dry_run=True))
                continue
            mtime = _mtime(file) if o.preserve_timestamp else None
            try:
                mp3gain.delete_ape_tag(file)
            except mp3gain.Mp3GainError as ex:
                if self.talk:
                    self.e(f"  x {name} - {ex}")
                tally[1] += 1
                results.append(_file_result(file, status="error", error=str(ex)))
                continue
            _restore(file, mtime)
            if self.talk:
                self.p(f"  v {name} (tags deleted)")
            tally[0] += 1
            results.append(_file_result(file, status="success"))
        if o.output_format == "json":
            _print_json(self.out, files=results, summary=_summary(len(o.files), tally[0], tally[1], o.dry_run))
        elif o.dry_run and not o.quiet:
            self.p()
            self.p("No files were modified.")
        return 0

    # ---- -s c, src/main.rs:796-917 -----------------------------------------------------------------------------
    def cmd_check_tags(self) -> int:
        o = self.o
        if self.talk:
            self.p(f"mp3rgain Checking stored tag info for {len(o.files)} file(s)")
            self.p()
        results = []
        keys = ("MP3GAIN_UNDO", "MP3GAIN_MINMAX", "REPLAYGAIN_TRACK_GAIN", "REPLAYGAIN_TRACK_PEAK", "REPLAYGAIN_ALBUM_GAIN",
                "REPLAYGAIN_ALBUM_PEAK")
        for file in o.files:
            name = _name(file)
            try:
                has_tag = mp3gain.has_ape_tag(file)
                vals = [mp3gain.read_ape_tag_value(file, k) for k in keys] if has_tag else []
            except (mp3gain.Mp3GainError, OSError) as ex:
                if o.output_format != "json":
                    self.e(f"{name} - {ex}")
                else:
                    results.append(_file_result(file, status="error", error=str(ex)))
                continue
            if not has_tag:
                if self.text:
                    self.p(name)
                    self.p("  (no APE tag found)")
                    self.p()
                elif o.output_format == "tsv":
                    self.p(f"{name}\t-\t-\t-\t-\t-\t-")
                else:
                    results.append(_file_result(file, status="no_tag"))
                continue
            if self.text:
                self.p(name)
                labels = ("  MP3GAIN_UNDO:         ", "  MP3GAIN_MINMAX:       ", "  REPLAYGAIN_TRACK_GAIN: ", "  REPLAYGAIN_TRACK_PEAK: ",
                          "  REPLAYGAIN_ALBUM_GAIN: ", "  REPLAYGAIN_ALBUM_PEAK: ")
                for lab, v in zip(labels, vals):
                    if v is not None:
                        self.p(f"{lab}{v}")
                if vals[0] is None and vals[1] is None and vals[2] is None:
                    self.p("  (no mp3gain tags found)")
                self.p()
            elif o.output_format == "tsv":
                self.p("\t".join([name] + [v if v is not None else "-" for v in vals]))
            else:
                results.append(_file_result(file, status="success"))
        if o.output_format == "json":
            _print_json(self.out, files=results)
        return 0

    # ---- -g, src/main.rs:948-1033 and 1488-1616 ------------------------------------------------------------------
    def cmd_apply(self, steps: int) -> int:
        o = self.o
        if steps == 0:
            if o.output_format == "json":
                _print_json(self.out, files=[], summary=_summary(len(o.files), 0, 0, o.dry_run))
            elif not o.quiet:
                self.p("info: gain is 0, nothing to do")
            return 0
        db = steps * GAIN_STEP_DB
        pre = "[DRY RUN] " if o.dry_run else ""
        if self.talk:
            self.p(f"{pre}mp3rgain {'Would apply' if o.dry_run else 'Applying'} {steps} step(s) ({db:+.1f} dB) to {len(o.files)} file(s)")
            if o.wrap_gain:
                self.p("  ! Wrap mode enabled")
            self.p()
        results, tally = [], [0, 0]
        for file in o.files:
            r = self.process_apply(file, steps)
            _count(r, tally)
            if o.output_format == "tsv":
                try:
                    info = mp3gain.analyze(file)
                    self.p(f"{_name(file)}\t{steps}\t{db:.1f}\t{1.0:.6f}\t{info.max_gain}\t{info.min_gain}")
                except mp3gain.Mp3GainError:
                    pass
            if o.output_format == "json":
                results.append(r)
        if o.output_format == "json":
            _print_json(self.out, files=results, summary=_summary(len(o.files), tally[0], tally[1], o.dry_run))
        else:
            self._dry_run_notice()
        return 0

    def _dry_run_notice(self):  # src/main.rs:941-946
        if self.o.dry_run and self.talk:
            self.p()
            self.p("No files were modified.")

    def _with_temp_file(self, file: Path, op):  # apply_with_temp_file, src/main.rs:1458-1486
        if not self.o.use_temp_file:
            return op(file)
        tmp = (file.parent if str(file.parent) else Path(".")) / f".mp3rgain_temp_{os.getpid()}.mp3"
        shutil.copyfile(file, tmp)
        try:
            frames = op(tmp)
        except Exception:
            try:
                tmp.unlink()
            except OSError:
                pass
            raise
        os.replace(tmp, file)
        return frames

    def process_apply(self, file: Path, steps: int) -> dict:
        o = self.o
        name = _name(file)
        pre = "[DRY RUN] " if o.dry_run else ""
        mtime = _mtime(file) if o.preserve_timestamp and not o.dry_run else None
        actual, warning = steps, None
        if steps > 0 and not o.wrap_gain:
            try:
                info = mp3gain.analyze(file)
            except mp3gain.Mp3GainError:
                info = None
            if info is not None and steps > info.headroom_steps:
                if o.prevent_clipping:
                    actual = info.headroom_steps
                    if self.talk:
                        self.e(f"  ! {pre}{name} - gain reduced from {steps} to {actual} steps to prevent clipping")
                    warning = f"gain reduced from {steps} to {actual} steps to prevent clipping"
                elif not o.ignore_clipping and not o.quiet:
                    if self.text:
                        self.e(f"  ! {pre}{name} - clipping warning: requested {steps} steps but only {info.headroom_steps} headroom")
                        self.e("      Use -c to ignore clipping warnings or -k to prevent clipping")
                    warning = f"clipping warning: requested {steps} steps but only {info.headroom_steps} headroom"
        if o.dry_run:
            if self.talk:
                self.p(f"  ~ [DRY RUN] {name} (would apply {actual} steps)")
            return _file_result(file, status="dry_run", gain_applied_steps=actual, gain_applied_db=actual * GAIN_STEP_DB,
                                warning=warning, dry_run=True)
        if o.stored_tag_mode == "skip":  # -s s: no undo tag
            fn = mp3gain.apply_gain_wrap if o.wrap_gain else mp3gain.apply_gain
        else:
            fn = mp3gain.apply_gain_with_undo_wrap if o.wrap_gain else mp3gain.apply_gain_with_undo
        try:
            frames = self._with_temp_file(file, lambda f: fn(f, actual))
        except (mp3gain.Mp3GainError, OSError) as ex:
            if self.talk:
                self.e(f"  x {name} - {ex}")
            return _file_result(file, status="error", error=str(ex))
        _restore(file, mtime)
        if self.talk:
            self.p(f"  v {name} ({frames} frames)")
        return _file_result(file, status="success", frames=frames, gain_applied_steps=actual,
                            gain_applied_db=actual * GAIN_STEP_DB, warning=warning)

    # ---- -l, src/main.rs:1035-1118 and 1618-1697 -----------------------------------------------------------------
    def cmd_apply_channel(self, channel: int, steps: int) -> int:
        o = self.o
        if steps == 0:
            if o.output_format == "json":
                _print_json(self.out, files=[], summary=_summary(len(o.files), 0, 0, o.dry_run))
            elif not o.quiet:
                self.p("info: gain is 0, nothing to do")
            return 0
        db = steps * GAIN_STEP_DB
        cname = "left" if channel == 0 else "right"
        pre = "[DRY RUN] " if o.dry_run else ""
        if self.talk:
            self.p(f"{pre}mp3rgain {'Would apply' if o.dry_run else 'Applying'} {steps} step(s) ({db:+.1f} dB) to {cname} channel of {len(o.files)} file(s)")
            self.p()
        results, tally = [], [0, 0]
        for file in o.files:
            name = _name(file)
            if o.dry_run:
                if self.talk:
                    self.p(f"  ~ [DRY RUN] {name} (would apply {steps} steps to {cname} channel)")
                r = _file_result(file, status="dry_run", gain_applied_steps=steps, gain_applied_db=db, dry_run=True)
            else:
                mtime = _mtime(file) if o.preserve_timestamp else None
                try:
                    frames = mp3gain.apply_gain_channel_with_undo(file, mp3gain.Channel(channel), steps)
                    _restore(file, mtime)
                    if self.talk:
                        self.p(f"  v {name} ({frames} frames, {cname} channel)")
                    r = _file_result(file, status="success", frames=frames, gain_applied_steps=steps, gain_applied_db=db)
                except mp3gain.Mp3GainError as ex:
                    if self.talk:
                        self.e(f"  x {name} - {ex}")
                    r = _file_result(file, status="error", error=str(ex))
            _count(r, tally)
            if o.output_format == "json":
                results.append(r)
        if o.output_format == "json":
            _print_json(self.out, files=results, summary=_summary(len(o.files), tally[0], tally[1], o.dry_run))
        else:
            self._dry_run_notice()
        return 0

    # ---- default command, src/main.rs:1120-1153 and 1699-1854 ------------------------------------------------------
    def cmd_info(self) -> int:
        o = self.o
        if o.output_format == "tsv":
            self.p("File\tMP3 gain\tdB gain\tMax Amplitude\tMax global_gain\tMin global_gain")
        results = []
        for file in o.files:
            r = self.process_info(file)
            if o.output_format == "json":
                results.append(r)
        if o.output_format == "json":
            _print_json(self.out, files=results)
        return 0

    def process_info(self, file: Path) -> dict:
        o = self.o
        name = _name(file)
        if o.output_format == "tsv":  # mp3gain-compatible TSV (what beets parses): ReplayGain analysis
            try:
                rg = self.analyze_track(file)
            except rgmod.ReplayGainError as ex:
                self.e(f"{name} - {ex}")
                return _file_result(file, status="error", error=str(ex))
            try:
                max_amp, max_gain, min_gain = self.find_max_amplitude(file)
            except (mp3gain.Mp3GainError, rgmod.ReplayGainError):
                max_amp, max_gain, min_gain = 1.0, 255, 0
            gain_db = rg.gain_db + o.gain_modifier_db  # -d shifts the suggested gain
            steps = mp3gain.db_to_steps(gain_db)
            self.p(f"{name}\t{steps}\t{gain_db:.6f}\t{rg.peak * 32768.0:.6f}\t{max_gain}\t{min_gain}")
            return _file_result(file, loudness_db=rg.loudness_db, gain_applied_db=gain_db, gain_applied_steps=steps, peak=rg.peak,
                                max_amplitude=max_amp, max_gain=max_gain, min_gain=min_gain)
        if mp4meta.is_mp4_file(file):
            if self.text:
                if o.quiet:
                    self.p(f"{name}\tM4A/AAC\t-\t-\t-\t-\t-")
                else:
                    self.p(name)
                    self.p("  Format:      M4A/AAC")
                    self.p("  Note: Use -r or -a for ReplayGain analysis")
                    self.p()
            return _file_result(file, status="info")
        try:
            info = mp3gain.analyze(file)
        except mp3gain.Mp3GainError as ex:
            if o.output_format != "json":
                self.e(f"{name} - {ex}")
            return _file_result(file, status="error", error=str(ex))
        if self.text:
            if o.quiet:
                self.p(f"{name}\t{info.frame_count}\t{info.min_gain}\t{info.max_gain}\t{info.avg_gain:.1f}\t{info.headroom_steps}\t{info.headroom_db:.1f}")
            else:
                self.p(name)
                self.p(f"  Format:      {info.mpeg_version} Layer III, {info.channel_mode}")
                self.p(f"  Frames:      {info.frame_count}")
                self.p(f"  Gain range:  {info.min_gain} - {info.max_gain} (avg: {info.avg_gain:.1f})")
                self.p(f"  Headroom:    {info.headroom_steps} steps ({info.headroom_db:+.1f} dB)")
                self.p()
        return _file_result(file, mpeg_version=info.mpeg_version, channel_mode=info.channel_mode, frames=info.frame_count,
                            min_gain=info.min_gain, max_gain=info.max_gain, avg_gain=info.avg_gain,
                            headroom_steps=info.headroom_steps, headroom_db=info.headroom_db)

    # ---- -u, src/main.rs:1155-1211 and 1856-1935 ---------------------------------------------------------------------
    def cmd_undo(self) -> int:
        o = self.o
        pre = "[DRY RUN] " if o.dry_run else ""
        if self.talk:
            self.p(f"{pre}mp3rgain {'Would undo' if o.dry_run else 'Undoing'} gain changes on {len(o.files)} file(s)")
            self.p()
        results, tally = [], [0, 0]
        for file in o.files:
            name = _name(file)
            if o.dry_run:
                if self.talk:
                    self.p(f"  ~ [DRY RUN] {name} (would undo)")
                r = _file_result(file, status="dry_run", dry_run=True)
            else:
                mtime = _mtime(file) if o.preserve_timestamp else None
                try:
                    frames = mp3gain.undo_gain(file)
                    if frames == 0:
                        if self.talk:
                            self.p(f"  . {name} (no changes to undo)")
                        r = _file_result(file, status="skipped", frames=0)
                    else:
                        _restore(file, mtime)
                        if self.talk:
                            self.p(f"  v {name} ({frames} frames restored)")
                        r = _file_result(file, status="success", frames=frames)
                except mp3gain.Mp3GainError as ex:
                    if self.talk:
                        self.e(f"  x {name} - {ex}")
                    r = _file_result(file, status="error", error=str(ex))
            _count(r, tally)
            if o.output_format == "json":
                results.append(r)
        if o.output_format == "json":
            _print_json(self.out, files=results, summary=_summary(len(o.files), tally[0], tally[1], o.dry_run))
        else:
            self._dry_run_notice()
        return 0

    # ---- -r / -e, src/main.rs:1213-1282 and 1937-2001 ------------------------------------------------------------------
    def cmd_track_gain(self) -> int:
        o = self.o
        pre = "[DRY RUN] " if o.dry_run else ""
        if self.talk:
            self.p(f"{pre}mp3rgain Analyzing and {'would apply' if o.dry_run else 'applying'} track gain to {len(o.files)} file(s)")
            self.p(f"  Target: {_rust_float(REFERENCE_DB)} dB (ReplayGain 1.0)")
            if o.gain_modifier != 0:
                self.p(f"  Gain modifier: {o.gain_modifier:+d} steps")
            self.p()
        results, tally = [], [0, 0]
        for file in o.files:
            r = self.process_track_gain(file)
            _count(r, tally)
            if o.output_format == "json":
                results.append(r)
        if o.output_format == "json":
            _print_json(self.out, files=results, summary=_summary(len(o.files), tally[0], tally[1], o.dry_run))
        else:
            self._dry_run_notice()
        return 0

    def process_track_gain(self, file: Path) -> dict:
        o = self.o
        name = _name(file)
        pre = "[DRY RUN] " if o.dry_run else ""
        if self.talk:
            self.p(f"  -> {pre}Analyzing {name}...")
        try:
            rg = self.analyze_track(file)
        except rgmod.ReplayGainError as ex:
            if self.talk:
                self.e(f"  x {name} - {ex}")
            return _file_result(file, status="error", error=str(ex))
        base = rg.gain_steps()
        modified = base + o.gain_modifier
        if self.talk:
            extra = f" + {o.gain_modifier} = {modified}" if o.gain_modifier != 0 else ""
            self.p(f"      Loudness: {rg.loudness_db:.1f} dB, Gain: {rg.gain_db:+.1f} dB ({base} steps{extra}), Peak: {rg.peak:.4f}")
        if modified == 0:
            if self.talk:
                self.p(f"  . {name} (no adjustment needed)")
            return _file_result(file, status="skipped", loudness_db=rg.loudness_db, peak=rg.peak, gain_applied_steps=0,
                                gain_applied_db=0.0)
        return self.apply_replaygain(file, modified, rg, None)

    # ---- process_apply_replaygain_with_album, src/main.rs:2012-2170; AAC branch :2173-2241 -----------------------------
    def apply_replaygain(self, file: Path, steps: int, rg: rgmod.ReplayGainResult, album: Optional[Tuple[float, float]]) -> dict:
        o = self.o
        name = _name(file)
        pre = "[DRY RUN] " if o.dry_run else ""
        mtime = _mtime(file) if o.preserve_timestamp and not o.dry_run else None
        actual, warning = steps, None
        if steps > 0 and not o.wrap_gain:
            new_peak = rg.peak * 10.0 ** (rg.gain_db / 20.0)
            if new_peak > 1.0:
                if o.prevent_clipping:
                    actual = max(mp3gain.db_to_steps(-20.0 * math.log10(rg.peak)), 0)
                    if self.talk:
                        self.e(f"  ! {pre}{name} - gain reduced from {steps} to {actual} steps to prevent clipping (peak: {rg.peak:.4f})")
                    warning = f"gain reduced from {steps} to {actual} steps to prevent clipping (peak: {rg.peak:.4f})"
                elif not o.ignore_clipping and not o.quiet:
                    if self.text:
                        self.e(f"  ! {pre}{name} - clipping warning: peak would be {new_peak:.2f} (>{1.0:.2f})")
                        self.e("      Use -c to ignore clipping warnings or -k to prevent clipping")
                    warning = f"clipping warning: peak would be {new_peak:.2f} (>1.00)"
        is_aac = rg.file_type == rgmod.AudioFileType.Aac
        if o.dry_run:
            if self.talk:
                self.p(f"  ~ [DRY RUN] {name} (would apply {actual * GAIN_STEP_DB:+.1f} dB, {actual} steps{' (tags only)' if is_aac else ''})")
            return _file_result(file, status="dry_run", loudness_db=rg.loudness_db, peak=rg.peak, gain_applied_steps=actual,
                                gain_applied_db=actual * GAIN_STEP_DB, warning=warning, dry_run=True)
        if is_aac:  # AAC samples cannot be changed losslessly: ReplayGain tags only
            tags = mp4meta.ReplayGainTags()
            tags.set_track(rg.gain_db, rg.peak)
            if album is not None:
                tags.set_album(*album)
            try:
                mp4meta.write_replaygain_tags(file, tags)
            except mp4meta.Mp4MetaError as ex:
                if self.talk:
                    self.e(f"  x {name} - {ex}")
                return _file_result(file, status="error", error=str(ex))
            _restore(file, mtime)
            if self.talk:
                self.p(f"  v {name} ({'track+album tags' if album is not None else 'tags'} written, {rg.gain_db:+.1f} dB)")
            return _file_result(file, status="success", loudness_db=rg.loudness_db, peak=rg.peak, gain_applied_steps=rg.gain_steps(),
                                gain_applied_db=rg.gain_db, warning=warning)
        if _is_riff_wave(file):
            # The reference cannot get here (its probe knows no WAV); this library analyses WAV, but PCM has no global_gain
            # fields: the frame scanner would take false syncs in the samples for frames and rewrite audio bytes.
            msg = "RIFF/WAVE input is analysed only: lossless gain applies to MPEG Layer III frames"
            if self.talk:
                self.e(f"  x {name} - {msg}")
            return _file_result(file, status="error", error=msg)
        fn = mp3gain.apply_gain_with_undo_wrap if o.wrap_gain else mp3gain.apply_gain_with_undo
        try:
            frames = self._with_temp_file(file, lambda f: fn(f, actual))
        except (mp3gain.Mp3GainError, OSError) as ex:
            if self.talk:
                self.e(f"  x {name} - {ex}")
            return _file_result(file, status="error", error=str(ex))
        _restore(file, mtime)
        if self.talk:
            self.p(f"  v {name} ({frames} frames, {actual * GAIN_STEP_DB:+.1f} dB)")
        return _file_result(file, status="success", frames=frames, loudness_db=rg.loudness_db, peak=rg.peak, gain_applied_steps=actual,
                            gain_applied_db=actual * GAIN_STEP_DB, warning=warning)

    # ---- -a, src/main.rs:1284-1452 -------------------------------------------------------------------------------------
    def cmd_album_gain(self) -> int:
        o = self.o
        pre = "[DRY RUN] " if o.dry_run else ""
        if self.talk:
            self.p(f"{pre}mp3rgain Analyzing album gain for {len(o.files)} file(s)")
            self.p(f"  Target: {_rust_float(REFERENCE_DB)} dB (ReplayGain 1.0)")
            if o.gain_modifier != 0:
                self.p(f"  Gain modifier: {o.gain_modifier:+d} steps")
            self.p()
            self.p("  -> Analyzing tracks...")
        try:
            album = self.analyzer().analyze_album_files(o.files, o.track_index)
        except rgmod.ReplayGainError as ex:
            if o.output_format == "json":
                _print_json(self.out, summary=_summary(len(o.files), 0, len(o.files), o.dry_run))
            else:
                self.e(f"error: Failed to analyze album: {ex}")
            return 1
        base = album.album_gain_steps()
        steps = base + o.gain_modifier
        if self.talk:
            self.p()
            self.p(f"  Album loudness: {album.album_loudness_db:.1f} dB")
            extra = f" + {o.gain_modifier} = {steps}" if o.gain_modifier != 0 else ""
            self.p(f"  Album gain:     {album.album_gain_db:+.1f} dB ({base} steps{extra})")
            self.p(f"  Album peak:     {album.album_peak:.4f}")
            self.p()
        album_json = {"loudness_db": album.album_loudness_db, "gain_db": album.album_gain_db, "gain_steps": steps, "peak": album.album_peak}
        if steps == 0:
            if o.output_format == "json":
                files = [_file_result(f, status="skipped", loudness_db=t.loudness_db, peak=t.peak, gain_applied_steps=0, gain_applied_db=0.0)
                         for f, t in zip(o.files, album.tracks)]
                _print_json(self.out, files=files, album=album_json, summary=_summary(len(o.files), 0, 0, o.dry_run))
            elif not o.quiet:
                self.p("  . No adjustment needed")
            return 0
        results, tally = [], [0, 0]
        for file, track in zip(o.files, album.tracks):
            r = self.apply_replaygain(file, steps, track, (album.album_gain_db, album.album_peak))
            _count(r, tally)
            if o.output_format == "json":
                results.append(r)
        if o.output_format == "json":
            _print_json(self.out, files=results, album=album_json, summary=_summary(len(o.files), tally[0], tally[1], o.dry_run))
        else:
            self._dry_run_notice()
        return 0


def _signed(x: float, prec: int) -> str:  # `{:+.N}`
    return "+inf" if math.isinf(x) and x > 0 else f"{x:+.{prec}f}"


def _plain(x: float, prec: int) -> str:  # `{:.N}`
    return "inf" if math.isinf(x) and x > 0 else f"{x:.{prec}f}"


def _mtime(file: Path):
    try:
        st = os.stat(file)
        return (st.st_atime_ns, st.st_mtime_ns)
    except OSError:
        return None


def _restore(file: Path, t):  # restore_timestamp, src/main.rs:2243-2247
    if t is not None:
        try:
            os.utime(file, ns=(os.stat(file).st_atime_ns, t[1]))
        except OSError:
            pass


def expand_files_recursive(paths: List[Path]) -> List[Path]:  # src/main.rs:436-470
    out: List[Path] = []

    def walk(d: Path):
        for entry in os.scandir(d):
            p = Path(entry.path)
            if entry.is_dir():
                walk(p)
            elif p.suffix.lower() in (".mp3", ".m4a", ".aac", ".mp4"):
                out.append(p)

    for p in paths:
        if p.is_dir():
            walk(p)
        else:
            out.append(p)
    return sorted(out)


def print_version(out):  # src/main.rs:2254-2259
    print(f"mp3rgain_amd version {VERSION} (command-line compatible with mp3rgain 1.5.0)", file=out)
    print("MI355X-native ReplayGain analysis behind mp3rgain's interface", file=out)
    print(file=out)
    print(f"Each gain step = {_rust_float(GAIN_STEP_DB)} dB", file=out)


def print_usage(out):  # src/main.rs:2261-2346, shortened to the option table
    print(f"mp3rgain_amd version {VERSION}", file=out)
    print("Lossless MP3 volume adjustment and GPU ReplayGain analysis - mp3rgain's command line", file=out)
    print(file=out)
    print("USAGE:", file=out)
    print("    python -m mp3rgain_amd [OPTIONS] <FILES>...", file=out)
    print(file=out)
    print("OPTIONS:", file=out)
    for line in (
        f"-g <i>      Apply gain of i steps (each step = {_rust_float(GAIN_STEP_DB)} dB)",
        "-d <n>      Modify the suggested dB gain by n (TSV info output)",
        "-l <c> <g>  Apply gain to left (0) or right (1) channel only",
        "-m <i>      Modify suggested gain by integer i",
        "-r          Apply Track gain (ReplayGain analysis)",
        "-a          Apply Album gain (ReplayGain analysis)",
        "-e          Skip album analysis (even with multiple files)",
        "-i <n>      Specify which audio track to process (default: 0)",
        "-u          Undo gain changes (restore from APEv2 tag)",
        "-x          Only find max amplitude of file",
        "-s <mode>   Stored tag handling: c check, d delete, s skip, r recalc, i ID3v2, a APEv2",
        "-p          Preserve original file timestamp",
        "-c          Ignore clipping warnings",
        "-k          Prevent clipping (automatically limit gain)",
        "-w          Wrap gain values (instead of clamping)",
        "-t          Use temp file for writing",
        "-f          Assume MPEG 2 Layer III (compatibility, no effect)",
        "-q          Quiet mode (less output)",
        "-R          Process directories recursively",
        "-n          Dry-run mode (show what would be done)",
        "--dry-run   Same as -n",
        "-o <fmt>    Output format: 'text' (default), 'json', or 'tsv'",
        "--decoder <cmd>  Decoder command for files that are not WAV: writes a WAV stream to stdout, {} = file",
        "-v          Show version",
        "-h          Show this help",
    ):
        print("    " + line, file=out)
    print(file=out)
    print("NOTES:", file=out)
    print(f"    - Each gain step = {_rust_float(GAIN_STEP_DB)} dB (fixed by MP3 specification)", file=out)
    print("    - Changes are lossless and reversible; gain changes are stored in APEv2 tags for undo support", file=out)
    print(f"    - ReplayGain analysis runs on the GPU (target: {_rust_float(REFERENCE_DB)} dB); there is no CPU path", file=out)


def main(argv: Optional[List[str]] = None, out=None, err=None) -> int:
    """main, src/main.rs:171-181."""
    out = out or sys.stdout
    err = err or sys.stderr
    args = list(sys.argv[1:] if argv is None else argv)
    if not args:
        print_usage(out)
        return 0
    try:
        opts = parse_args(args, out, err)
        return Cli(opts, out, err).run()
    except Exit as ex:
        return ex.code
    except CliError as ex:
        print(f"Error: {ex}", file=err)
        return 1
