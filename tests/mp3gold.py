"""Golden PCM of the MP3 streams under tests/golden/mp3/: ffmpeg's decode (tools/ffmpeg_golden.py), int16
[channels][frames]; short streams are .npy, the long dense ones (tools/make_mp3_dense.py) compressed .npz."""
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
GOLD = ROOT / "tests" / "golden" / "mp3"
FIX = ROOT / "tests" / "golden" / "fixtures"
LAME_DELAY = 576 + 528 + 1  # encoder delay + decoder delay, what a gapless-aware decoder cuts from the front
# streams with an ffmpeg golden (the reference's test_stereo.mp3 is damaged: its sane copy is test_stereo_minus125.mp3)
STREAMS = sorted(GOLD.glob("*.mp3")) + [FIX / n for n in ("test_joint_stereo.mp3", "test_mono.mp3", "test_vbr.mp3")]
# max |delta| and RMS in steps of 2^-15 against ffmpeg's fixed-point decoder (itself good to about one step)
MAX_STEPS, RMS_STEPS = 1.5, 0.6


def load_gold(path: Path) -> np.ndarray:
    npy = GOLD / (path.stem + ".ffmpeg.npy")
    if npy.exists():
        return np.load(npy)
    with np.load(GOLD / (path.stem + ".ffmpeg.npz")) as z:
        return z["pcm"]


def compare_with_gold(pcm: np.ndarray, info_frame: int, gold: np.ndarray):
    """-> (max |delta|, rms, offset, n) in steps of 2^-15 over the golden decode's span.  The golden decoder trims by the
    Xing/LAME header when there is one; this decoder, like the reference (FormatOptions::default()), never trims."""
    off = LAME_DELAY if info_frame else 0
    n = min(gold.shape[1], pcm.shape[1] - off)
    assert n >= gold.shape[1] - 1152 and n > 4000
    d = pcm[:, off:off + n].astype(np.float64) * 32768.0 - gold[:, :n].astype(np.float64)
    return float(np.abs(d).max()), float(np.sqrt((d ** 2).mean())), off, n
