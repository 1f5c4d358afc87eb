"""SURVEY.md §8f rows 2 and 3 on the GPU box.  The lossless MP3 gain scanner/patcher with its APEv2 undo tags
(tests/test_mp3gain.py) and the MP4 freeform ReplayGain tags (tests/test_mp4meta.py) are host byte work with no kernel:
their tests belong to the CPU suite (`-m "not gpu"`) and are NOT re-collected under the gpu mark (until round 3 they were,
which put 175 kernel-less tests into the GPU count).  What is left here is the one thing only the GPU box can show: that
the library build the kernels live in -- the libmp3rgain_amd.so loaded by the analysis tests of this same session -- is
the one that serves those entry points, end to end on one file of each kind."""
import shutil
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
FIX = ROOT / "tests" / "golden" / "fixtures"

pytestmark = pytest.mark.gpu


def test_gain_patch_and_undo_on_the_gpu_box_build(tmp_path):
    """apply_gain / undo_gain (lib.rs:262-616, 838-1163) through the same shared object as rg_create"""
    import mp3rgain_amd as rg
    from mp3rgain_amd import _capi, mp3gain

    with rg.Analyzer(0):  # the kernels' library is loaded and has a device
        pass
    assert Path(_capi.LIB_PATH).resolve() == (ROOT / "mp3rgain_amd" / "libmp3rgain_amd.so").resolve() or "MP3RGAIN_AMD_LIB" in __import__("os").environ
    src = FIX / "test_stereo.mp3"
    dst = tmp_path / "t.mp3"
    shutil.copy(src, dst)
    before = mp3gain.analyze(dst)
    n = mp3gain.apply_gain_with_undo(dst, -2)  # (the fixture sits at global_gain 255: upwards would saturate)
    assert n == before.frame_count
    after = mp3gain.analyze(dst)
    assert after.min_gain == before.min_gain - 2 and after.max_gain == before.max_gain - 2
    mp3gain.undo_gain(dst)
    again = mp3gain.analyze(dst)
    assert (again.min_gain, again.max_gain) == (before.min_gain, before.max_gain)


def test_mp4_tags_round_trip_on_the_gpu_box_build(tmp_path):
    """mp4meta.rs: freeform ReplayGain tags written and read back through the same shared object"""
    sys.path.insert(0, str(ROOT / "tests"))
    import test_mp4meta as T

    T.test_file_level_round_trip_and_errors(tmp_path)
