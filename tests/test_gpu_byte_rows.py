"""SURVEY.md §8f rows 2 and 3 on the GPU box: the lossless MP3 gain scanner/patcher with its APEv2 undo tags
(tests/test_mp3gain.py) and the MP4 freeform ReplayGain tags (tests/test_mp4meta.py) are host byte work with no
kernel, so their tests are not GPU tests -- but the driver's round-end record only covers `-m gpu`.  This module
re-collects every test of those two files under the gpu mark, so that the record shows them running against the
library build that is on the GPU box (the same libmp3rgain_amd.so the kernels live in)."""
import sys
from pathlib import Path

import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent))

from test_mp3gain import *  # noqa: F401,F403,E402
from test_mp4meta import *  # noqa: F401,F403,E402

pytestmark = pytest.mark.gpu
