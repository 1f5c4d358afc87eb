"""`python -m mp3rgain_amd [OPTIONS] <FILES>...` -- mp3rgain's command line (see cli.py)."""
import os
import sys

# the command line never needs PyTorch: the library runs on the system's HIP runtime and the process starts 1.5 s sooner
os.environ.setdefault("MP3RGAIN_AMD_STANDALONE", "1")

from .cli import main  # noqa: E402

sys.exit(main())
