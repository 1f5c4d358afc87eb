"""`python -m mp3rgain_amd [OPTIONS] <FILES>...` -- mp3rgain's command line (see cli.py)."""
import sys

from .cli import main

sys.exit(main())
