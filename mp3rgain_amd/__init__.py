"""mp3rgain_amd -- MI355X (gfx950) ReplayGain 1.0 analysis path behind mp3rgain's `replaygain` API.

Only what the hot path needs lives here: `csrc/` (HIP kernels + the C ABI declared in
include/mp3rgain_amd.h), the ctypes binding (`_capi`) and the host-side mirror of the
reference's interface (`replaygain`).
"""
from . import _capi, replaygain  # noqa: F401
from .replaygain import (  # noqa: F401
    REPLAYGAIN_REFERENCE_DB,
    AlbumGainResult,
    Analyzer,
    AudioFileType,
    PcmTrack,
    PeakAmplitudeResult,
    ReplayGainError,
    ReplayGainResult,
    analyze_album,
    analyze_track,
    find_peak_amplitude,
    is_available,
)

__version__ = "0.1.0"
