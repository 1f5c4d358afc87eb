#!/usr/bin/env python3
"""Dense, music-like MP3 streams and their golden PCM (tests/golden/mp3/dense_*.mp3 + .ffmpeg.npz).

    tools/make_mp3_dense.py            streams (oracle/mp3_encoder.py, seeded) + ffmpeg PCM via kaleido
    tools/make_mp3_dense.py --no-pcm   streams only

The other golden streams are a few frames of random quantised spectra each (tools/make_mp3_golden.py), and the
reference's fixtures are one-second sines.  These are tens of seconds of a synthetic piece of music -- bass, chords,
a melody with vibrato, kick / snare / hi-hat with real attacks, a crescendo, instruments panned across the stereo
image -- put through a real encoder chain (analysis filterbank, MDCT with window switching, mid/side decisions per
frame, scalefactors, bit reservoir, cheapest Huffman tables), so the decoders see what an encoder produces from a dense
signal.  The golden PCM is ffmpeg's decode (tools/ffmpeg_golden.py); it is stored compressed (int16, npz).
Run in the build container only (kaleido does not travel); the outputs are committed.
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))
sys.path.insert(0, str(ROOT / "tools"))

OUT = ROOT / "tests" / "golden" / "mp3"


def piece(rate: int, seconds: float, channels: int, seed: int) -> np.ndarray:
    """A seeded piece of synthetic music, float64 [channels][frames], peak 0.85."""
    rng = np.random.default_rng(seed)
    n = int(rate * seconds)
    t = np.arange(n) / rate
    nyq = rate / 2
    left = np.zeros(n)
    right = np.zeros(n)
    bpm = 112.0
    beat = 60.0 / bpm
    chords = [(57, 60, 64, 69), (53, 57, 60, 65), (48, 55, 60, 64), (55, 59, 62, 67)]  # Am F C G (MIDI notes)

    def hz(m):
        return 440.0 * 2.0 ** ((m - 69) / 12.0)

    def add(sig, start, pan, gain):
        a = int(start * rate)
        if a >= n:
            return
        m = min(len(sig), n - a)
        left[a:a + m] += sig[:m] * gain * np.cos(pan * np.pi / 2)
        right[a:a + m] += sig[:m] * gain * np.sin(pan * np.pi / 2)

    def tone(f, dur, harmonics, decay, vibrato=0.0):
        k = int(dur * rate)
        tt = np.arange(k) / rate
        ph = 2 * np.pi * f * tt + (vibrato * np.sin(2 * np.pi * 5.5 * tt) if vibrato else 0.0)
        s = np.zeros(k)
        for h, a in enumerate(harmonics, start=1):
            if f * h < 0.45 * rate:
                s += a * np.sin(h * ph + 0.3 * h)
        env = np.minimum(1.0, tt / 0.008) * np.exp(-tt * decay) * np.minimum(1.0, (dur - tt) / 0.01)
        return s * env

    nbars = int(seconds / (4 * beat)) + 1
    for bar in range(nbars):
        ch = chords[bar % 4]
        t0 = bar * 4 * beat
        level = min(1.0, 0.25 + 0.75 * t0 / (0.6 * seconds))  # crescendo over the first 60 %
        # pad: the chord, slightly detuned pairs left and right
        for note in ch:
            add(tone(hz(note) * 1.003, 4 * beat, [1, 0.5, 0.33, 0.2, 0.1, 0.07], 0.4), t0, 0.25, 0.11 * level)
            add(tone(hz(note) * 0.997, 4 * beat, [1, 0.4, 0.3, 0.25, 0.12, 0.05], 0.4), t0, 0.75, 0.11 * level)
        # bass on every beat, centre
        for b in range(4):
            add(tone(hz(ch[0] - 24), beat * 0.9, [1, 0.7, 0.45, 0.3, 0.2, 0.12, 0.08], 3.0), t0 + b * beat, 0.5, 0.32 * level)
        # melody: eighth notes from the chord's scale, vibrato, wandering pan
        for e in range(8):
            if rng.random() < 0.8:
                note = ch[rng.integers(0, 4)] + 12 + int(rng.choice([0, 0, 2, -2, 7]))
                add(tone(hz(note), beat * 0.5 * rng.uniform(0.6, 1.6), [1, 0.2, 0.35, 0.05, 0.1], 4.0, vibrato=0.6),
                    t0 + e * beat / 2, float(rng.uniform(0.2, 0.8)), 0.2 * level)
        # drums (from the second bar on): kick on 1 and 3, snare on 2 and 4, hi-hats on the off-beats
        if bar >= 1:
            for b in range(4):
                if b % 2 == 0:
                    k = int(0.25 * rate)
                    tt = np.arange(k) / rate
                    kick = np.sin(2 * np.pi * (45 * tt + 60 * (1 - np.exp(-tt * 30)) / 30)) * np.exp(-tt * 14)
                    add(kick, t0 + b * beat, 0.5, 0.55 * level)
                else:
                    k = int(0.18 * rate)
                    tt = np.arange(k) / rate
                    sn = rng.standard_normal(k) * np.exp(-tt * 28) + 0.5 * np.sin(2 * np.pi * 190 * tt) * np.exp(-tt * 20)
                    add(sn, t0 + b * beat, 0.45, 0.3 * level)
                k = int(0.05 * rate)
                tt = np.arange(k) / rate
                hh = np.diff(rng.standard_normal(k + 1)) * np.exp(-tt * 90)  # differenced noise: high-passed
                add(hh, t0 + (b + 0.5) * beat, 0.7, 0.16 * level)
    # a breath of room noise under everything, one silent half second near the start
    left += 0.0008 * rng.standard_normal(n)
    right += 0.0008 * rng.standard_normal(n)
    a = int(1.5 * rate)
    left[a:a + rate // 2] = 0.0
    right[a:a + rate // 2] = 0.0
    out = np.stack([left, right]) if channels == 2 else (0.5 * (left + right))[None, :]
    return out * (0.85 / np.abs(out).max())


# name, rate, channels, seconds, bitrate, seed, allow_ms
CASES = [
    ("dense_44k_joint_128", 44100, 2, 24.0, 128, 11, True),
    ("dense_48k_stereo_192", 48000, 2, 8.0, 192, 12, False),
    ("dense_22k_mono_56", 22050, 1, 10.0, 56, 13, False),
    ("dense_32k_joint_96", 32000, 2, 6.0, 96, 14, True),     # MPEG-1 at its lowest rate, mid/side
    ("dense_24k_joint_64", 24000, 2, 6.0, 64, 15, True),     # MPEG-2 (one granule per frame), mid/side, short blocks in the LSF syntax
    ("dense_11k_stereo_32", 11025, 2, 6.0, 32, 16, False),   # MPEG-2.5
]


def main():
    import mp3_encoder as E
    from mp3rgain_amd import mp3dec

    want_pcm = "--no-pcm" not in sys.argv
    only = [a for a in sys.argv[1:] if not a.startswith("--")]
    for name, rate, nch, secs, br, seed, ms in CASES:
        if only and name not in only:
            continue
        pcm = piece(rate, secs, nch, seed)
        data = E.encode(pcm, rate, br, seed=seed, allow_ms=ms)
        (OUT / f"{name}.mp3").write_bytes(data)
        dec, info = mp3dec.decode(data)
        d = 1057  # analysis + synthesis filterbank (481) and one granule of MDCT overlap (576)
        m = min(dec.shape[1] - d, pcm.shape[1])
        err = dec[:, d:d + m] - pcm[:, :m]
        snr = 10 * np.log10((pcm[:, :m] ** 2).sum() / (err ** 2).sum())
        print(f"{name}: {len(data)} bytes, {info.audio_frames} frames, round-trip SNR {snr:.1f} dB, peak {np.abs(dec).max():.3f}")
        if want_pcm:
            import ffmpeg_golden as G

            g, ginfo = G.decode(data, rate)
            p64 = g.astype(np.float64)
            q = np.round(np.where(p64 > 0, p64 * 32767.0, p64 * 32768.0))
            assert np.abs(np.where(q > 0, q / 32767.0, q / 32768.0) - p64).max() < 1e-6 and np.abs(g).max() < 0.999
            np.savez_compressed(OUT / f"{name}.ffmpeg.npz", pcm=np.clip(q, -32768, 32767).astype(np.int16))
            k = min(q.shape[1], dec.shape[1])
            dd = dec[:, :k].astype(np.float64) * 32768.0 - q[:, :k]
            print(f"   ffmpeg: {ginfo['length']} frames; this decoder - ffmpeg: max {np.abs(dd).max():.2f} rms {np.sqrt((dd ** 2).mean()):.3f} steps")


if __name__ == "__main__":
    main()
