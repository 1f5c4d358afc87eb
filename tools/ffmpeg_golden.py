#!/usr/bin/env python3
"""Golden PCM for the MP3 decoder tests, from an independent conformant decoder that exists in THIS image only.

The reference decodes with symphonia, which cannot run here (no Rust toolchain, crate not vendored), and the image has
no mpg123 / ffmpeg / lame.  It does have a headless Chromium inside the `kaleido` wheel (plotly's static-image
exporter) whose Web Audio `decodeAudioData` runs ffmpeg's MPEG audio decoder.  This script drives it: kaleido's plotly
scope loads a user-supplied "plotly.js"; ours is a 30-line stand-in whose `Plotly.toImage` decodes the MP3 bytes it is
handed and returns the samples.  Output: int16 PCM (that build decodes MP3 with ffmpeg's fixed-point decoder: every
sample is a multiple of 2^-15, accurate to about one such step), gapless-trimmed by the Xing/LAME header when there
is one (the fixtures: 1105 = 576 + 528 + 1 frames of encoder + decoder delay cut from the front, 44100 kept).

    tools/ffmpeg_golden.py out_dir file.mp3 [file.mp3 ...]      -> out_dir/<stem>.ffmpeg.npy  (int16 [channels][frames])

Only the outputs are committed (tests/golden/mp3/); kaleido does not travel and the tests do not need it.
"""
import base64
import json
import sys
import tempfile
from pathlib import Path

import numpy as np

FAKE_PLOTLY = r"""
window.Plotly = {
  version: '2.18.0',
  toImage: function (fig, opts) {
    return new Promise(function (resolve) {
      try {
        var req = fig.data[0];
        var bin = atob(req.mp3);
        var buf = new Uint8Array(bin.length);
        for (var i = 0; i < bin.length; i++) buf[i] = bin.charCodeAt(i);
        var ctx = new OfflineAudioContext(2, 128, req.rate);
        ctx.decodeAudioData(buf.buffer, function (ab) {
          var out = {rate: ab.sampleRate, channels: ab.numberOfChannels, length: ab.length, ua: navigator.userAgent, pcm: []};
          for (var c = 0; c < ab.numberOfChannels; c++) {
            var f = ab.getChannelData(c);
            var u8 = new Uint8Array(f.buffer, f.byteOffset, f.byteLength);
            var s = '';
            for (var k = 0; k < u8.length; k += 8192) s += String.fromCharCode.apply(null, u8.subarray(k, k + 8192));
            out.pcm.push(btoa(s));
          }
          resolve(JSON.stringify(out));
        }, function (e) { resolve(JSON.stringify({error: 'decode: ' + e})); });
      } catch (e) { resolve(JSON.stringify({error: 'exc: ' + e})); }
    });
  }
};
"""

_scope = None


def _get_scope():
    global _scope
    if _scope is None:
        from kaleido.scopes.plotly import PlotlyScope

        js = Path(tempfile.mkdtemp()) / "fake_plotly.js"
        js.write_text(FAKE_PLOTLY)
        _scope = PlotlyScope(plotlyjs=str(js))
    return _scope


def decode(mp3: bytes, rate: int):
    """-> (float32 [channels][frames], info dict).  `rate` must be the stream's rate (no resampling then)."""
    fig = {"data": [{"mp3": base64.b64encode(mp3).decode(), "rate": rate}], "layout": {}}
    resp = _get_scope()._perform_transform(fig, format="json", width=100, height=100, scale=1)
    if resp.get("code") != 0:
        raise RuntimeError(f"kaleido: {resp}")
    r = json.loads(resp["result"])
    if "error" in r:
        raise RuntimeError(r["error"])
    chans = [np.frombuffer(base64.b64decode(p), dtype="<f4") for p in r["pcm"]]
    return np.stack(chans), {k: r[k] for k in ("rate", "channels", "length", "ua")}


def mp3_rate(data: bytes) -> int:
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    from mp3rgain_amd import mp3dec

    return int(mp3dec.scan(data).sample_rate)


def main():
    out = Path(sys.argv[1])
    out.mkdir(parents=True, exist_ok=True)
    for f in sys.argv[2:]:
        data = Path(f).read_bytes()
        rate = mp3_rate(data)
        pcm, info = decode(data, rate)
        assert info["rate"] == rate
        # that build's MP3 decoder is ffmpeg's fixed-point one: every sample is a multiple of 2^-15
        p64 = pcm.astype(np.float64)
        q = np.round(np.where(p64 > 0, p64 * 32767.0, p64 * 32768.0))  # Chromium's int16 -> float: /32767 above zero, /32768 below
        assert np.abs(np.where(q > 0, q / 32767.0, q / 32768.0) - p64).max() < 1e-6, "not 16-bit quantised?"
        np.save(out / (Path(f).stem + ".ffmpeg.npy"), np.clip(q, -32768, 32767).astype(np.int16))
        print(f, info, "peak", float(np.abs(pcm).max()))


if __name__ == "__main__":
    main()
