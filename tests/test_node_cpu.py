"""include/mp3rgain_amd_node.h without a GPU: the node's own logic -- dealing files out by size, running every device's
share on its own host thread, the first failing file IN INPUT ORDER ending the album whichever device had it
(src/replaygain.rs:1055), nobody being asked for its pack after a failure, the host fold of the [histogram | peak]
packs (LoudnessHistogram::accumulate :658-662, album_peak.max :1056), results back in input order (:1061) -- driven
through rg_node_create_backend with a table of Python functions standing in for the per-device engines.  What an engine
computes here is a stand-in (a "file" is a few numbers); the real engine, rg_ctx, is exercised by tests/test_gpu_node.py."""
import ctypes as C
import json
import threading

import numpy as np
import pytest

from mp3rgain_amd import _capi, album
from mp3rgain_amd import replaygain as R

H = _capi.HISTOGRAM_SIZE


class FakeEngines:
    """Per-device engine: a file is JSON {"bins": {bin: count}, "peak": p} or {"fail": "text", "code": rc}."""

    def __init__(self):
        self.lib = _capi.load()
        self.lock = threading.Lock()
        self.engines = {}        # handle -> state
        self.calls = []          # (what, device, n)
        self.threads = set()
        self._keep = []
        self.table = _capi.NodeBackend(
            _capi.NODE_OPEN(self.open), _capi.NODE_CLOSE(self.close), _capi.NODE_ALBUM_BEGIN(self.album_begin),
            _capi.NODE_ALBUM_PACK(self.album_pack), _capi.NODE_TRACKS(self.tracks), _capi.NODE_TRACKS_ERROR(self.tracks_error),
            _capi.NODE_LAST_ERROR(self.last_error), None)

    def open(self, device, user):
        if device == 666:
            return None
        with self.lock:
            h = 1000 + len(self.engines)
            self.engines[h] = {"device": device, "err": b"", "pack": None, "terr": []}
        return h

    def close(self, h, user):
        with self.lock:
            self.engines[h]["closed"] = True

    def _one(self, path):
        d = json.loads(open(path).read().split("\n")[0])
        if "fail" in d:
            return None, d
        hist = np.zeros(H, np.uint32)
        for b, c in d["bins"].items():
            hist[int(b)] = c
        loud = self.lib.rg_hist_loudness(hist.ctypes.data)
        return (hist, loud, float(d["peak"])), d

    def album_begin(self, h, paths, n, track_index, out, failed, user):
        e = self.engines[h]
        with self.lock:
            self.calls.append(("begin", e["device"], n))
            self.threads.add(threading.get_ident())
        pack = np.zeros(H + 2, np.uint32)
        peak = 0.0
        for i in range(n):
            try:
                r, d = self._one(paths[i].decode())
            except OSError:
                e["err"] = b"Failed to open: " + paths[i]
                failed[0] = i
                return -8
            if r is None:
                e["err"] = d["fail"].encode()
                failed[0] = i
                return int(d["code"])
            hist, loud, pk = r
            pack[:H] += hist
            peak = max(peak, pk)
            out[i].loudness_db, out[i].gain_db, out[i].peak = loud, self.lib.rg_gain_from_loudness(loud), pk
            out[i].sample_rate, out[i].windows = 44100, int(hist.sum())
        pack[H:] = np.array([peak]).view(np.uint32)
        e["pack"] = pack
        return 0

    def album_pack(self, h, pack_out, user):
        e = self.engines[h]
        with self.lock:
            self.calls.append(("pack", e["device"], 0))
        C.memmove(pack_out, e["pack"].ctypes.data, (H + 2) * 4)
        return 0

    def tracks(self, h, paths, n, track_index, out, status, user):
        e = self.engines[h]
        with self.lock:
            self.calls.append(("tracks", e["device"], n))
        e["terr"] = [b""] * n
        for i in range(n):
            try:
                r, d = self._one(paths[i].decode())
            except OSError:
                status[i], e["terr"][i] = -8, b"Failed to open: " + paths[i]
                continue
            if r is None:
                status[i], e["terr"][i] = int(d["code"]), d["fail"].encode()
                continue
            hist, loud, pk = r
            status[i] = 0
            out[i].loudness_db, out[i].gain_db, out[i].peak, out[i].sample_rate, out[i].windows = loud, 64.82 - loud, pk, 44100, int(hist.sum())
        return 0

    def _cstr(self, b):
        buf = C.create_string_buffer(b)
        self._keep.append(buf)  # const char * handed to C: must outlive the call
        return C.addressof(buf)

    def tracks_error(self, h, i, user):
        return self._cstr(self.engines[h]["terr"][i])

    def last_error(self, h, user):
        return self._cstr(self.engines[h]["err"])


def _write(tmp_path, k, bins=None, peak=0.5, pad=0, fail=None, code=-9):
    f = tmp_path / f"t{k:03d}.fake"
    body = {"fail": fail, "code": code} if fail else {"bins": bins, "peak": peak}
    f.write_text(json.dumps(body) + "\n" + " " * pad)  # the padding is the file's "length": files are dealt by size
    return f


def _album(tmp_path, n, seed=1):
    rng = np.random.default_rng(seed)
    files = []
    for k in range(n):
        bins = {int(b): int(c) for b, c in zip(rng.integers(5000, 9000, 6), rng.integers(1, 400, 6))}
        files.append(_write(tmp_path, k, bins, peak=float(rng.uniform(0.1, 1.2)), pad=int(rng.integers(0, 20000))))
    return files


def test_partition_rule():
    """Heaviest first, each to the least loaded device, ties to the lower index / lower device: the rule
    album.shard_indices(frames=...) uses for torchrun-launched ranks."""
    rng = np.random.default_rng(3)
    for world in (1, 2, 3, 8):
        for n in (0, 1, 5, 40):
            sizes = [int(x) for x in rng.integers(0, 50, n)]  # many ties
            own = R.node_partition(sizes, world)
            for r in range(world):
                assert [i for i in range(n) if own[i] == r] == album.shard_indices(n, world, r, frames=sizes)
            if n >= 4 * world and world > 1:
                load = [sum(s for s, o in zip(sizes, own) if o == r) for r in range(world)]
                assert max(load) - min(load) <= max(sizes)


@pytest.mark.parametrize("devices", [[0], [0, 1], [3, 1, 2], list(range(8))])
def test_album_over_fake_devices(tmp_path, devices):
    files = _album(tmp_path, 23)
    fe = FakeEngines()
    with R.Node(devices, _backend=fe.table) as node:
        assert node.devices == len(devices)
        got = node.analyze_album_files(files)
        own = node.last_partition(len(files))
    # the dealing is by file size
    import os

    assert own == R.node_partition([os.path.getsize(f) for f in files], len(devices))
    begins = sorted((d, n) for what, d, n in fe.calls if what == "begin")
    assert begins == sorted((devices[r], own.count(r)) for r in range(len(devices)))
    assert sum(1 for what, _, _ in fe.calls if what == "pack") == len(devices)
    if len(devices) > 1:
        assert len(fe.threads) == len(devices)  # one host thread per device
    # per-file results in input order, album = fold of everything
    total = np.zeros(H, np.uint32)
    peak = 0.0
    for f, t in zip(files, got.tracks):
        (hist, loud, pk), _ = fe._one(str(f))
        assert (t.loudness_db, t.peak) == (loud, pk)
        total += hist
        peak = max(peak, pk)
    want = album.album_result_from_hist(total, peak)
    assert (got.album_loudness_db, got.album_gain_db, got.album_peak) == (want["album_loudness_db"], want["album_gain_db"], want["album_peak"])
    assert all(e.get("closed") for e in fe.engines.values())


def test_first_failing_file_in_input_order_ends_the_album(tmp_path):
    """Two files fail, on different devices; the album's error is the one with the lower input index
    (src/replaygain.rs:1055 meets it first) and no device is asked for its pack."""
    files = _album(tmp_path, 12)
    import os

    for pad in range(100, 30000, 700):  # sizes such that the two failing files land on different devices
        files[9] = _write(tmp_path, 9, fail="Unsupported sample rate: 44000 Hz. Supported rates: ...", code=-2, pad=pad)
        files[4] = _write(tmp_path, 4, fail=f"Failed to probe format: {tmp_path}/t004.fake", code=-9, pad=100)
        own = R.node_partition([os.path.getsize(f) for f in files], 3)
        if own[4] != own[9]:
            break
    fe = FakeEngines()
    with R.Node([0, 1, 2], _backend=fe.table) as node:
        with pytest.raises(R.ReplayGainError) as ei:
            node.analyze_album_files(files)
        own = node.last_partition(len(files))
        assert own[4] != own[9], "the case needs the two failures on different devices"
        assert ei.value.code == -9 and "t004.fake" in str(ei.value)
        assert not [c for c in fe.calls if c[0] == "pack"]
        # a missing file: its owner reports it with the reference's text; and the node still works afterwards
        files2 = list(files)
        files2[4] = _album(tmp_path / ".", 5)[4]
        files2[9] = tmp_path / "missing.fake"
        with pytest.raises(R.ReplayGainError, match="Failed to open: .*missing.fake") as ei:
            node.analyze_album_files(files2)
        assert ei.value.code == -8
        files2[9] = _album(tmp_path, 10)[9]
        ok = node.analyze_album_files(files2)
        assert len(ok.tracks) == 12


def test_tracks_over_fake_devices(tmp_path):
    files = _album(tmp_path, 17)
    files[3] = _write(tmp_path, 3, fail="Failed to probe format: x", code=-9)
    files[11] = tmp_path / "nope.fake"
    fe = FakeEngines()
    with R.Node([0, 1, 2, 3], _backend=fe.table) as node:
        got = node.analyze_track_files(files)
        assert node.analyze_track_files([]) == []
    assert len(got) == 17
    for i, (f, g) in enumerate(zip(files, got)):
        if i == 3:
            assert isinstance(g, R.ReplayGainError) and g.code == -9 and "probe" in str(g)
        elif i == 11:
            assert isinstance(g, R.ReplayGainError) and g.code == -8 and "nope.fake" in str(g)
        else:
            (hist, loud, pk), _ = fe._one(str(f))
            assert (g.loudness_db, g.peak) == (loud, pk)


def test_empty_album_and_more_devices_than_files(tmp_path):
    fe = FakeEngines()
    with R.Node([0, 1, 2, 3], _backend=fe.table) as node:
        got = node.analyze_album_files([])
        assert got.tracks == [] and got.album_loudness_db == -20.0  # LoudnessHistogram::get_loudness on an empty histogram
        files = _album(tmp_path, 2)
        two = node.analyze_album_files(files)
        assert len(two.tracks) == 2


def test_creation_fails_loudly(tmp_path):
    fe = FakeEngines()
    with pytest.raises(R.ReplayGainError):
        R.Node([0, 666], _backend=fe.table)
    with pytest.raises(R.ReplayGainError, match="no usable gfx950 device|hip"):
        R.Node()  # the built-in engine needs a GPU: there is no CPU path
    # the RCCL exchange belongs to the built-in engine
    with R.Node([0], _backend=fe.table) as node:
        with pytest.raises(R.ReplayGainError):
            node.set_exchange(R.Node.EXCHANGE_RCCL)
