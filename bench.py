#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X: stereo PCM samples/s through the
IIR + 50 ms RMS + histogram path, with the dB delta against the CPU oracle.

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one pass of the hot path over one batch of synthetic PCM that is already resident
in HBM (generated there by the library's rg_synth kernel; include/rg_synth.h):

  N == 1   BASELINE configs[2], the largest single-GPU configuration: a batch of 1000 synthetic
           3-minute 44.1 kHz stereo tracks (7 938 000 frames each, 63.5 GB planar f32 -- far beyond
           the 256 MiB Infinity Cache), per-track histogram, percentile, gain, peak (-r mode).
           The same line carries configs[1] (ONE 10-minute track, 211.7 MB) as `configs1`.
  N  > 1   BASELINE configs[3]: album mode (-a), the same 1000 x 3-min batch on every rank (weak scaling,
           8000 tracks at N = 8; rank r owns tracks r, r + N, ... of the album); after the per-rank
           kernels the [12 000-bin album histogram | album peak] packs of all ranks are all-gathered
           over RCCL (one collective, on the stream of the batch) and folded on the device (sum / max),
           then every rank runs the album percentile.  `--scaling strong --total-tracks T` keeps the
           album fixed instead and shards it by cumulative frames (configs[3]'s 8000-track album on N GPUs:
           `--scaling strong --total-tracks 8000`); both curves can come out of one 8-GPU lease.  One rank per GPU,
           launched by torch.distributed.run.  Every N > 1 line carries `exchange` (ranks of the library's communicator,
           transport, ncclGetVersion), `parity` (rank 0's first tracks against the oracle, bin for bin) and `cpu_baseline`.

Timing: an untimed pre-roll (0.25 s worth of steps, so that clock and power have settled), W warm-up
steps, then exactly K steps between barrier + torch.cuda.synchronize() pairs; the time is the MAX
over ranks; value = frames processed by all ranks / that time.
Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (IIR+RMS+histogram), measured with HIP
events on the streams it is launched on.  It carries BOTH bounds: `hbm` (algorithmic 8 B per stereo frame /
8 TB/s -- what BASELINE asks for) and `fp64` (algorithmic 108 flop per stereo frame / 78.6 TFLOP/s FP64
vector FMA -- the one that binds: 13.5 flop/B is above the ridge).  `traffic` and the executed instruction
counts come from committed PMC passes of THIS workload (profiles/r04_pmc_<workload>.json), else null.
`cpu_baseline` is the CPU oracle (a C restatement of the reference's sequential algorithm -- not the Rust
binary, which cannot be built in this image) on a bounded sample of the same tracks.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
PRE_ROLL_SECONDS = 0.25
sys.path.insert(0, str(ROOT))

RATE = 44100
FRAMES_10MIN = 600 * RATE          # 26 460 000
FRAMES_3MIN = 180 * RATE           # 7 938 000
ALGO_BYTES_PER_FRAME = 8           # 2 channels x f32, read once
ALGO_FLOP_PER_FRAME = 108          # 2 x (21 + 5 + 1) FMA
HBM_PEAK_GBPS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_PEAK_TFLOPS = 78.6            # MI355X FP64 vector: 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz
PROFILE_ROUND = "r06"


class _DevArray:
    """Zero-copy view of a raw device pointer for torch.as_tensor (CUDA array interface)."""

    def __init__(self, ptr: int, shape, typestr: str):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (ptr, False), "version": 3}


def _usable_cores() -> int:
    """Host threads this process may really run: the affinity mask, cut by the cgroup CPU quota if there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def workload_tag(ntr: int, frames: int, album: bool) -> str:
    """Name under which PMC passes of a workload are committed (profiles/r04_pmc_<tag>.json)."""
    if ntr == 1000 and frames == FRAMES_3MIN:
        return "cfg3_album" if album else "cfg2"
    if ntr == 1 and frames == FRAMES_10MIN:
        return "cfg1"
    return f"{ntr}x{frames}{'_album' if album else ''}"


def pmc_for(tag: str, frames_per_launch: int):
    """The newest committed PMC pass of this workload (this round's if the kernel changed this round, else the last one taken)."""
    for rnd in (PROFILE_ROUND, "r04"):
        try:
            pm = json.loads((ROOT / "profiles" / f"{rnd}_pmc_{tag}.json").read_text())
        except (OSError, ValueError):
            continue
        if pm.get("frames_per_launch") == frames_per_launch:
            pm["_source"] = f"profiles/{rnd}_pmc_{tag}.json"
            return pm
    return None


def roofline_block(frames_per_launch: int, k_ms_sum: float, k_launches: int, k_span_ms: float, tag: str,
                   algo_bytes: int = 0, algo_flop: int = 0, launches_per_step: int = 1, channel_samples: int = 0) -> dict:
    """Both bounds for the dominant kernel.

    Consecutive batches run in separate pipeline slots, so several launches can be in flight at once:
      kernel_ms    average duration of one launch (what rocprofv3 --stats reports as the average)
      concurrency  sum of the launch durations / span from the first start to the last end
      achieved     algorithmic bytes (flops) of all launches / the span, i.e. what the launches consumed
                   divided by the time during which the kernel was running
    With one launch in flight at a time (concurrency 1.0) achieved = bytes per launch / kernel_ms."""
    k_ms = k_ms_sum / max(1, k_launches)
    conc = k_ms_sum / k_span_ms if k_span_ms > 0 else 1.0
    span_s = k_span_ms * 1e-3
    # per step (= per launch for a uniform batch; a mixed batch splits into launches_per_step launch groups)
    algo_bytes = algo_bytes or ALGO_BYTES_PER_FRAME * frames_per_launch
    algo_flop = algo_flop or ALGO_FLOP_PER_FRAME * frames_per_launch
    k_steps = k_launches / launches_per_step
    gbps = algo_bytes * k_steps / span_s / 1e9 if span_s > 0 else 0.0
    tflops = algo_flop * k_steps / span_s / 1e12 if span_s > 0 else 0.0
    pm = pmc_for(tag, frames_per_launch)
    traffic = pm["hbm_bytes_per_launch"] if pm else None
    fp64 = {"achieved": tflops, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / FP64_PEAK_TFLOPS,
            "algorithmic_flop_per_launch": algo_flop, "from_profiles": None}
    if pm and pm.get("valu_insts_per_launch"):
        # executed: PMC wave-instruction counts (mean per launch) x launches per step x 64 lanes; FMA = 2 flop
        chs = channel_samples or 2 * frames_per_launch
        ex = {"source": pm["_source"],
              "valu_wave_insts_per_launch": pm["valu_insts_per_launch"]}
        fma = pm.get("fma_f64_insts_per_launch")
        if fma:
            # the servo-form stage has two adds and one multiply among its 26 FP64 operations per channel-sample
            addmul = (pm.get("add_f64_insts_per_launch") or 0.0) + (pm.get("mul_f64_insts_per_launch") or 0.0)
            ex["fma_f64_wave_insts_per_launch"] = fma
            ex["fma_f64_per_channel_sample"] = fma * launches_per_step * 64.0 / chs
            ex["fp64_ops_per_channel_sample"] = (fma + addmul) * launches_per_step * 64.0 / chs
            ex["tflops"] = 64.0 * (2.0 * fma + addmul) * k_launches / span_s / 1e12 if span_s > 0 else 0.0
            ex["frac_of_peak"] = ex["tflops"] / FP64_PEAK_TFLOPS
        else:  # no FP64-specific counter: every VALU instruction priced as an FMA (upper bound on the flops)
            ex["tflops_upper_bound"] = 2.0 * 64.0 * pm["valu_insts_per_launch"] * k_launches / span_s / 1e12 if span_s > 0 else 0.0
        ex["valu_per_channel_sample"] = pm["valu_insts_per_launch"] * launches_per_step * 64.0 / chs
        fp64["from_profiles"] = ex  # NOT measured in this run: counters of a committed PMC pass of the same workload (ex["source"])
    return {"bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS,
            "traffic": traffic,
            "binding_bound": "fp64 (vector FMA): 13.5 algorithmic flop/B is above the 9.8 flop/B ridge, see roofline.fp64",
            "fp64": fp64,
            "kernel": "rg_tm_main_kernel", "kernel_ms": k_ms, "kernel_launches": int(k_launches),
            "kernel_concurrency": conc, "kernel_span_ms": k_span_ms,
            "launch_groups_per_step": launches_per_step,
            "algorithmic_bytes_per_launch": algo_bytes // launches_per_step}


def mp3_end_to_end(an, nfiles: int) -> dict:
    """rg_analyze_album over `nfiles` three-minute MP3 files (the frames of tests/golden/mp3/v1_44k_stereo_long.mp3, a
    dense 320 kb/s 44.1 kHz stereo stream, repeated), with each of the library's four decode routes (tuning key 6)."""
    import tempfile

    from mp3rgain_amd import mp3dec

    body = (ROOT / "tests" / "golden" / "mp3" / "v1_44k_stereo_long.mp3").read_bytes()
    one = mp3dec.scan(body)
    reps = max(1, int(180.0 / (one.frames / one.sample_rate)))
    stream = body * reps
    si = mp3dec.scan(stream)
    tmp = Path(tempfile.mkdtemp(prefix="rg_bench_mp3_"))
    files = []
    for k in range(nfiles):
        p = tmp / f"t{k:04d}.mp3"
        p.write_bytes(stream)
        files.append(p)
    leg = {"workload": f"{nfiles} files x {si.frames / si.sample_rate:.0f} s, 320 kb/s 44.1 kHz stereo Layer III ({len(stream) / 1e6:.1f} MB each), "
                       "rg_analyze_album(paths): file read + decode + analysis + album percentile",
           "unit": "stereo samples/s", "routes": {}}
    try:
        # untimed: every pipeline slot's buffers grow to this batch's size once (grow-only allocations), and the host-to-device
        # path itself takes about a second of sustained copies to reach its rate (the first dozen calls of a process run at
        # 37 GB/s, later ones at 47: tools/ab_parts.py, whichever stream comes first)
        t_warm = time.perf_counter()
        for k in range(64):
            an.analyze_album_files(files)
            if k >= 7 and time.perf_counter() - t_warm > 1.5:
                break
        for mode, name in ((3, "device: host strips headers only, pipelined"), (2, "device: host parses side info"), (1, "split: Huffman on host"), (0, "host decoder")):
            an.set_tuning(6, mode)
            sub = files if mode >= 2 else files[:min(nfiles, 64)]  # the host-bound routes on fewer files: they take seconds
            an.analyze_album_files(sub[:2])
            dt = 1e9
            for _ in range(2 if mode else 1):
                tm = {}
                t0 = time.perf_counter()
                res = an.analyze_album_files(sub, timing=tm)
                dt = min(dt, tm["c_call_seconds"])  # the C call rg_analyze_album alone (the ctypes wrapper adds a few us per file)
                if os.environ.get("RG_TRACE_FILES"):
                    print(f"[bench] mode {mode}: {time.perf_counter() - t0:.3f} s", file=sys.stderr)
            leg["routes"][name] = {"files": len(sub), "seconds": dt, "value": len(sub) * si.frames / dt,
                                   "x_real_time": len(sub) * si.frames / si.sample_rate / dt, "album_loudness_db": res.album_loudness_db}
    finally:
        an.set_tuning(6, 3)
        for p in files:
            p.unlink()
        tmp.rmdir()
    loud = {r["album_loudness_db"] for r in leg["routes"].values()}
    leg["routes_agree"] = len(loud) == 1
    # ---- the same call on the two other kinds of stream the round's end-to-end claims are made on (default route only): an
    # encoder-made 128 kb/s joint-stereo stream and the reference's own VBR fixture, both repeated to three minutes.  The
    # 320 kb/s album above is bound by the host-to-device copy of its compressed bytes; these two are not.  Album loudness
    # is checked against the host decoder's route on 8 of the files (all files of a stream are equal, so the album's
    # percentile does not depend on how many there are).
    leg["streams"] = {}
    for label, src in (("vbr_fixture", ROOT / "tests" / "golden" / "fixtures" / "test_vbr.mp3"),
                       ("dense128_joint", ROOT / "tests" / "golden" / "mp3" / "dense_44k_joint_128.mp3")):
        try:
            data = src.read_bytes()
            body2 = data[int(mp3dec.scan(data).first_frame_offset):]
            one2 = mp3dec.scan(body2)
            stream2 = body2 * max(1, int(180.0 / (one2.frames / one2.sample_rate)))
            si2 = mp3dec.scan(stream2)
            tmp2 = Path(tempfile.mkdtemp(prefix="rg_bench_mp3_"))
            files2 = []
            for k in range(nfiles):
                p2 = tmp2 / f"t{k:04d}.mp3"
                p2.write_bytes(stream2)
                files2.append(p2)
            try:
                for _ in range(3):
                    an.analyze_album_files(files2)
                dt2, res2 = 1e9, None
                for _ in range(3):
                    tm = {}
                    res2 = an.analyze_album_files(files2, timing=tm)
                    dt2 = min(dt2, tm["c_call_seconds"])
                an.set_tuning(6, 0)
                host = an.analyze_album_files(files2[:min(nfiles, 8)])
                an.set_tuning(6, 3)
                leg["streams"][label] = {
                    "files": nfiles, "seconds": dt2, "value": nfiles * si2.frames / dt2, "x_real_time": nfiles * si2.frames / si2.sample_rate / dt2,
                    "bytes_per_file": len(stream2), "kbps": len(stream2) * 8 / (si2.frames / si2.sample_rate) / 1e3,
                    "album_loudness_db": res2.album_loudness_db, "album_loudness_db_host_decoder_route": host.album_loudness_db,
                    "agrees_with_host_decoder_route": res2.album_loudness_db == host.album_loudness_db}
            finally:
                an.set_tuning(6, 3)
                for p2 in files2:
                    p2.unlink()
                tmp2.rmdir()
        except Exception as ex:  # noqa: BLE001
            leg["streams"][label] = {"error": str(ex)}
    # ---- how many of these decoded tracks does the fast kernel flag (RG_TRACK_FLAG_IMPRECISE), and what does the exact
    # repeat of the synchronous entry points cost then?  The file-level calls above return the repeated (exact) results, whose
    # flag is gone; here the decoded PCM of one file, 16 copies, goes through the asynchronous pair (hands the flag over)
    # and through the synchronous call.
    try:
        import numpy as np
        import torch

        from mp3rgain_amd.replaygain import PcmTrack, pack_tracks

        pcm1, di = an.decode_mp3_device(stream)
        trk = [PcmTrack([np.ascontiguousarray(pcm1[c]) for c in range(pcm1.shape[0])], int(di.sample_rate))] * 16
        arena, descs_f = pack_tracks(trk)
        dev = torch.from_numpy(arena).cuda()
        for _ in range(2):
            an.enqueue_device(descs_f, 16, dev.data_ptr(), arena.nbytes)
            rf = an.collect(16)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        an.enqueue_device(descs_f, 16, dev.data_ptr(), arena.nbytes)
        rf = an.collect(16)
        t_async = time.perf_counter() - t0
        an.analyze_device(descs_f, 16, dev.data_ptr(), arena.nbytes)
        t0 = time.perf_counter()
        an.analyze_device(descs_f, 16, dev.data_ptr(), arena.nbytes)
        t_sync = time.perf_counter() - t0
        leg["tracks_flagged_imprecise"] = {"tracks": 16, "flagged": sum(1 for r in rf if r.flags & 2),
                                           "async_pair_ms": t_async * 1e3, "synchronous_call_ms": t_sync * 1e3,
                                           "note": "decoded PCM of one of the files, 16 copies resident in HBM; the synchronous call repeats a batch "
                                                   "that has a flagged track with it on the order-faithful kernel; profiles/r06_h10_flag_rate.txt has "
                                                   "every golden stream"}
        del dev
    except Exception as ex:  # noqa: BLE001
        leg["tracks_flagged_imprecise"] = {"error": str(ex)}
    # ---- the decode chain alone, measured like the headline kernel: HIP events on the stream the kernels run on
    # (rg_mp3_decode_bench), algorithmic bytes = compressed bytes in + 4 bytes per decoded sample out ----
    try:
        units_per = si.audio_frames * (2 if si.mpeg_version == 1 else 1) * si.channels
        copies = max(1, round(786432 / units_per))
        ch = an.decode_mp3_bench(stream, copies, reps=30)
        algo = ch["compressed_bytes"] + 4 * ch["frames"] * si.channels
        chain_s = ch["ms"]["chain"] * 1e-3
        k256 = (1 << 18) / ch["units"]
        dominant = max(("frames", "huffman", "backhalf"), key=lambda n: ch["ms"][n])
        traffic = None
        try:
            pm = json.loads((ROOT / "profiles" / f"{PROFILE_ROUND}_pmc_mp3.json").read_text())
            traffic = pm["hbm_bytes_per_unit"] * ch["units"]  # committed PMC pass of tools/mp3_chain.py, scaled by units
        except (OSError, ValueError, KeyError, TypeError):
            pass
        pcie_gbps = 50.0  # measured H2D rate of this pool's boards; spec 63 GB/s
        leg["roofline"] = {
            "bound": "hbm", "achieved": algo / chain_s / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": algo / chain_s / 1e9 / HBM_PEAK_GBPS, "traffic": traffic,
            "kernel": f"rg_mp3_{dominant}_kernel (longest of the chain: frames -> huffman -> backhalf)",
            "kernel_ms": ch["ms"][dominant], "chain_ms": ch["ms"]["chain"], "kernels_ms": ch["ms"],
            "units_per_launch": ch["units"], "ms_per_256k_units": {n: v * k256 for n, v in ch["ms"].items()},
            "algorithmic_bytes_per_launch": algo, "compressed_bytes_per_launch": ch["compressed_bytes"],
            "stereo_samples_per_s": ch["frames"] / chain_s,
            "binding_bound": "instruction issue and LDS/memory latency of short dependent stages (integer bit parsing, table look-ups, "
                             "per-subband transforms), not HBM: the chain moves a few percent of what HBM could",
            "file_route_pcie_bound": {"compressed_bytes_per_stereo_sample": len(stream) / si.frames,
                                      "h2d_GBps_assumed": pcie_gbps,
                                      "stereo_samples_per_s": pcie_gbps * 1e9 / (len(stream) / si.frames)},
            "note": "HIP events around each kernel on the stream it runs on; one chunk of `units_per_launch` granule-channels, 30 repetitions "
                    "back to back; the compressed bytes are copied H2D per repetition on the copy stream, beside the kernels; `chain` has the "
                    "frame parser and lane sort in line between the events, `chain_pipelined` is per chunk the way the file route enqueues "
                    "chunks (parser and sort on the copy stream, beside the Huffman / back-half kernels of the chunk before)"}
    except Exception as ex:  # noqa: BLE001
        leg["roofline"] = {"error": str(ex)}
    return leg


def cpu_baseline_leg(po, l0, r0, rate0: int, fr0: int, label: str, cpu_seconds: float) -> dict:
    """The CPU oracle (oracle/rg_oracle.c, a C restatement of replaygain.rs -- not the Rust binary) on a bounded sample of the
    workload: one of its tracks, repeated for `cpu_seconds` on one thread (the reference is single-threaded), then
    the same code on every usable host core with tracks as the parallel unit."""
    c0 = time.perf_counter()
    reps = 0
    while True:
        po.analyze_pcm(l0, r0, rate0)
        reps += 1
        if time.perf_counter() - c0 >= cpu_seconds:
            break
    cdt = time.perf_counter() - c0
    from concurrent.futures import ThreadPoolExecutor

    nthr = _usable_cores()
    per_thread = max(1, int(reps * min(1.0, 4.0 / max(cdt, 1e-9))))  # about 4 s per thread
    m0 = time.perf_counter()
    with ThreadPoolExecutor(nthr) as pool:  # ctypes releases the GIL inside the C call
        list(pool.map(lambda _: [po.analyze_pcm(l0, r0, rate0) for _ in range(per_thread)], range(nthr)))
    mdt = time.perf_counter() - m0
    return {"value": fr0 * reps / cdt, "unit": "stereo samples/s", "cores": 1, "kind": "port",
            "sample": f"{label} ({fr0} stereo frames @{rate0 / 1000:g} kHz) x{reps} = {cdt:.1f} s, "
                      f"oracle/rg_oracle.c (C restatement of replaygain.rs, not the Rust binary), 1 thread, "
                      f"host has {os.cpu_count()} cores, {_usable_cores()} usable by this process",
            "all_cores": {"value": fr0 * per_thread * nthr / mdt, "cores": nthr,
                          "sample": f"the same track x{per_thread} on each of {nthr} threads (tracks as the parallel unit)"}}


def exchange_block(an, transport: str) -> dict:
    """What the album exchange of a multi-GPU line ran over: ranks the library's communicator spans, the transport, the
    collective library's version (ncclGetVersion; 0 = the library has no such entry point, e.g. the tests' stand-in)."""
    info = an.comm_info()
    return {"ranks": info["ranks"], "transport": transport, "nccl_version": info["nccl_version"],
            "collective": "one all-gather of 12 002-word [histogram | peak] packs + device fold, on the batch's stream"}


def node_main(args) -> int:
    """`--node` / `--gpus N` without torch.distributed.run: the album workload of configs[3] in ONE process.  A node
    (mp3rgain_amd.Node: one context per GPU) with the library's in-process RCCL communicators (ncclCommInitAll), one host
    thread per GPU: every thread fills its device's arena with its share of the album (tracks i, i + N, ...), and a step is
    enqueue -> all-gather of the 48 KB album packs + fold on the batch's stream -> album percentile, exactly the calls a
    torchrun rank makes.  Threads meet at a barrier before and after the timed steps; the time is the slowest thread's."""
    import threading

    import numpy as np

    import torch

    import mp3rgain_amd as rg
    from mp3rgain_amd import _capi

    n_dev = args.gpus
    rehearsal = os.environ.get("RG_BENCH_REHEARSAL") == "1"  # see main(): contexts share devices, stand-in transport
    if not torch.cuda.is_available() or (torch.cuda.device_count() < n_dev and not rehearsal):
        raise SystemExit(f"bench.py --node --gpus {n_dev}: {torch.cuda.device_count()} device(s) visible; there is no CPU path")
    dev_of = [i % torch.cuda.device_count() for i in range(n_dev)] if rehearsal else list(range(n_dev))
    if rehearsal:
        standin = os.environ.get("MP3RGAIN_AMD_RCCL_LIBRARY")
        if not standin or _capi.load().rg_comm_library(os.fsencode(standin)) != 0:
            raise SystemExit("RG_BENCH_REHEARSAL=1 needs MP3RGAIN_AMD_RCCL_LIBRARY=<tests/standin_rccl/librccl_standin.so>")
    frames = int(round(args.minutes * 60 * RATE))
    node = rg.Node(dev_of)
    node.set_exchange(rg.Node.EXCHANGE_RCCL)  # one communicator per device, built in this process
    total_tracks = args.tracks_per_rank * n_dev
    gate = threading.Barrier(n_dev)
    out = [None] * n_dev
    errors = []

    def device_thread(i: int):
        try:
            torch.cuda.set_device(dev_of[i])
            an = node.analyzer(i)
            if args.kernel:
                an.set_kernel(args.kernel)
            mine = list(range(i, total_tracks, n_dev))
            ntr = len(mine)
            pcm = torch.empty(max(1, 2 * frames * ntr), dtype=torch.float32, device=f"cuda:{dev_of[i]}")
            descs = (_capi.TrackDesc * max(1, ntr))()
            for t, g in enumerate(mine):
                off = 2 * frames * t
                for c in range(2):
                    an.synth_fill_device(pcm[off + c * frames:].data_ptr(), 0x5EED0000 + g, c, RATE, 0, frames)
                descs[t].offset_bytes = off * 4
                descs[t].frames = frames
                descs[t].sample_rate = RATE
                descs[t].channels = 2
                descs[t].format = _capi.FMT_F32_PLANAR
            torch.cuda.synchronize(dev_of[i])

            def step():
                an.enqueue_device(descs, ntr, pcm.data_ptr(), pcm.numel() * 4, album=True)
                an.album_exchange()
                an.album_result_enqueue()

            pre_steps = max(2, min(8192, int(args.pre_roll / (max(1, frames * ntr) / 3.5e11))))
            for k in range(pre_steps + args.warmup):
                step()
                if k % 64 == 63:
                    torch.cuda.synchronize(dev_of[i])
            torch.cuda.synchronize(dev_of[i])
            gate.wait()
            an.timing_enable(True)
            an.timing_read(reset=True)
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize(dev_of[i])
            gate.wait()
            dt = time.perf_counter() - t0
            ks, kl, ksp = an.timing_read(reset=True)
            an.timing_enable(False)
            res, hists = an.collect(ntr, want_hist=True) if i == 0 else (an.collect(ntr), None)
            alb = an.album_finish()
            npar = min(args.parity_tracks, ntr)
            out[i] = {"dt": dt, "frames": frames * ntr, "k": (ks, kl, ksp), "album": alb, "first": res[0] if res else None,
                      "flagged": sum(1 for r in res if r.flags & 2), "seeds": [0x5EED0000 + g for g in mine[:npar]],
                      "res": res[:npar], "hists": hists[:npar].copy() if hists is not None else None,
                      "exchange": exchange_block(an, "rccl (library communicators of this process, ncclCommInitAll)")}
        except Exception as ex:  # noqa: BLE001
            errors.append(f"device {i}: {ex}")
            gate.abort()

    threads = [threading.Thread(target=device_thread, args=(i,)) for i in range(n_dev)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise SystemExit("; ".join(errors))
    dt = max(o["dt"] for o in out)
    total_frames = sum(o["frames"] for o in out) * args.steps
    ks, kl, ksp = out[0]["k"]
    tag = workload_tag(args.tracks_per_rank, frames, True)
    roof = roofline_block(out[0]["frames"], ks, kl, ksp, tag)
    alb0 = out[0]["album"]
    parity = cpu = None
    if args.cpu_seconds > 0 and out[0]["seeds"]:
        # device 0's first tracks of the album (global indices 0, n_dev, 2 n_dev, ...) against the oracle, bin for bin
        from oracle import pyoracle as po

        diff_bins, max_db, peaks_equal = 0, 0.0, True
        for seed_t, r_t, h_t in zip(out[0]["seeds"], out[0]["res"], out[0]["hists"]):
            l, r = po.synth_f32(seed_t, 0, RATE, frames), po.synth_f32(seed_t, 1, RATE, frames)
            want, want_hist = po.analyze_pcm(l, r, RATE)
            diff_bins += int(np.count_nonzero(h_t != want_hist))
            max_db = max(max_db, abs(r_t.loudness_db - want["loudness_db"]))
            peaks_equal = peaks_equal and r_t.peak == want["peak"]
        parity = {"tracks_compared": len(out[0]["seeds"]), "which": "device 0's first tracks of the album", "differing_histogram_bins": diff_bins,
                  "max_abs_db_delta": max_db, "peaks_equal": peaks_equal,
                  "every_device_agrees_on_the_album": len({(o["album"].album_loudness_db, o["album"].album_peak) for o in out}) == 1,
                  "tracks_flagged_imprecise": sum(o["flagged"] for o in out)}
        seed0 = out[0]["seeds"][0]  # the CPU baseline's sample: the album's first track, regenerated here
        cpu = cpu_baseline_leg(po, po.synth_f32(seed0, 0, RATE, frames), po.synth_f32(seed0, 1, RATE, frames), RATE, frames,
                               "the first track of the album", args.cpu_seconds)
    line = {
        "metric": "stereo PCM samples/s through IIR+RMS+histogram", "value": total_frames / dt, "unit": "stereo samples/s",
        "n_gpus": n_dev, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[3] shape: album mode, {args.tracks_per_rank} synthetic "
                               f"{args.minutes:g}-min 44.1 kHz stereo tracks per GPU per step, resident in HBM",
                   "launch": "ONE process: rg_node, one context + one host thread per GPU, album packs exchanged over the "
                             "library's in-process RCCL communicators (ncclCommInitAll)",
                   "tracks_per_gpu": args.tracks_per_rank, "album_tracks": total_tracks},
        "roofline": roof,
        "exchange": out[0]["exchange"],
        "cpu_baseline": cpu,
        "parity": parity,
        "result": {"album_loudness_db": alb0.album_loudness_db if alb0 else None,
                   "album_gain_db": alb0.album_gain_db if alb0 else None,
                   "every_device_agrees": len({(o["album"].album_loudness_db, o["album"].album_peak) for o in out}) == 1,
                   "tracks_flagged_imprecise": sum(o["flagged"] for o in out)},
    }
    if rehearsal:
        line["rehearsal"] = ("NOT A MEASUREMENT: contexts share devices, the library communicators run over the tests' stand-in "
                             "transport (RG_BENCH_REHEARSAL=1)")
    print(json.dumps(line))
    return 0


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--kernel", type=int, default=0, help="kernel variant (0 = library default)")
    ap.add_argument("--tracks-per-rank", type=int, default=1000, help="tracks per GPU and step (default: configs[2] / configs[3])")
    ap.add_argument("--minutes", type=float, default=3.0, help="track length (default: configs[2]'s 3 min)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: --tracks-per-rank on every GPU; strong: --total-tracks sharded over the GPUs by cumulative frames")
    ap.add_argument("--total-tracks", type=int, default=0, help="album size for --scaling strong (default: --tracks-per-rank)")
    ap.add_argument("--tm-segment", type=int, default=0, help="force variant 2 segment length (tuning)")
    ap.add_argument("--tm-windows", type=int, default=0, help="most windows per lane of variant 2 (tuning; 0 = library default, 1 = one)")
    ap.add_argument("--slots", type=int, default=0, help="pipeline slots of the library (0 = default)")
    ap.add_argument("--mixed", action="store_true",
                    help="BASELINE configs[4]'s PCM side instead: half the tracks at 44.1 kHz, half at 48 kHz, every 10th mono, "
                         "every 20th with full-scale (clipped) peaks")
    ap.add_argument("--album", action="store_true", help="force the album path (collectives) even on one GPU")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="CPU time budget of each cpu_baseline leg (0 = skip baseline and parity)")
    ap.add_argument("--parity-tracks", type=int, default=16, help="tracks of the batch whose full histogram is compared with the oracle")
    ap.add_argument("--no-configs1", action="store_true", help="skip the secondary configs[1] measurement")
    ap.add_argument("--no-mp3", action="store_true", help="skip the MP3 end-to-end leg (decode + analysis from compressed files)")
    ap.add_argument("--no-one-shot", action="store_true", help="skip the synchronous-call leg (profiling passes of the pipelined workload)")
    ap.add_argument("--mp3-files", type=int, default=256, help="files of the MP3 end-to-end leg (3-minute 320 kb/s streams; the two host-bound routes run on the first 64)")
    ap.add_argument("--configs1-steps", type=int, default=300)
    ap.add_argument("--pre-roll", type=float, default=PRE_ROLL_SECONDS, help="untimed pre-roll before the warm-up steps, seconds of work")
    ap.add_argument("--node", action="store_true",
                    help="ONE process for all --gpus devices (include/mp3rgain_amd_node.h: one context and one host thread per GPU, "
                         "the library's in-process RCCL communicators for the album exchange) instead of one rank per GPU; "
                         "also what `--gpus N` does when it is not launched through torch.distributed.run")
    args = ap.parse_args()

    import numpy as np
    import torch

    import mp3rgain_amd as rg
    from mp3rgain_amd import _capi
    from mp3rgain_amd import album as album_mod

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.node or (world == 1 and args.gpus > 1):
        return node_main(args)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU path")
    # Rehearsal (RG_BENCH_REHEARSAL=1; tests/test_gpu_multirank.py): the world > 1 code of this file on a box with fewer GPUs
    # than ranks.  Ranks share devices (rank % device count), torch.distributed runs over gloo and the library's communicator
    # over the tests' stand-in transport (MP3RGAIN_AMD_RCCL_LIBRARY) -- RCCL itself refuses two ranks on one device.  The line
    # says so and is not a measurement.
    rehearsal = os.environ.get("RG_BENCH_REHEARSAL") == "1"
    device = local_rank % torch.cuda.device_count() if rehearsal else local_rank
    torch.cuda.set_device(device)
    frames = int(round(args.minutes * 60 * RATE))
    # which tracks of the album this rank owns (global indices; the seed of a track is 0x5EED0000 + its index)
    if args.scaling == "strong":
        total_tracks = args.total_tracks or args.tracks_per_rank
        mine = album_mod.shard_indices(total_tracks, world, rank, frames=[frames] * total_tracks)
    else:
        total_tracks = args.tracks_per_rank * world
        mine = album_mod.shard_indices(total_tracks, world, rank)
    ntr = len(mine)
    # The context first: its pipeline streams should each get a hardware queue of their own (the runtime has 4
    # per process and deals them out as streams are created; torch.distributed / RCCL create several more).
    an = rg.Analyzer(device)
    dist = None
    if world > 1 or (args.album and "RANK" in os.environ):
        import torch.distributed as dist  # noqa: PLC0415

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    if args.kernel:
        an.set_kernel(args.kernel)
    if args.tm_segment:
        an.set_tuning(1, args.tm_segment)
    if args.slots:
        an.set_tuning(3, args.slots)
    if args.tm_windows:
        an.set_tuning(4, args.tm_windows)
    # No caller stream is attached: every batch, its album tail and the collective in between run on the
    # context's own pipeline streams (rg_batch_stream), which costs no cross-stream event per step.

    # ---- synthetic PCM straight into HBM ---------------------------------------------------
    def make_batch(specs):
        """specs: [(seed, rate, channels, frames)] -> (arena tensor, descriptors, bytes)"""
        n_tracks = len(specs)
        total = sum(ch * fr for _, _, ch, fr in specs)
        t_pcm = torch.empty(max(1, total), dtype=torch.float32, device="cuda")
        t_descs = (_capi.TrackDesc * max(1, n_tracks))()
        off = 0
        for t, (seed, rate, ch, fr) in enumerate(specs):
            for c in range(ch):
                an.synth_fill_device(t_pcm[off + c * fr:].data_ptr(), seed, c, rate, 0, fr)
            t_descs[t].offset_bytes = off * 4
            t_descs[t].frames = fr
            t_descs[t].sample_rate = rate
            t_descs[t].channels = ch
            t_descs[t].format = _capi.FMT_F32_PLANAR
            off += ch * fr
        torch.cuda.synchronize()
        return t_pcm, t_descs, total * 4

    HOT = 1 << 40  # RG_SYNTH_HOT_BIT: boosted and hard-clipped, peak >= 1.0 (the -k rule's input)
    if args.mixed:
        specs = []
        for k, g in enumerate(mine):
            rate = RATE if k < (ntr + 1) // 2 else 48000
            specs.append((0x5EED0000 + g + (HOT if g % 20 == 7 else 0), rate, 1 if g % 10 == 9 else 2,
                          int(round(args.minutes * 60 * rate))))
    else:
        specs = [(0x5EED0000 + g, RATE, 2, frames) for g in mine]
    seeds = [sp[0] for sp in specs]
    pcm, descs, pcm_bytes = make_batch(specs)
    batch_frames = sum(sp[3] for sp in specs)                    # frames this rank analyses per step
    batch_algo_bytes = sum(4 * sp[2] * sp[3] for sp in specs)    # f32, each channel read once
    batch_algo_flop = sum(54 * sp[2] * sp[3] for sp in specs)    # 27 FMA per channel-sample

    album = world > 1 or args.album
    # The album exchange (LoudnessHistogram::accumulate / album_peak.max across ranks, replaygain.rs:1056-1059) is
    # ONE collective per step: all-gather of the 48 KB [histogram | peak] packs + a device fold.  It runs over a
    # communicator the library owns (bootstrapped through torch.distributed), on the stream of the batch, so a
    # step costs no cross-stream event.  Fallback if that communicator cannot be built: the same all-gather
    # through torch.distributed, issued on the batch's stream.
    exchange = "none"
    if dist is not None:
        try:
            an.comm_init_torch()
            exchange = "rccl (library communicator, batch stream)"
        except Exception as ex:  # noqa: BLE001
            print(f"[bench] library communicator unavailable ({ex}); falling back to torch.distributed", file=sys.stderr)
            exchange = "torch.distributed all_gather_into_tensor (batch stream)"
    ext_streams = {}  # torch views of the library's pipeline streams (fallback path)
    views = {}  # the library rotates through pipeline slots: one set of tensor views / gather buffers per slot

    def step():
        an.enqueue_device(descs, ntr, pcm.data_ptr(), pcm_bytes, album=album)
        if album:
            if exchange.startswith("rccl"):
                an.album_exchange()
            elif dist is not None:
                view = an.device_view()
                if view.d_album_hist not in views:
                    views[view.d_album_hist] = (
                        torch.as_tensor(_DevArray(view.d_album_hist, (album_mod.ALBUM_PACK_WORDS,), "<i4"), device="cuda"),
                        torch.empty(world * album_mod.ALBUM_PACK_WORDS, dtype=torch.int32, device="cuda"))
                pack_t, gathered_t = views[view.d_album_hist]
                h = an.batch_stream()
                if h not in ext_streams:
                    ext_streams[h] = torch.cuda.ExternalStream(h)
                with torch.cuda.stream(ext_streams[h]):
                    album_mod.allgather_album(pack_t, gathered_t)
                an.album_reduce_gathered(gathered_t.data_ptr(), world)
            an.album_result_enqueue()

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(step_fn, n_steps: int, n_warm: int, work_frames: int):
        """pre-roll + warm-up + exactly n_steps timed steps -> (seconds, kernel ms sum, launches, span ms)"""
        # Pre-roll, untimed and before the warm-up steps: the shader clock and the power controller take some
        # milliseconds to settle once the kernel starts running (the first ~10 ms run 10-15 % slower), and a short
        # --warmup would otherwise put that ramp into the timed region.  The timed region is exactly K steps.
        # The step count comes from the workload size, not from a clock: every rank must issue the same collectives.
        pre_steps = max(2, min(8192, int(args.pre_roll / (work_frames / 3.5e11))))
        for i in range(pre_steps):
            step_fn()
            if i % 64 == 63:
                torch.cuda.synchronize()  # keep the host from running seconds ahead of the device
        for _ in range(n_warm):
            step_fn()
        fence()
        an.timing_enable(True)
        an.timing_read(reset=True)
        t0 = time.perf_counter()
        for _ in range(n_steps):
            step_fn()
        fence()
        dt_ = time.perf_counter() - t0
        ks, kl, ksp = an.timing_read(reset=True)
        an.timing_enable(False)
        return dt_, ks, kl, ksp

    # (the pre-roll's step count must be the same on every rank -- each step is a collective -- so it is sized from the
    # album's frames per rank, not from this rank's share, which differs under --scaling strong)
    dt, k1_ms_sum, k1_launches, k1_span_ms = timed(step, args.steps, args.warmup,
                                                   max(1, batch_frames if world == 1 else (total_tracks * frames) // world))

    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # ---- results of the last step (outside the timed region) ----
    res, hists = an.collect(ntr, want_hist=True)
    alb = an.album_finish() if album else None
    n_imprecise = sum(1 for r in res if r.flags & 2)
    n_nonfinite = sum(1 for r in res if r.flags & 1)

    # ---- one batch in flight: the synchronous entry point (rg_analyze_pcm_batch on the resident arena) -- the shape the
    # reference's blocking API has.  Nothing overlaps here: the dominant kernel's HIP-event duration IS one launch alone.
    one_shot = None
    if world == 1 and not album and ntr > 0 and not args.no_one_shot:
        for _ in range(3):
            an.analyze_device(descs, ntr, pcm.data_ptr(), pcm_bytes)
        an.timing_enable(True)
        an.timing_read(reset=True)
        calls = max(3, min(10, args.steps))
        call_ms = []
        raw1 = (_capi.TrackResult * ntr)()
        for _ in range(calls):
            c0 = time.perf_counter()
            an.analyze_device(descs, ntr, pcm.data_ptr(), pcm_bytes, out=raw1)  # the C call alone: launch .. results in host memory
            call_ms.append((time.perf_counter() - c0) * 1e3)
        res1 = an.analyze_device(descs, ntr, pcm.data_ptr(), pcm_bytes)
        ks1, kl1, _ = an.timing_read(reset=True)
        an.timing_enable(False)
        groups1 = len({(sp[1], sp[2]) for sp in specs}) or 1
        kern_ms = ks1 / max(1, kl1) * groups1          # per call: the launch groups of a call run one after the other
        mean_ms = sum(call_ms) / len(call_ms)
        one_shot = {"entry_point": "rg_analyze_pcm_batch(pcm_on_device = 1): one batch in flight, results on return",
                    "calls": calls, "ms_per_call": mean_ms, "min_ms": min(call_ms),
                    "value": batch_frames / (mean_ms * 1e-3), "unit": "stereo samples/s",
                    "hbm_frac_call": batch_algo_bytes / (mean_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                    "kernel": "rg_tm_main_kernel", "kernel_ms_alone": kern_ms, "kernel_launches": int(kl1),
                    "achieved_one_launch_alone": batch_algo_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0,
                    "hbm_frac_one_launch_alone": batch_algo_bytes / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS if kern_ms > 0 else 0.0,
                    "same_results_as_pipelined": all(a.loudness_db == b.loudness_db and a.peak == b.peak for a, b in zip(res, res1))}

    if dist is not None:  # every rank's tracks, every step
        tf = torch.tensor([batch_frames], dtype=torch.int64, device="cuda")
        dist.all_reduce(tf)
        total_frames = int(tf.item()) * args.steps
    else:
        total_frames = batch_frames * args.steps
    value = total_frames / dt
    tag = "cfg5" if args.mixed and ntr == 1000 and args.minutes == 3.0 else workload_tag(ntr, frames, album)
    # a mixed batch is several launch groups (rate x channel count) per step: the per-launch figures are per group
    groups = len({(sp[1], sp[2]) for sp in specs}) or 1
    roof = roofline_block(batch_frames, k1_ms_sum, k1_launches, k1_span_ms, tag, algo_bytes=batch_algo_bytes,
                          algo_flop=batch_algo_flop, launches_per_step=groups,
                          channel_samples=sum(sp[2] * sp[3] for sp in specs))
    if one_shot:
        # `kernel_ms` above is the OVERLAPPED duration of a launch (kernel_concurrency launches are resident at once on the
        # pipeline streams, so it exceeds ms_per_step); this is the same kernel with nothing beside it (one_shot below)
        roof["kernel_ms_alone"] = one_shot["kernel_ms_alone"]
        roof["frac_one_launch_alone"] = one_shot["hbm_frac_one_launch_alone"]

    # ---- secondary workload on one GPU: configs[1], one 10-minute track (fits the Infinity Cache) ----------
    configs1 = None
    if world == 1 and not args.no_configs1 and not album and not args.mixed:
        pcm1, descs1, bytes1 = make_batch([(0x5EED0000, RATE, 2, FRAMES_10MIN)])

        def step1():
            an.enqueue_device(descs1, 1, pcm1.data_ptr(), bytes1, album=False)

        d1, s1, l1, sp1 = timed(step1, args.configs1_steps, 20, FRAMES_10MIN)
        r1 = an.collect(1)
        configs1 = {"workload": "configs[1]: 1 track, 10 min synthetic 44.1 kHz stereo PCM resident in HBM "
                                "(211.7 MB: inside the 256 MiB Infinity Cache, re-read every step)",
                    "value": FRAMES_10MIN * args.configs1_steps / d1, "unit": "stereo samples/s",
                    "steps": args.configs1_steps, "ms_per_step": d1 / args.configs1_steps * 1e3,
                    "roofline": roofline_block(FRAMES_10MIN, s1, l1, sp1, "cfg1"),
                    "bound": "block-retire arithmetic (DESIGN.md section 7): 24 000 windows are too few lanes for multi-window segments, so "
                             "L = 735 (35.1 vector instructions per channel-sample against 28.3); 282 blocks of 256 lanes, three per CU, "
                             "181 us each at three waves per SIMD and 88 % issue -> 256 CUs retire one launch per 66.5 us = 0.398 of HBM",
                    "result": {"loudness_db": r1[0].loudness_db, "gain_db": r1[0].gain_db, "peak": r1[0].peak,
                               "flags": r1[0].flags}}
        del pcm1

    # ---- informational leg on one GPU: from compressed MP3 files to album gain (north_star's "MP3 frame decode to
    # PCM" in front of the measured path).  Never `value`: it includes file reads, the host's frame walk and PCIe. ----
    mp3_leg = None
    if world == 1 and not args.no_mp3 and not album and not args.mixed:
        try:
            mp3_leg = mp3_end_to_end(an, args.mp3_files)
        except Exception as ex:  # noqa: BLE001
            mp3_leg = {"error": str(ex)}

    out = None
    if rank == 0:
        cpu = None
        parity = None
        if args.cpu_seconds > 0 and ntr > 0:
            from oracle import pyoracle as po

            # ---- parity: full histograms of the first tracks of this rank's batch against the oracle ----
            npar = min(args.parity_tracks, ntr)
            diff_bins = 0
            max_db = 0.0
            peaks_equal = True
            same_input = True
            want0 = None
            # in a mixed batch the compared tracks are spread over it, so that every launch group is covered
            if args.mixed:
                pick = {(k * ntr) // npar for k in range(npar)}
                pick |= {next((i for i in range(ntr) if specs[i][2] == 1), 0), next((i for i in range(ntr) if specs[i][0] & HOT), 0)}
                pick |= {next((i for i in range(ntr - 1, -1, -1) if specs[i][2] == 1), 0)}  # a mono track of the second rate
                pick = sorted(pick)
            else:
                pick = list(range(npar))
            for n_done, t in enumerate(pick):
                seed_t, rate_t, ch_t, fr_t = specs[t]
                l = po.synth_f32(seed_t, 0, rate_t, fr_t)
                r = po.synth_f32(seed_t, 1, rate_t, fr_t) if ch_t == 2 else None
                if n_done == 0:
                    off0 = descs[t].offset_bytes // 4
                    host = pcm[off0:off0 + ch_t * fr_t].cpu().numpy().reshape(ch_t, fr_t)
                    same_input = bool(np.array_equal(host[0], l) and (r is None or np.array_equal(host[1], r)))
                want, want_hist = po.analyze_pcm(l, r, rate_t)
                if n_done == 0:
                    want0, l0, r0, rate0, fr0, t0_ = want, l, r, rate_t, fr_t, t
                diff_bins += int(np.count_nonzero(hists[t] != want_hist))
                max_db = max(max_db, abs(res[t].loudness_db - want["loudness_db"]))
                peaks_equal = peaks_equal and res[t].peak == want["peak"]
            parity = {"tracks_compared": len(pick), "same_input_bits": same_input, "differing_histogram_bins": diff_bins,
                      "max_abs_db_delta": max_db, "peaks_equal": peaks_equal,
                      "loudness_db_gpu": res[t0_].loudness_db, "loudness_db_oracle": want0["loudness_db"],
                      "db_delta": res[t0_].loudness_db - want0["loudness_db"],
                      "tracks_flagged_imprecise": n_imprecise, "tracks_flagged_nonfinite": n_nonfinite,
                      "tracks_in_batch": ntr,
                      "note": "async enqueue/collect path: a flagged track may have windows off by <= 3 bins (0.03 dB); "
                              "the synchronous entry points and rg_collect_exact re-run flagged tracks on the order-faithful kernel"}
            # ---- CPU baseline: a bounded sample of the same workload (one of its tracks, repeated) ----
            cpu = cpu_baseline_leg(po, l0, r0, rate0, fr0, f"track {t0_} of the batch", args.cpu_seconds)
        if args.mixed:
            wl = (f"configs[4], PCM side: {ntr} synthetic {args.minutes:g}-min tracks per GPU, half 44.1 kHz and half 48 kHz, every 10th mono, "
                  f"every 20th clipped at full scale (peak >= 1.0), {pcm_bytes / 1e9:.1f} GB planar f32 resident in HBM; "
                  f"{groups} launch groups per step (AAC/M4A decode itself is not part of this line)")
        elif world == 1 and not album:
            wl = (f"configs[2]: batch of {ntr} synthetic {frames / RATE / 60:g}-min 44.1 kHz stereo tracks, per-track gain (-r), "
                  f"{pcm_bytes / 1e9:.1f} GB planar f32 resident in HBM" if tag == "cfg2"
                  else f"track mode: {ntr} x {frames / RATE / 60:g}-min 44.1 kHz stereo track(s), {pcm_bytes / 1e9:.2f} GB resident in HBM")
        else:
            wl = (f"configs[3] shape: album mode over {total_tracks} synthetic {frames / RATE / 60:g}-min 44.1 kHz stereo tracks, "
                  f"{ntr} per GPU on {world} GPU(s), RCCL exchange of the album histogram")
        out = {
            "metric": "stereo PCM samples/s through IIR+RMS+histogram",
            "value": value,
            "unit": "stereo samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": wl,
                "tracks_per_gpu": ntr, "total_tracks": total_tracks, "frames_per_track": frames, "sample_rate": RATE,
                "mode": "album (-a): RCCL all-gather of the per-rank [12000-bin histogram | peak] packs + device fold" if album else "track (-r)",
                "exchange": exchange,
            },
            "exchange": (exchange_block(an, "rccl (library communicator, ncclCommInitRank, bootstrapped over torch.distributed)" if exchange.startswith("rccl")
                                        else "torch.distributed all_gather_into_tensor + device fold (library communicator unavailable)")
                         if album else None),
            "roofline": roof,
            "one_shot": one_shot,
            "cpu_baseline": cpu,
            "parity": parity,
            "configs1": configs1,
            "mp3_end_to_end": mp3_leg,
            "result": {"loudness_db": res[0].loudness_db if ntr else None, "gain_db": res[0].gain_db if ntr else None,
                       "peak": res[0].peak if ntr else None,
                       "album_loudness_db": alb.album_loudness_db if alb else None,
                       "album_peak": alb.album_peak if alb else None,
                       "tracks_flagged_imprecise": n_imprecise},
        }
        if rehearsal:
            out["rehearsal"] = ("NOT A MEASUREMENT: ranks share devices, torch.distributed over gloo, the library communicator over "
                                "the tests' stand-in transport (RG_BENCH_REHEARSAL=1)")
        print(json.dumps(out), flush=True)
    an.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
