#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X: stereo PCM samples/s through the
IIR + 50 ms RMS + histogram path, with the dB delta against the CPU oracle.

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one pass of the hot path over one batch of synthetic PCM that is already resident
in HBM (generated there by the library's rg_synth kernel; include/rg_synth.h):

  N == 1   BASELINE configs[1]: one 10-minute 44.1 kHz stereo track (26 460 000 frames, 211.7 MB
           planar f32) -> per-track histogram, percentile, gain, peak.
  N  > 1   album mode, weak scaling: every rank owns one such 10-minute track per step; after the
           per-rank kernels the [12 000-bin album histogram | album peak] packs of all ranks are
           all-gathered over RCCL (one collective, on the stream of the batch) and folded on the
           device (sum / max), then every rank runs the album percentile.  One rank per GPU, launched
           by torch.distributed.run.

Timing: an untimed pre-roll (0.25 s worth of steps, so that clock and power have settled), W warm-up
steps, then exactly K steps between barrier + torch.cuda.synchronize() pairs; the time is the MAX
over ranks; value = frames processed by all ranks / that time.
Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (IIR+RMS+histogram),
measured with HIP events on the stream it is launched on; `cpu_baseline` is the CPU oracle
(a C restatement of the reference's sequential algorithm -- not the Rust binary, which cannot
be built in this image) on the same track, one thread.
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
ALGO_BYTES_PER_FRAME = 8           # 2 channels x f32, read once
HBM_PEAK_GBPS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


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


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--kernel", type=int, default=0, help="kernel variant (0 = library default)")
    ap.add_argument("--tracks-per-rank", type=int, default=1)
    ap.add_argument("--minutes", type=float, default=10.0, help="track length (default: BASELINE's 10 min)")
    ap.add_argument("--tm-segment", type=int, default=0, help="force variant 2 segment length (tuning)")
    ap.add_argument("--slots", type=int, default=0, help="pipeline slots of the library (0 = default)")
    ap.add_argument("--album", action="store_true", help="force the album path (collectives) even on one GPU")
    ap.add_argument("--cpu-reps", type=int, default=16, help="oracle repetitions for cpu_baseline (0 = skip)")
    ap.add_argument("--pre-roll", type=float, default=PRE_ROLL_SECONDS, help="untimed pre-roll before the warm-up steps, seconds of work")
    args = ap.parse_args()

    import numpy as np
    import torch

    import mp3rgain_amd as rg
    from mp3rgain_amd import _capi
    from mp3rgain_amd import album as album_mod

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU path")
    torch.cuda.set_device(local_rank)
    frames = int(round(args.minutes * 60 * RATE))
    ntr = args.tracks_per_rank
    # The context first: its pipeline streams should each get a hardware queue of their own (the runtime has 4
    # per process and deals them out as streams are created; torch.distributed / RCCL create several more).
    an = rg.Analyzer(local_rank)
    dist = None
    if world > 1 or (args.album and "RANK" in os.environ):
        import torch.distributed as dist  # noqa: PLC0415

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    if args.kernel:
        an.set_kernel(args.kernel)
    if args.tm_segment:
        an.set_tuning(1, args.tm_segment)
    if args.slots:
        an.set_tuning(3, args.slots)
    # No caller stream is attached: every batch, its album tail and the collective in between run on the
    # context's own pipeline streams (rg_batch_stream), which costs no cross-stream event per step.

    # ---- synthetic PCM straight into HBM ---------------------------------------------------
    pcm = torch.empty((ntr, 2, frames), dtype=torch.float32, device="cuda")
    descs = (_capi.TrackDesc * ntr)()
    seeds = [0x5EED0000 + rank * ntr + t for t in range(ntr)]
    for t in range(ntr):
        for c in range(2):
            an.synth_fill_device(pcm[t, c].data_ptr(), seeds[t], c, RATE, 0, frames)
        descs[t].offset_bytes = t * 2 * frames * 4
        descs[t].frames = frames
        descs[t].sample_rate = RATE
        descs[t].channels = 2
        descs[t].format = _capi.FMT_F32_PLANAR
    pcm_bytes = pcm.numel() * 4
    torch.cuda.synchronize()

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

    # Pre-roll, untimed and before the W warm-up steps: the shader clock and the power controller take some
    # milliseconds to settle once the kernel starts running (the first ~10 ms run 10-15 % slower), and a short
    # --warmup would otherwise put that ramp into the timed region.  The timed region below is exactly K steps.
    # The step count comes from the workload size, not from a clock: every rank must issue the same collectives.
    pre_steps = max(4, min(8192, int(args.pre_roll / (frames * ntr / 3.5e11))))
    for i in range(pre_steps):
        step()
        if i % 64 == 63:
            torch.cuda.synchronize()  # keep the host from running seconds ahead of the device
    for _ in range(args.warmup):
        step()
    fence()
    an.timing_enable(True)
    an.timing_read(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    k1_ms_sum, k1_launches, k1_span_ms = an.timing_read(reset=True)
    an.timing_enable(False)

    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # ---- results of the last step + parity against the oracle (outside the timed region) ----
    res = an.collect(ntr)
    alb = an.album_finish() if album else None

    total_frames = frames * ntr * world * args.steps
    value = total_frames / dt
    # Dominant kernel (rg_tm_main_kernel), HIP events on the streams it is launched on.  Consecutive
    # batches run in separate pipeline slots, so several launches are in flight at once:
    #   kernel_ms    average duration of one launch (what rocprofv3 --stats reports as the average)
    #   concurrency  sum of the launch durations / span from the first start to the last end
    #   achieved     algorithmic bytes per launch / (kernel_ms / concurrency), i.e. the bytes all launches
    #                consumed divided by the time during which the kernel was running
    k1_ms = k1_ms_sum / max(1, k1_launches)
    concurrency = k1_ms_sum / k1_span_ms if k1_span_ms > 0 else 1.0
    algo_bytes = ALGO_BYTES_PER_FRAME * frames * ntr
    achieved = algo_bytes * k1_launches / (k1_span_ms * 1e-3) / 1e9 if k1_span_ms > 0 else 0.0
    traffic = None
    try:  # HBM bytes per launch from the committed PMC passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE), same workload only
        pm = json.loads((ROOT / "profiles" / "pmc_traffic.json").read_text())
        if pm.get("frames_per_launch") == frames * ntr:
            traffic = pm["hbm_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass

    out = None
    if rank == 0:
        cpu = None
        parity = None
        if args.cpu_reps > 0:
            from oracle import pyoracle as po

            l = po.synth_f32(seeds[0], 0, RATE, frames)
            r = po.synth_f32(seeds[0], 1, RATE, frames)
            host = pcm[0].cpu().numpy()
            same_input = bool(np.array_equal(host[0], l) and np.array_equal(host[1], r))
            want, _ = po.analyze_pcm(l, r, RATE)  # warm + parity
            c0 = time.perf_counter()
            for _ in range(args.cpu_reps):
                po.analyze_pcm(l, r, RATE)
            cdt = time.perf_counter() - c0
            # the reference is single-threaded; with tracks as the parallel unit the same code on every
            # host core is the most a CPU deployment could do, so that figure is reported next to it
            from concurrent.futures import ThreadPoolExecutor

            nthr = _usable_cores()
            per_thread = 3
            m0 = time.perf_counter()
            with ThreadPoolExecutor(nthr) as pool:  # ctypes releases the GIL inside the C call
                list(pool.map(lambda _: [po.analyze_pcm(l, r, RATE) for _ in range(per_thread)], range(nthr)))
            mdt = time.perf_counter() - m0
            cpu = {"value": frames * args.cpu_reps / cdt, "unit": "stereo samples/s", "cores": 1, "kind": "port",
                   "sample": f"the bench track ({frames} stereo frames @44.1 kHz) x{args.cpu_reps}, "
                             f"oracle/rg_oracle.c (C restatement of replaygain.rs, not the Rust binary), 1 thread, "
                             f"host has {os.cpu_count()} cores, {_usable_cores()} usable by this process",
                   "all_cores": {"value": frames * per_thread * nthr / mdt, "cores": nthr,
                                 "sample": f"the same track x{per_thread} on each of {nthr} threads (tracks as the parallel unit)"}}
            parity = {"same_input_bits": same_input, "loudness_db_gpu": res[0].loudness_db,
                      "loudness_db_oracle": want["loudness_db"],
                      "db_delta": res[0].loudness_db - want["loudness_db"], "peak_equal": res[0].peak == want["peak"]}
        out = {
            "metric": "stereo PCM samples/s through IIR+RMS+histogram",
            "value": value,
            "unit": "stereo samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": ("configs[1]: 1 track, 10 min synthetic 44.1 kHz stereo PCM resident in HBM" if world == 1 and ntr == 1 and frames == FRAMES_10MIN
                             else f"album mode: {ntr} x {frames / RATE / 60:.1f}-min 44.1 kHz stereo track(s) per GPU, {world} GPU(s)"),
                "tracks_per_gpu": ntr, "frames_per_track": frames, "sample_rate": RATE,
                "mode": "album (-a): RCCL all-gather of the per-rank [12000-bin histogram | peak] packs + device fold" if album else "track (-r)",
                "exchange": exchange,
            },
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel": "rg_tm_main_kernel", "kernel_ms": k1_ms, "kernel_launches": int(k1_launches),
                         "kernel_concurrency": concurrency, "kernel_span_ms": k1_span_ms,
                         "achieved_one_launch_alone": algo_bytes / (k1_ms * 1e-3) / 1e9 if k1_ms > 0 else 0.0,
                         "algorithmic_bytes_per_launch": algo_bytes,
                         "fp64_fma_tflops": 2.0 * 62.0 * frames * ntr * k1_launches / (k1_span_ms * 1e-3) / 1e12 if k1_span_ms > 0 else 0.0},
            "cpu_baseline": cpu,
            "parity": parity,
            "result": {"loudness_db": res[0].loudness_db, "gain_db": res[0].gain_db, "peak": res[0].peak,
                       "album_loudness_db": alb.album_loudness_db if alb else None},
        }
        print(json.dumps(out), flush=True)
    an.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
