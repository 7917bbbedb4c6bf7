#!/usr/bin/env python3
"""Headline benchmark: BLS12-381 G1 MSM at n = 2^26 (pf = 1) on MI355X, plus the 2^27 NTT latency.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one complete MSM over the 2^26 synthetic (scalar, point) pairs, inputs already resident
in HBM, through the reference's call sequence (initialize -> start_process -> set_data ->
wait_result -> result) on the C ABI.  Like the reference device's task queue, two tasks are kept in
flight; every one of the K timed MSMs is submitted and collected inside the timed region.  With
N > 1 the SAME 2^26 job is sharded over the N GPUs of one node (strong scaling) as
blz_msm_shard_layout_ex cuts it: element chunks x ranges of the scalars' bits (at 2^26 two and four
ranks split the bits of all elements, eight take 64-bit ranges of half of the elements each;
BLAZE_SHARD=elements forces contiguous element chunks, timed beside the headline as
`alt_layout_elements`).  Each rank runs its shard, one all-gather of the 144-byte partials (RCCL),
every rank adds them in rank order.  value = MSMs per second, whole job.

N > 1 cannot hang: the timed loop exchanges the partials through torch.distributed (the process group
the launcher's rendezvous already brought up); the exchange inside the library (blz_msm_comm_init /
blz_msm_all_gather_combine: a second RCCL communicator on the handle's own stream) is tried AFTER the
headline is measured and reported as the extra key `exchange_native`; every bring-up and every phase
runs under a deadline, and a rank whose deadline expires prints what it was waiting for and exits
non-zero (an exit, never a re-exec) - unless the headline is already measured and checked: a deadline
that expires in one of the EXTRA legs prints the line without the remaining extras (`extras_aborted`
names the leg) and exits 0.

With --gpus N > 1 and no WORLD_SIZE in the environment the script starts the N ranks itself (a child
`torch.distributed.run`, before this process touches the GPU) and relays their output.

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` for the dominant
kernel (k_accumulate) and `cpu_baseline` (the CPU oracle's Pippenger on a bounded sample).  The last
timed result is CHECKED before anything is printed (`result_check`): the synthetic points are
P_i = (i + 1) G, so the MSM must equal (sum_i s_i (i + 1) mod r) G; a mismatch raises."""
import argparse
import ctypes as C
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import bench_extras as extras  # noqa: E402  (the legs behind the headline; stdlib imports only at module level)
from bench_extras import SclkSampler, host_threads  # noqa: E402,F401

LOG_N = int(os.environ.get("BLAZE_BENCH_LOGN", "26"))
CURVE = "BLS381"
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MSM_BYTES_PER_ELEM = 128  # 32 B scalar + 96 B point
NTT_LOG = int(os.environ.get("BLAZE_BENCH_NTT_LOGN", "27"))


def spawn_ranks(n_ranks: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as children of a fresh
    torch.distributed.run (this process has not initialised the GPU and never will) and wait."""
    import socket
    import subprocess

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


class Watchdog:
    """A deadline around phases that can block for ever on a peer that died (rendezvous, collectives, a wedged
    kernel).  On expiry the rank says what it was waiting for and exits with status 3: the launcher then tears the
    other ranks down - unless the headline is already in hand (`fallback`).  os._exit, never an exec: this process has
    initialised the GPU."""

    def __init__(self, rank):
        self.rank, self.timer = rank, None
        # once the headline has been measured and checked: a callable that returns the line as far as it has got (rank 0) or
        # None (other ranks).  A deadline that expires in one of the EXTRA legs then costs the extras, not the headline: the
        # rank prints what it has - with `extras_aborted` saying which leg hung - and exits 0
        self.fallback = None

    def arm(self, seconds, what):
        import threading

        self.disarm()

        def fire():
            if self.fallback is not None:
                print(f"[bench rank {self.rank}] DEADLINE: {what} did not complete within {seconds} s; the headline is printed without the "
                      f"remaining extras", file=sys.stderr, flush=True)
                try:
                    line = self.fallback(f"{what}: not complete within {seconds} s")
                    if line is not None:
                        print(json.dumps(line), flush=True)
                finally:
                    os._exit(0)
            print(f"[bench rank {self.rank}] DEADLINE: {what} did not complete within {seconds} s; exiting 3", file=sys.stderr, flush=True)
            os._exit(3)

        self.timer = threading.Timer(seconds, fire)
        self.timer.daemon = True
        self.timer.start()

    def disarm(self):
        if self.timer is not None:
            self.timer.cancel()
            self.timer = None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-ntt", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="skip the result check (profiling runs only)")
    ap.add_argument("--no-extras", action="store_true", help="skip the reference-semantics legs (hbm_flow, config2_dma)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher and the flag disagree")
    dist = None
    wd = Watchdog(rank)
    # BLAZE_BENCH_FORCE_EXCHANGE=1: a one-rank job still goes through the process group and both exchanges (what the
    # single-GPU box can exercise of the N > 1 path: torch imported first, torch's NCCL process group alive, then the
    # library's own communicator next to it)
    force_exchange = os.environ.get("BLAZE_BENCH_FORCE_EXCHANGE") == "1"
    multi = world > 1 or force_exchange
    if multi:
        import datetime

        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            if world > 1:   # every rank picking its own port would rendezvous nowhere, until the watchdog fires
                raise SystemExit("bench: WORLD_SIZE > 1 without MASTER_PORT - start the ranks with torch.distributed.run "
                                 "(or `python bench.py --gpus N`, which does)")
            import socket

            with socket.socket() as so:   # the one-rank BLAZE_BENCH_FORCE_EXCHANGE path
                so.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(so.getsockname()[1])
        # "nccl" is RCCL on ROCm (xGMI between the GPUs of the node).  BLAZE_BENCH_BACKEND=gloo with
        # BLAZE_BENCH_ONE_GPU=1 lets the sharded path be exercised by several ranks on a 1-GPU box.
        backend = os.environ.get("BLAZE_BENCH_BACKEND", "nccl")
        if os.environ.get("BLAZE_BENCH_ONE_GPU") == "1":
            local_rank = 0
        ndev = torch.cuda.device_count()
        if ndev and local_rank >= ndev:
            # A launcher that narrows every rank's visibility to its own GPU(s) (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES set per
            # rank) leaves LOCAL_RANK pointing past the devices the rank can see; a plain mis-launch - more ranks than GPUs -
            # looks the same and must not be folded onto shared GPUs (nccl: "duplicate GPU" or a hang; a scaling number from
            # shared devices otherwise).
            narrowed = any(os.environ.get(v) for v in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"))
            if not narrowed:
                raise SystemExit(f"bench: LOCAL_RANK {local_rank} but only {ndev} device(s) visible and no *_VISIBLE_DEVICES narrowing: "
                                 f"{world} ranks need {world} GPUs (BLAZE_BENCH_ONE_GPU=1 BLAZE_BENCH_BACKEND=gloo runs them on one, for tests)")
            local_rank %= ndev
        torch.cuda.set_device(local_rank)
        t_pg = int(os.environ.get("BLAZE_BENCH_PG_TIMEOUT_S", "180"))
        wd.arm(t_pg + 30, "process-group rendezvous + first collective")
        dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=t_pg))
        probe = torch.ones(1, dtype=torch.int32, device=torch.device("cuda", local_rank) if backend == "nccl" else "cpu")
        dist.all_reduce(probe)               # the communicator is built lazily: build it here, under the deadline
        assert int(probe.item()) == world
        wd.disarm()
    dev = local_rank
    tdev = torch.device("cuda", dev)
    gather_dev = tdev if (not multi or dist.get_backend() == "nccl") else None

    import blaze_amd
    from blaze_amd import DeviceBuffer
    from blaze_amd._lib import check
    from blaze_amd.driver_client import DriverClient
    from blaze_amd.ingo_msm import Curve, MSMClient, MSMInit, MSMInput, MSMParams, PointMemoryType
    from blaze_amd.multi_gpu import SHARD_SCALARS_FROM_HOST, shard_layout_ex, sharded_msm

    L = blaze_amd.lib()
    # The interpreter's cyclic garbage collector: with torch imported the process holds millions of long-lived objects, and a
    # full collection - triggered by an allocation count, i.e. at some arbitrary get_api() of some timed loop - takes 0.6 - 0.7 s
    # (it showed up as ONE 700 ms interval among the 17 ms ones of the config 4 leg).  Everything alive now is moved out of the
    # collector's sight, and the timed loops run with it switched off.
    import gc

    gc.collect()
    gc.freeze()
    n = 1 << LOG_N
    # rank's shard: an element chunk x a range of the scalars' bits (blz_msm_shard_layout: the library picks the mix by the
    # window planner's cost - 2 and 4 ranks split the bits of all 2^26 elements, 8 ranks take 64-bit ranges of half of the
    # elements each; BLAZE_SHARD=elements forces the plain element split).  Partial results add up either way.
    # (blz_msm_shard_layout_ex prices the flow: the timed loop's scalars are resident, so nothing crosses a rank's link)
    lay = shard_layout_ex(Curve[CURVE], n, rank, world, 0)
    lo, n_loc = lay["first"], lay["count"]
    ranged = (lay["bit_lo"], lay["bit_hi"]) != (0, 256)
    cid = int(Curve[CURVE])

    # ---- synthetic inputs, generated on the device: P_i = (i+1) G, scalars uniform-ish in [0, r)
    d_pts = DeviceBuffer(dev, n_loc * 96)
    d_sc = DeviceBuffer(dev, n_loc * 32)
    check(blaze_amd.aux().blz_synth_points(dev, cid, d_pts.ptr, n_loc, 1, lo))
    check(blaze_amd.aux().blz_synth_scalars_at(dev, cid, d_sc.ptr, n_loc, 0xB1A2E, lo))  # same global set for every N

    # Points live in the device arena, as in the reference's HBM flow (tests/integration_msm_hbm.rs:
    # load_data_to_hbm once, then scalars-only set_data with hbm_point_addr): the bases of a prover are
    # fixed, the scalars change per MSM.  BLAZE_BENCH_MODE=dma streams points + scalars every step instead.
    hbm_mode = os.environ.get("BLAZE_BENCH_MODE", "hbm") == "hbm"
    if hbm_mode:
        L.blz_arena_release(dev)
        client = MSMClient(MSMInit(PointMemoryType.HBM, False, Curve[CURVE]), DriverClient(dev))
        client.load_data_to_hbm(d_pts, 0, 0)
        params = MSMParams(n_loc, (0, 0))
        step_points = None
    else:
        client = MSMClient(MSMInit(PointMemoryType.DMA, False, Curve[CURVE]), DriverClient(dev))
        params = MSMParams(n_loc, None)
        step_points = d_pts
    if ranged:
        client.set_scalar_range(lay["bit_lo"], lay["bit_hi"])

    # The device has a task queue and a result queue (src/ingo_msm/msm_hw_code.rs:19-25): QUEUE tasks
    # are kept in flight, so the few-lane tail of one MSM (upper bucket-reduce levels, Horner, inversion)
    # overlaps the sort + accumulation of the next.  BLAZE_BENCH_QUEUE=1 runs strictly one at a time.
    queue = max(1, min(2, int(os.environ.get("BLAZE_BENCH_QUEUE", "2"))))

    # N > 1: the timed loop exchanges the partials over the process group that already exists (one all-gather of
    # 144 bytes per MSM, off the critical path: two tasks are in flight) + the device combine.  The library's own
    # exchange (a second RCCL communicator) is measured after the headline, as `exchange_native`.
    exchange = "torch.distributed all_gather + combine_partials (device)" if multi else "none"
    last_partial = [None]

    def submit():
        client.initialize(params)
        client.start_process()
        client.set_data(MSMInput(step_points, d_sc, params))

    def collect_local():
        client.wait_result()
        part = client.result().result
        return part, client.get_api()  # HIP-event timers recorded on the streams the kernels run on

    def exchange_partial(loc):
        part, api = loc
        if multi:
            last_partial[0] = part
            part = sharded_msm(part, client.combine_partials, dist, gather_dev)
        return part, api

    def run_steps(k):
        # a collected result frees a queue slot: the next task is handed to the device BEFORE the collected partial is
        # exchanged (its digit sort has to be enqueued early enough to finish underneath the accumulation in flight)
        out, pending, submitted = [], 0, 0
        while submitted < k or pending:
            if pending >= queue or submitted >= k:
                loc = collect_local()
                pending -= 1
                if submitted < k:
                    submit()
                    submitted += 1
                    pending += 1
                out.append(exchange_partial(loc))
            else:
                submit()
                submitted += 1
                pending += 1
        return out

    def fence():
        if multi:
            dist.barrier()
        torch.cuda.synchronize(tdev)

    t_run = int(os.environ.get("BLAZE_BENCH_RUN_TIMEOUT_S", "900"))
    wd.arm(t_run, f"warm-up + {args.steps} timed steps")
    run_steps(args.warmup)
    fence()
    sclk = SclkSampler(torch, dev)
    sclk.start()
    gc.disable()
    t0 = time.perf_counter()
    done = run_steps(args.steps)
    fence()
    dt = time.perf_counter() - t0
    gc.enable()
    sclk_rec = sclk.stop()
    res, api = done[-1]
    accum_ms = [a["accumulate_kernel_ms"] for _, a in done]
    total_ms = [a["total_ms"] for _, a in done]
    if multi:
        t = torch.tensor([dt], dtype=torch.float64, device=gather_dev if gather_dev is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    wd.disarm()
    # the multiply-add issue rate of THIS chip at its clocks of this moment (50 ms of independent v_mad_u64_u32
    # chains on every SIMD, right behind the timed steps): what roofline.integer_issue is priced against
    calib = None
    try:
        cal = (C.c_double * 4)()
        check(blaze_amd.aux().blz_calib_mad_rate(dev, 50, cal))
        calib = {"mad_lane_ops_per_s": cal[0], "kernel_ms": round(cal[1], 3), "nominal_clock_mhz": cal[2]}
    except Exception as e:   # noqa: BLE001 - a measurement aid, never fatal
        print(f"[bench rank {rank}] mad-rate calibration failed: {e}", file=sys.stderr, flush=True)
    ms_per_step = dt / args.steps * 1e3
    value = args.steps / dt

    # ---- the number is only worth printing if the timed result is right: linearity over P_i = (i + 1) G
    # (oracle = test infrastructure, used here as the checker only, outside the timed region)
    check_rec = None
    if not args.no_check:
        if rank == 0:
            import oracle

            if world == 1:
                sc_all = d_sc.download()
            else:   # the other ranks' scalars: same generator, same global indices
                d_all = DeviceBuffer(dev, n * 32)
                check(blaze_amd.aux().blz_synth_scalars_at(dev, cid, d_all.ptr, n, 0xB1A2E, 0))
                sc_all = d_all.download()
                d_all.free()
            t1 = time.perf_counter()
            k = oracle.index_weighted_sum(CURVE, sc_all, n, 0, threads=min(64, host_threads()))
            exp = oracle.result_from_affine(CURVE, oracle.generator_mul(CURVE, k))
            del sc_all
            if res != exp:
                raise SystemExit(f"bench: the timed MSM result is WRONG (got {res.hex()[:32]}..., expected {exp.hex()[:32]}...)")
            check_rec = {"ok": True, "method": "result == (sum_i s_i (i+1) mod r) G over all 2^%d scalars, CPU oracle, %.1f s"
                         % (LOG_N, time.perf_counter() - t1)}
        if multi:
            wd.arm(300, "barrier after the result check")
            dist.barrier()
            wd.disarm()

    native = None   # (the library's own exchange is measured below, once the headline line is in hand)

    # ---- roofline of the dominant kernel (k_accumulate: one launch covers this rank's whole shard)
    acc_avg_ms = statistics.mean(accum_ms)
    algo_bytes = n_loc * MSM_BYTES_PER_ELEM
    achieved = algo_bytes / (acc_avg_ms * 1e-3) / 1e9
    traffic, traffic_dropped = None, None
    tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tf):
        try:
            rec = json.load(open(tf)).get(f"k_accumulate_2e{LOG_N}_{CURVE}")
            # the PMC passes cannot run inside this process; the record is only quoted while the kernel it was
            # taken on is the kernel being timed (same duration within 10 %), else it is stale and dropped
            if rec and world == 1:
                rec_ms = rec.get("kernel_ms_at_measurement", acc_avg_ms)
                if abs(acc_avg_ms - rec_ms) <= 0.10 * acc_avg_ms:
                    traffic = rec["hbm_bytes_per_launch"]
                else:
                    traffic_dropped = f"kernel {acc_avg_ms:.1f} ms here vs {rec_ms:.1f} ms when the counters were read (> 10 %)"
            elif world == 1:
                traffic_dropped = "no counter record for this workload in profiles/pmc_traffic.json"
            else:
                traffic_dropped = "counters were read on the 1-GPU workload only"
        except Exception as e:   # noqa: BLE001
            traffic, traffic_dropped = None, f"profiles/pmc_traffic.json unreadable: {e}"
    else:
        traffic_dropped = "profiles/pmc_traffic.json missing"
    roofline = {"bound": "hbm", "kernel": "k_accumulate", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                "algorithmic_bytes_per_launch": algo_bytes, "kernel_ms": round(acc_avg_ms, 3),
                "pipeline_ms": round(statistics.mean(total_ms), 3)}
    if traffic is None:
        roofline["traffic_dropped"] = traffic_dropped
    # The resource this kernel actually saturates, beside the prescribed HBM figure: 32-bit integer multiply issue.
    # One bucket addition (mixed XYZZ add on 14 x 28-bit limbs) is 3542 v_mad_u64_u32; a launch does one addition per
    # non-zero window digit (255-bit scalars in 12 occupied windows at this size).  The peak is that instruction's
    # rate on THIS chip, measured by the calibration kernel right behind the timed steps (blz_calib_mad_rate); the
    # constant of the builder's boxes (3.1e13, profiles/r02_mul_variants.txt) is kept beside it for comparison.
    if CURVE in ("BLS381", "BLS377") and LOG_N >= 24:
        occupied = -(-(255 if CURVE == "BLS381" else 253) // int(api["window_bits"]))
        if ranged:   # every window of a 64 / 128-bit range holds digits (the top one fewer: counted whole, an upper bound)
            occupied = -(-(min(lay["bit_hi"], 255 if CURVE == "BLS381" else 253) - lay["bit_lo"]) // int(api["window_bits"]))
        mads = n_loc * occupied * 3542
        peak = calib["mad_lane_ops_per_s"] if calib else 3.1e13
        roofline["integer_issue"] = {"unit": "v_mad_u64_u32 lane-ops/s", "achieved": round(mads / (acc_avg_ms * 1e-3), 0),
                                     "peak": round(peak, 0), "frac": round(mads / (acc_avg_ms * 1e-3) / peak, 4),
                                     "peak_source": "calibration kernel on this device, this run (50 ms of independent multiply-add chains, "
                                                    "4 waves per SIMD)" if calib else "constant measured on another box",
                                     "peak_reference_boxes": 3.1e13, "multiply_adds_per_launch": mads,
                                     "multiply_adds_source": "3542 per bucket addition = the v_mad_u64_u32 count of k_accumulate<Fq_BLS381>'s loop in the "
                                                             "shipped gfx950 code object (tests/test_isa_counts.py disassembles it), x one addition per "
                                                             "element and occupied window"}
    clock = {"nominal_mhz": calib["nominal_clock_mhz"] if calib else None, "sclk_mhz_timed_steps": sclk_rec,
             "mad_calibration": calib}

    # every extra key exists from here on (None until its leg has run): the line can be printed at any point after this
    table_rec = alt_rec = hbm_flow = cfg2 = cfg3 = cfg4 = lone_small = ntt = cpu = cpu_ref = None

    def make_line(extras_aborted=None):
        if rank != 0:
            return None
        # The driver's record keeps the depth-1 SCALARS of roofline / config / cpu_baseline and drops nested objects and extra
        # top-level keys (BENCH_r05.parsed): the second half of BASELINE.json's metric ("+ 2^27 NTT ms") and the two
        # integer-issue fractions are therefore repeated here as scalars, next to the objects that explain them.
        ii = roofline.get("integer_issue")
        roofline["integer_issue_frac"] = ii["frac"] if ii else None
        roofline["ntt_2e27_ms"] = roofline["ntt_frac"] = roofline["ntt_integer_issue_frac"] = roofline["ntt_traffic"] = None
        if ntt is not None and ntt.get("log_size") == 27:
            roofline["ntt_2e27_ms"] = ntt["kernel_ms"]
            roofline["ntt_frac"] = ntt["roofline"]["frac"]
            roofline["ntt_traffic"] = ntt["roofline"].get("traffic")
            nii = ntt["roofline"].get("integer_issue")
            roofline["ntt_integer_issue_frac"] = nii["frac"] if nii else None
            roofline["ntt_host_loop_ms"] = ntt.get("host_loop_ms")
            roofline["ntt_host_loop_pinned_ms"] = ntt.get("host_loop_pinned_ms")
        line = {
            "metric": f"BLS12-381 MSM/s at 2^{LOG_N}", "value": round(value, 4), "unit": "MSM/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "u32 registers holding 28-bit limbs (381-bit Fq, Montgomery)", "data": "synthetic",
            "config": {"workload": f"2^{LOG_N} BLS12-381 G1 MSM, pf=1, scalars+points resident in HBM"
                                   + (" (points in the device arena, scalars-only set_data)" if hbm_mode else " (DMA-mode set_data with device pointers)"),
                       "elements": n, "elements_per_gpu": n_loc, "parallelism": f"shard{world}" if world > 1 else "single",
                       "shard_rank0": lay, "exchange": exchange, "tasks_in_flight": queue,
                       "window_bits": int(api["window_bits"]), "windows": int(api["windows"]),
                       "sort_hidden_under_previous_accumulation": bool(api.get("sort_hidden", 0))},
            "roofline": roofline, "cpu_baseline": cpu, "cpu_baseline_ref_semantics": cpu_ref, "result_check": check_rec,
            "ntt_2e27": ntt, "clock": clock, "exchange_native": native, "window_table": table_rec, "alt_layout_elements": alt_rec, "hbm_flow": hbm_flow, "config2_dma": cfg2,
            "config3_bn254_pf8": cfg3, "config4_rank_task": cfg4, "lone_small_msm": lone_small,
            "phases_ms": {k: round(v, 3) for k, v in api.items() if k.endswith("_ms")},
        }
        if os.environ.get("BLAZE_BENCH_EMIT_RESULT") == "1":
            line["result_hex"] = res.hex()
        if extras_aborted is not None:
            line["extras_aborted"] = extras_aborted
        return line

    wd.fallback = make_line   # (from here on a hung extra leg costs the extras only)
    if os.environ.get("BLAZE_BENCH_TEST_STALL_EXTRAS") == "1":   # (tests: an extra leg that never comes back)
        wd.arm(2, "test stall in the extras")
        time.sleep(3600)

    # ---- everything below is an EXTRA leg (bench_extras.py): measured after the headline line is in hand, reported under its own key,
    # never the headline value; a deadline that expires in one of them costs the remaining extras only (wd.fallback)
    import types

    ctx = types.SimpleNamespace(args=args, rank=rank, world=world, multi=multi, dist=dist, torch=torch, tdev=tdev, dev=dev, gather_dev=gather_dev, wd=wd,
                                client=client, params=params, d_sc=d_sc, d_pts=d_pts, queue=queue, lay=lay, ranged=ranged, n=n, n_loc=n_loc, cid=cid, res=res,
                                last_partial=last_partial, hbm_mode=hbm_mode, calib=calib, L=L, tf=tf, CURVE=CURVE, LOG_N=LOG_N, NTT_LOG=NTT_LOG,
                                HBM_PEAK_GBS=HBM_PEAK_GBS, ntt=None)
    native = extras.native_exchange(ctx)
    table_rec = extras.window_table_leg(ctx)
    alt_rec, hbm_flow = extras.multi_rank_legs(ctx)
    hbm_flow_1, cfg2 = extras.reference_flows(ctx)
    if hbm_flow_1 is not None:
        hbm_flow = hbm_flow_1
    cfg3, cfg4, lone_small = extras.configs_3_4_and_small(ctx)
    ntt = ctx.ntt = extras.ntt_leg(ctx)
    cpu, cpu_ref = extras.cpu_baselines(ctx)

    if rank == 0:
        line = make_line()
        print(json.dumps(line), flush=True)
    if multi:
        wd.arm(120, "process-group teardown")
        dist.destroy_process_group()
        wd.disarm()
        if native is not None and not native["ok"]:
            # a bring-up thread abandoned inside RCCL must not keep the interpreter from exiting
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(0)


if __name__ == "__main__":
    main()
