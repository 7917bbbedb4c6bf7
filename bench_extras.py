"""The extra legs of bench.py - everything that is measured AFTER the headline line is in hand and never is the headline value: the
library's own RCCL exchange, the resident-base window table, the N > 1 element split and per-rank HBM flow, the reference's own
flows (hbm_flow, config2_dma), BASELINE configs 3 and 4 and lone small tasks, the 2^27 NTT, the CPU baselines.  Each leg takes the
context bench.py's main() built (a SimpleNamespace: the headline's client, inputs, layout, process group, watchdog) and returns its
record(s); a leg that fails reports {"error": ...} in its key and costs nothing else."""
import ctypes as C   # noqa: F401
import gc
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))


def host_threads() -> int:
    """Threads this process may really use: the affinity mask, capped by the cgroup CPU quota (a container on a
    256-thread host is often allowed a fraction of it; threads beyond the quota only add contention)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            pr = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / pr + 0.5)))
        except (OSError, ValueError):
            pass
    return n


class SclkSampler:
    """Shader clock of this rank's GPU while the timed steps run, read from the driver's sysfs table
    (pp_dpm_sclk marks the current level with '*'); None when the box does not expose it."""

    def __init__(self, torch, dev, period=0.05):
        import glob

        self.path, self.samples, self.stop_flag, self.thread, self.period = None, [], False, None, period
        try:
            pr = torch.cuda.get_device_properties(dev)
            cand = "/sys/bus/pci/devices/%04x:%02x:%02x.0/pp_dpm_sclk" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            if os.path.exists(cand):
                self.path = cand
        except Exception:
            pass
        if self.path is None:
            found = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
            if len(found) == 1:
                self.path = found[0]

    def _read(self):
        for ln in open(self.path).read().splitlines():
            if ln.rstrip().endswith("*"):
                return float(ln.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
        return None

    def start(self):
        import threading

        if self.path is None:
            return

        def loop():
            while not self.stop_flag:
                try:
                    v = self._read()
                    if v:
                        self.samples.append(v)
                except Exception:
                    return
                time.sleep(self.period)

        self.thread = threading.Thread(target=loop, daemon=True)
        self.thread.start()

    def stop(self):
        self.stop_flag = True
        if self.thread is not None:
            self.thread.join(timeout=1.0)
        if not self.samples:
            return None
        return {"min": min(self.samples), "mean": round(statistics.mean(self.samples), 1), "max": max(self.samples),
                "samples": len(self.samples), "source": f"pp_dpm_sclk (driver sysfs), sampled every {int(self.period * 1e3)} ms over the timed steps"}


def stream(ctx, cl, prm, pts_in, sc_in, k, on_set=None, on_done=None):
    """k tasks through `cl`, `queue` in flight (the headline's submission pattern); returns (seconds, results, apis)."""
    from blaze_amd.ingo_msm import MSMInput

    queue, torch, tdev = ctx.queue, ctx.torch, ctx.tdev
    outs, apis, pending = [], [], 0
    gc.disable()
    t_0 = time.perf_counter()
    for _ in range(k):
        cl.initialize(prm)
        cl.start_process()
        t_s = time.perf_counter()
        cl.set_data(MSMInput(pts_in, sc_in, prm))
        if on_set is not None:
            on_set((time.perf_counter() - t_s) * 1e3)
        pending += 1
        if pending >= queue:
            t_w = time.perf_counter()
            cl.wait_result()
            t_w1 = time.perf_counter()
            outs.append(cl.result().result)
            t_w2 = time.perf_counter()
            apis.append(cl.get_api())
            if os.environ.get("BENCH_DEBUG") and time.perf_counter() - t_w > 0.2:
                print(f"[bench debug] slow collect: wait_result {(t_w1 - t_w) * 1e3:.1f} ms, result {(t_w2 - t_w1) * 1e3:.1f} ms, get_api {(time.perf_counter() - t_w2) * 1e3:.1f} ms", file=sys.stderr, flush=True)
            if on_done is not None:
                on_done(time.perf_counter())
            pending -= 1
    while pending:
        cl.wait_result()
        outs.append(cl.result().result)
        apis.append(cl.get_api())
        if on_done is not None:
            on_done(time.perf_counter())
        pending -= 1
    torch.cuda.synchronize(tdev)
    dt_ = time.perf_counter() - t_0
    gc.enable()
    return dt_, outs, apis

def all_ranks(ctx, dt_local, err_local):
    """(max over ranks of a leg's time, error of any rank): every rank calls it once per leg, whatever happened to it"""
    multi, gather_dev, torch, dist = ctx.multi, ctx.gather_dev, ctx.torch, ctx.dist
    if not multi:
        return dt_local, err_local
    fdev2 = gather_dev if gather_dev is not None else "cpu"
    t = torch.tensor([dt_local if err_local is None else -1.0, 0.0 if err_local is None else 1.0], dtype=torch.float64, device=fdev2)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if float(t[1].item()) > 0:
        return -1.0, err_local or "the leg failed on another rank"
    return float(t[0].item()), None


def native_exchange(ctx):
    """blz_msm_all_gather_combine next to the torch.distributed exchange of the headline loop (N > 1): `exchange_native`"""
    args, rank, world, multi, dist, torch, tdev, dev, gather_dev, wd = (ctx.args, ctx.rank, ctx.world, ctx.multi, ctx.dist, ctx.torch, ctx.tdev, ctx.dev,
                                                                        ctx.gather_dev, ctx.wd)
    client, params, d_sc, d_pts, queue, lay, ranged, n, n_loc, cid = (ctx.client, ctx.params, ctx.d_sc, ctx.d_pts, ctx.queue, ctx.lay, ctx.ranged, ctx.n,
                                                                      ctx.n_loc, ctx.cid)
    res, last_partial, hbm_mode, calib, L, tf = ctx.res, ctx.last_partial, ctx.hbm_mode, ctx.calib, ctx.L, ctx.tf
    CURVE, LOG_N, NTT_LOG, HBM_PEAK_GBS = ctx.CURVE, ctx.LOG_N, ctx.NTT_LOG, ctx.HBM_PEAK_GBS
    import blaze_amd
    from blaze_amd import DeviceBuffer
    from blaze_amd._lib import check
    from blaze_amd.driver_client import DriverClient
    from blaze_amd.ingo_msm import Curve, MSMClient, MSMInit, MSMInput, MSMParams, PointMemoryType
    from blaze_amd.multi_gpu import SHARD_SCALARS_FROM_HOST, shard_layout_ex, sharded_msm
    native = None
    # ---- the library's own exchange, after the fact: RCCL resolved at run time inside
    # libblaze_hip, a communicator per handle, ncclAllGather on the handle's stream + k_combine_partials.  Nothing
    # here can hang the job: "is RCCL loadable" is agreed on first, rank 0 ALWAYS broadcasts (an id or None), the
    # bring-up has its own deadline inside the library (BLAZE_COMM_TIMEOUT_MS), the exchange is a bounded wait, and
    # the watchdog stands behind all of it.
    if multi and os.environ.get("BLAZE_BENCH_EXCHANGE", "native") == "native":
        wd.arm(240, "native RCCL exchange (bring-up + 5 exchanges)")
        fdev = gather_dev if gather_dev is not None else "cpu"
        err = None
        try:
            my_id = MSMClient.comm_unique_id()          # loads librccl through the library; every rank tries
            ok = 1
        except Exception as e:   # noqa: BLE001
            my_id, ok, err = None, 0, f"RCCL not loadable: {e}"
        flag = torch.tensor([ok if dist.get_backend() == "nccl" else 0], dtype=torch.int32, device=fdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            ids = [my_id if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)      # unconditional on every rank
            try:
                client.comm_init(rank, world, ids[0])
                ok = 1
            except Exception as e:   # noqa: BLE001
                ok, err = 0, f"comm_init: {e}"
            flag = torch.tensor([ok], dtype=torch.int32, device=fdev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                try:
                    my_part = last_partial[0]
                    outs, tms = [], []
                    for _ in range(5):
                        t1 = time.perf_counter()
                        outs.append(client.all_gather_combine(my_part))
                        tms.append((time.perf_counter() - t1) * 1e3)
                    same = all(o == res for o in outs)
                    native = {"ok": bool(same), "ms": round(statistics.median(tms), 3), "error": None if same else "result differs from the torch.distributed exchange",
                              "what": "blz_msm_all_gather_combine: ncclAllGather on the handle's stream + k_combine_partials, median of 5"}
                except Exception as e:   # noqa: BLE001
                    native = {"ok": False, "ms": None, "error": f"all_gather_combine: {e}"}
            else:
                native = {"ok": False, "ms": None, "error": err or "comm_init failed on another rank"}
        else:
            native = {"ok": False, "ms": None, "error": err or ("process group backend is not nccl" if dist.get_backend() != "nccl" else "RCCL not loadable on another rank")}
        agree = torch.tensor([1 if native["ok"] else 0], dtype=torch.int32, device=fdev)
        dist.all_reduce(agree, op=dist.ReduceOp.MIN)
        if native["ok"] and int(agree.item()) == 0:
            native = {"ok": False, "ms": native["ms"], "error": "the exchange failed on another rank"}
        wd.disarm()


    return native


def window_table_leg(ctx):
    """the headline workload on a handle that opted in to the resident-base window table: `window_table`"""
    args, rank, world, multi, dist, torch, tdev, dev, gather_dev, wd = (ctx.args, ctx.rank, ctx.world, ctx.multi, ctx.dist, ctx.torch, ctx.tdev, ctx.dev,
                                                                        ctx.gather_dev, ctx.wd)
    client, params, d_sc, d_pts, queue, lay, ranged, n, n_loc, cid = (ctx.client, ctx.params, ctx.d_sc, ctx.d_pts, ctx.queue, ctx.lay, ctx.ranged, ctx.n,
                                                                      ctx.n_loc, ctx.cid)
    res, last_partial, hbm_mode, calib, L, tf = ctx.res, ctx.last_partial, ctx.hbm_mode, ctx.calib, ctx.L, ctx.tf
    CURVE, LOG_N, NTT_LOG, HBM_PEAK_GBS = ctx.CURVE, ctx.LOG_N, ctx.NTT_LOG, ctx.HBM_PEAK_GBS
    import blaze_amd
    from blaze_amd import DeviceBuffer
    from blaze_amd._lib import check
    from blaze_amd.driver_client import DriverClient
    from blaze_amd.ingo_msm import Curve, MSMClient, MSMInit, MSMInput, MSMParams, PointMemoryType
    from blaze_amd.multi_gpu import SHARD_SCALARS_FROM_HOST, shard_layout_ex, sharded_msm
    # ---- the same workload with the resident-base window table (extra key, never the headline value): an opted-in
    # handle tabulates the window multiples of the bases once (blz_msm_set_window_table) and then runs fewer, wider
    # windows into one bucket set.  Every rank runs it on its shard (the exchange included), timed like the headline.
    table_rec = None
    # (a rank with a scalar range tabulates 2^(lo + c j) P: its few windows share one bucket set).  The leg is an extra: it
    # must never cost the headline line - a rank's failure is caught and reported, and the only collectives of the leg come
    # after it, reached by every rank whatever happened to it (N > 1: the ranks' tasks are timed without the 144-byte exchange)
    if hbm_mode and not args.no_extras and os.environ.get("BLAZE_BENCH_TABLE", "1") == "1":
        wd.arm(900, "window-table leg")
        terr, tdt, tinfo, first_ms, tkernel, n_before, until_ms, paced_ms, alloc_ms = None, -1.0, {"bytes": 0, "window_bits": 0, "windows": 0, "build_ms": 0.0}, 0.0, 0.0, 0, 0.0, 0.0, 0.0
        first_plain, paced_plain = True, 0
        k_t = args.steps
        tcl = None
        try:
            tcl = MSMClient(MSMInit(PointMemoryType.HBM, False, Curve[CURVE]), DriverClient(dev))
            if ranged:
                tcl.set_scalar_range(lay["bit_lo"], lay["bit_hi"])

            plain_flags = []

            def tsubmit():
                tcl.initialize(params)
                tcl.start_process()
                tcl.set_data(MSMInput(None, d_sc, params))
                plain_flags.append(tcl.window_table_info()["bytes"] == 0)   # (of the task just launched: did it take the plain path?)

            def tcollect():
                tcl.wait_result()
                return tcl.result().result, tcl.get_api()

            def trun(k):
                out, pending, submitted = [], 0, 0
                while submitted < k or pending:
                    if pending >= queue or submitted >= k:
                        out.append(tcollect())
                        pending -= 1
                    if submitted < k:
                        tsubmit()
                        submitted += 1
                        pending += 1
                return out

            # The build is paced by the tasks (arena_tables.hip arena_points_table): every task over the bases first enqueues four
            # ~5.5 ms chunks of it and takes the plain path until the table is complete.  first_task_ms is the first task's
            # latency (plain path + its four chunks); a few more tasks show the surcharge in a stream; then the host says it
            # would rather have the table now (prepare_window_table with a wait: all the remaining chunks at once) and the
            # steady state is timed.
            tsubmit()                       # (a fresh handle's first task pays its workspace allocations: not the table's doing)
            tcollect()
            tcl.set_window_table(True)
            # the table's allocation (80 GiB: 0.3 ms on a clean device, seconds when the driver first has to scrub memory an
            # earlier process freed - and a hipMalloc stalls every HIP call of the process, on any thread) belongs with the load:
            # prepare_window_table without a wait allocates and enqueues the first chunks
            t1 = time.perf_counter()
            tcl.prepare_window_table(n_loc, (0, 0), 0)
            alloc_ms = (time.perf_counter() - t1) * 1e3
            del plain_flags[:]
            t1 = time.perf_counter()
            tsubmit()
            first = tcollect()
            first_ms = (time.perf_counter() - t1) * 1e3
            first_plain = plain_flags[0]
            t_sw = time.perf_counter()
            paced = trun(4)
            paced_ms = (time.perf_counter() - t_sw) / 4 * 1e3
            n_before = sum(1 for f_ in plain_flags if f_)     # measured: a small table is complete after a task or two
            paced_plain = sum(1 for f_ in plain_flags[1:] if f_)
            if any(r_ != first[0] for r_, _ in paced):
                raise RuntimeError("results differ while the table is being built")
            t_sw = time.perf_counter()
            tcl.prepare_window_table(n_loc, (0, 0), -1)
            until_ms = (time.perf_counter() - t_sw) * 1e3
            tsubmit()
            tinfo = tcl.window_table_info()
            tcollect()
            trun(max(2, args.warmup))
            torch.cuda.synchronize(tdev)
            t1 = time.perf_counter()
            tdone = trun(k_t)
            torch.cuda.synchronize(tdev)
            tdt = time.perf_counter() - t1
            want = last_partial[0] if multi else res     # this rank's own (partial) result of the headline loop
            if tdone[-1][0] != want or first[0] != want:
                terr = "the window-table result differs from the headline's"
            tkernel = statistics.mean(a["accumulate_kernel_ms"] for _, a in tdone)
        except Exception as e:   # noqa: BLE001 - an extra key, never fatal
            terr = f"{type(e).__name__}: {e}"
        try:
            if tcl is not None:
                tcl.close()
        except Exception:   # noqa: BLE001
            pass
        if multi:
            fdev2 = gather_dev if gather_dev is not None else "cpu"
            t = torch.tensor([tdt if terr is None else -1.0, 0.0 if terr is None else 1.0], dtype=torch.float64, device=fdev2)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            tdt = float(t[0].item())
            if float(t[1].item()) > 0 and terr is None:
                terr = "the leg failed on another rank"
        if rank == 0:
            if terr is not None:
                table_rec = {"error": terr}
            else:
                table_rec = {"ms_per_step": round(tdt / k_t * 1e3, 3), "msm_per_s": round(k_t / tdt, 4), "steps": k_t,
                             "used": tinfo["bytes"] > 0, "table_bytes_per_gpu": tinfo["bytes"], "window_bits": tinfo["window_bits"],
                             "windows": tinfo["windows"], "build_ms": round(tinfo["build_ms"], 1),
                             "prepare_no_wait_ms": round(alloc_ms, 1), "first_task_ms": round(first_ms, 1), "ms_per_task_while_building": round(paced_ms, 1),
                             "tasks_on_the_plain_path": n_before, "first_task_on_the_plain_path": bool(first_plain), "paced_tasks_on_the_plain_path": f"{paced_plain} of 4",
                             "prepare_wait_ms": round(until_ms, 1), "kernel_ms": round(tkernel, 3),
                             "result_check": "this rank's result bytes equal its result in the headline loop (which the oracle checked"
                                             + (" after the exchange)" if multi else ")"),
                             "what": "opt-in blz_msm_set_window_table: the bases' window multiples 2^(c j) P tabulated once per load - paced by "
                                     "the tasks (four ~5.5 ms chunks ahead of each task, which takes the plain path meanwhile: first_task_ms, "
                                     "ms_per_task_while_building), the rest at once when the host asks for it (prepare_wait_ms) - then every "
                                     "window's digit added into one bucket set; same steps / queue as the headline" + (", the slowest rank's time, without the 144-byte exchange" if multi else "")}
        wd.disarm()

    return table_rec


def multi_rank_legs(ctx):
    """N > 1: the plain element split beside the headline's layout, and the reference's HBM flow per rank: (`alt_layout_elements`, `hbm_flow`)"""
    args, rank, world, multi, dist, torch, tdev, dev, gather_dev, wd = (ctx.args, ctx.rank, ctx.world, ctx.multi, ctx.dist, ctx.torch, ctx.tdev, ctx.dev,
                                                                        ctx.gather_dev, ctx.wd)
    client, params, d_sc, d_pts, queue, lay, ranged, n, n_loc, cid = (ctx.client, ctx.params, ctx.d_sc, ctx.d_pts, ctx.queue, ctx.lay, ctx.ranged, ctx.n,
                                                                      ctx.n_loc, ctx.cid)
    res, last_partial, hbm_mode, calib, L, tf = ctx.res, ctx.last_partial, ctx.hbm_mode, ctx.calib, ctx.L, ctx.tf
    CURVE, LOG_N, NTT_LOG, HBM_PEAK_GBS = ctx.CURVE, ctx.LOG_N, ctx.NTT_LOG, ctx.HBM_PEAK_GBS
    import blaze_amd
    from blaze_amd import DeviceBuffer
    from blaze_amd._lib import check
    from blaze_amd.driver_client import DriverClient
    from blaze_amd.ingo_msm import Curve, MSMClient, MSMInit, MSMInput, MSMParams, PointMemoryType
    from blaze_amd.multi_gpu import SHARD_SCALARS_FROM_HOST, shard_layout_ex, sharded_msm
    # ---- N > 1: the PLAIN ELEMENT SPLIT beside the headline's layout, and the reference's HBM flow per rank (extra keys).
    # The headline shards by blz_msm_shard_layout_ex's pick for resident scalars (scalar ranges of larger element chunks at
    # 2 / 4 / 8 ranks); `alt_layout_elements` times the same job cut into plain element chunks, so a scaling run records both
    # (the ranged layout had never met N > 1 hardware).  `hbm_flow` is the layout_ex pick for scalars that come from
    # host memory with every task - the element split - timed with pageable host scalars.  Each rank's tasks are timed
    # without the 144-byte exchange (the slowest rank's time is reported); ONE exchange per leg checks the result.
    alt_rec = None
    hbm_flow = cfg2 = None
    el_client, el_params, el_sc, el_pts = client, params, d_sc, None
    if multi and not args.no_extras and hbm_mode:
        wd.arm(900, "element-split / hbm_flow legs")
        err, dt_alt, api_alt, part_alt = None, -1.0, None, None
        # the element split, asked for by name (R = 1); blz_msm_shard_layout_ex's own pick for host scalars is recorded beside it
        # (it IS the element split at 2^26 on 2 / 4 / 8 ranks: tests/test_dist.py, profiles/r04_shard_layouts.txt)
        lay_e = shard_layout_ex(Curve[CURVE], n, rank, world, SHARD_SCALARS_FROM_HOST, 1)
        pick_host = shard_layout_ex(Curve[CURVE], n, rank, world, SHARD_SCALARS_FROM_HOST)["ranges"]
        try:
            if (lay_e["first"], lay_e["count"], lay_e["bit_lo"], lay_e["bit_hi"]) != (lay["first"], lay["count"], lay["bit_lo"], lay["bit_hi"]):
                el_pts = DeviceBuffer(dev, max(lay_e["count"], 1) * 96)
                el_sc = DeviceBuffer(dev, max(lay_e["count"], 1) * 32)
                check(blaze_amd.aux().blz_synth_points(dev, cid, el_pts.ptr, lay_e["count"], 1, lay_e["first"]))
                check(blaze_amd.aux().blz_synth_scalars_at(dev, cid, el_sc.ptr, lay_e["count"], 0xB1A2E, lay_e["first"]))
                el_client = MSMClient(MSMInit(PointMemoryType.HBM, False, Curve[CURVE]), DriverClient(dev))
                el_base = 1 << 40   # its own extent of the device arena, far from the headline's bases at 0
                el_client.load_data_to_hbm(el_pts, el_base, 0)
                el_params = MSMParams(lay_e["count"], (el_base, 0))
                if (lay_e["bit_lo"], lay_e["bit_hi"]) != (0, 256):
                    el_client.set_scalar_range(lay_e["bit_lo"], lay_e["bit_hi"])
                stream(ctx, el_client, el_params, None, el_sc, max(2, args.warmup))
                dt_alt, outs_alt, apis_alt = stream(ctx, el_client, el_params, None, el_sc, args.steps)
                part_alt, api_alt = outs_alt[-1], apis_alt[-1]
                if any(o != part_alt for o in outs_alt):
                    err = "a rank's results differ from task to task"
        except Exception as e:   # noqa: BLE001 - an extra key, never fatal
            err = f"{type(e).__name__}: {e}"
        ran_alt = el_client is not client
        dt_alt, err = all_ranks(ctx, dt_alt if ran_alt else 0.0, err)
        if err is None and ran_alt:
            full = sharded_msm(part_alt, client.combine_partials, dist, gather_dev)
            if full != res:
                err = "the element split's combined result differs from the headline's"
        if rank == 0:
            if err is not None:
                alt_rec = {"error": err}
            elif ran_alt:
                alt_rec = {"ms_per_step": round(dt_alt / args.steps * 1e3, 3), "msm_per_s": round(args.steps / dt_alt, 4), "steps": args.steps,
                           "shard_rank0": lay_e, "kernel_ms": round(api_alt["accumulate_kernel_ms"], 3),
                           "what": "the same job cut into plain element chunks (BLAZE_SHARD=elements would make it the headline): the slowest "
                                   "rank's time per task, two in flight, without the 144-byte exchange",
                           "result_check": "one exchange of the ranks' partials: bytes equal the headline result (which the oracle checked)"}
            else:
                alt_rec = {"same_as_headline": True}
        # the reference's HBM flow on every rank: bases resident, the rank's scalars from pageable host memory with every task
        err, dt_h, part_h, set_ms, done_at = None, -1.0, None, [], []
        k_hf = max(4, min(args.steps, 8))
        try:
            sc_host = el_sc.download()
            stream(ctx, el_client, el_params, None, sc_host, 2)
            dt_h, outs_h, _ = stream(ctx, el_client, el_params, None, sc_host, k_hf, on_set=set_ms.append, on_done=done_at.append)
            part_h = outs_h[-1]
            if any(o != part_h for o in outs_h):
                err = "a rank's results differ from task to task"
            del sc_host
        except Exception as e:   # noqa: BLE001
            err = f"{type(e).__name__}: {e}"
        gaps = [(b_ - a_) * 1e3 for a_, b_ in zip(done_at, done_at[1:])]
        steady = statistics.median(gaps) if gaps else -1.0
        dt_h, err = all_ranks(ctx, steady, err)
        if err is None:
            full = sharded_msm(part_h, client.combine_partials, dist, gather_dev)
            if full != res:
                err = "the combined result differs from the headline's"
        if rank == 0:
            hbm_flow = {"error": err} if err is not None else {
                "ms_per_msm_steady": round(dt_h, 3), "msm_per_s_steady": round(1e3 / dt_h, 4), "set_data_ms_median": round(statistics.median(set_ms), 3),
                "msms": k_hf, "tasks_in_flight": queue, "shard_rank0": lay_e, "layout_ex_pick_for_host_scalars_ranges": pick_host,
                "what": f"2^{LOG_N} BLS12-381 on {world} ranks: each rank's bases in its device arena, its scalars from pageable host memory "
                        "every task (tests/integration_msm_hbm.rs:57-100), element split; the "
                        "slowest rank's steady-state interval between results; PCIe-inclusive, never the headline value",
                "result_check": {"ok": True, "method": "one exchange of the ranks' partials: bytes equal the headline result"}}
        wd.disarm()

    return alt_rec, hbm_flow


def reference_flows(ctx):
    """one GPU: the reference's HBM flow with host scalars and config 2 (DMA mode, host buffers): (`hbm_flow`, `config2_dma`)"""
    args, rank, world, multi, dist, torch, tdev, dev, gather_dev, wd = (ctx.args, ctx.rank, ctx.world, ctx.multi, ctx.dist, ctx.torch, ctx.tdev, ctx.dev,
                                                                        ctx.gather_dev, ctx.wd)
    client, params, d_sc, d_pts, queue, lay, ranged, n, n_loc, cid = (ctx.client, ctx.params, ctx.d_sc, ctx.d_pts, ctx.queue, ctx.lay, ctx.ranged, ctx.n,
                                                                      ctx.n_loc, ctx.cid)
    res, last_partial, hbm_mode, calib, L, tf = ctx.res, ctx.last_partial, ctx.hbm_mode, ctx.calib, ctx.L, ctx.tf
    CURVE, LOG_N, NTT_LOG, HBM_PEAK_GBS = ctx.CURVE, ctx.LOG_N, ctx.NTT_LOG, ctx.HBM_PEAK_GBS
    import blaze_amd
    from blaze_amd import DeviceBuffer
    from blaze_amd._lib import check
    from blaze_amd.driver_client import DriverClient
    from blaze_amd.ingo_msm import Curve, MSMClient, MSMInit, MSMInput, MSMParams, PointMemoryType
    from blaze_amd.multi_gpu import SHARD_SCALARS_FROM_HOST, shard_layout_ex, sharded_msm
    hbm_flow = cfg2 = None
    # ---- the reference's own flows, timed as the reference runs them (extra keys, never the headline value)
    if rank == 0 and world == 1 and not multi and not args.no_extras and hbm_mode:
        # (1) tests/integration_msm_hbm.rs:57-100: bases resident in the card's memory (loaded once), the SCALARS
        # come from a host Vec<u8> with every task; two tasks in flight like the headline.  The 2 GiB host -> device
        # copy of task k+1 runs under the accumulation of task k (copy stream + two staging sets).
        sc_host = d_sc.download()
        k_hf = max(4, min(args.steps, 8))
        set_ms, done_at = [], []
        wd.arm(600, "hbm_flow leg")
        stream(ctx, client, params, None, sc_host, 2)
        t_hf, outs, _ = stream(ctx, client, params, None, sc_host, k_hf, on_set=set_ms.append, on_done=done_at.append)
        t_hf = t_hf / k_hf * 1e3
        wd.disarm()
        # steady state: the interval between consecutive results (the first MSM of the leg pays its 2 GiB copy with the
        # GPU idle - pipeline fill - which a stream of tasks pays once)
        gaps = [(b - a) * 1e3 for a, b in zip(done_at, done_at[1:])]
        t_steady = statistics.median(gaps) if gaps else t_hf
        ok_hf = all(o == res for o in outs)
        if not ok_hf:
            raise SystemExit("bench: hbm_flow result differs from the (checked) headline result")
        hbm_flow = {"ms_per_msm_steady": round(t_steady, 3), "msm_per_s_steady": round(1e3 / t_steady, 4),
                    "ms_per_msm_incl_pipeline_fill": round(t_hf, 3), "set_data_ms_median": round(statistics.median(set_ms), 3),
                    "msms": k_hf, "tasks_in_flight": queue,
                    "what": f"2^{LOG_N} BLS12-381: bases in the device arena, scalars from pageable host memory every task "
                            "(tests/integration_msm_hbm.rs:57-100); PCIe-inclusive, never the headline value",
                    "result_check": {"ok": True, "method": "bytes equal the headline result (same scalars), which the oracle checked"}}
        del sc_host
        # (2) config 2, tests/integration_msm.rs:149-207 with the timers of :338-355: 2^22 elements, points AND scalars
        # handed over as host buffers through set_data (DMA mode), one task at a time
        n2 = min(1 << 22, n_loc)
        p2, s2 = d_pts.download(n2 * 96), d_sc.download(n2 * 32)
        c2 = MSMClient(MSMInit(PointMemoryType.DMA, False, Curve[CURVE]), DriverClient(dev))
        prm2 = MSMParams(n2, None)
        runs = []
        wd.arm(600, "config2_dma leg")
        for i in range(2 + 5):
            t1 = time.perf_counter()
            c2.initialize(prm2)
            c2.start_process()
            c2.set_data(MSMInput(p2, s2, prm2))
            t2 = time.perf_counter()
            c2.wait_result()
            r2 = c2.result().result
            t3 = time.perf_counter()
            if i >= 2:
                runs.append(((t2 - t1) * 1e3, (t3 - t2) * 1e3, (t3 - t1) * 1e3))
        wd.disarm()
        c2.close()
        chk2 = None
        if not args.no_check:
            import oracle

            k2 = oracle.index_weighted_sum(CURVE, s2, n2, 0, threads=min(64, host_threads()))
            if r2 != oracle.result_from_affine(CURVE, oracle.generator_mul(CURVE, k2)):
                raise SystemExit("bench: the config2_dma result is WRONG")
            chk2 = {"ok": True, "method": "result == (sum_i s_i (i+1) mod r) G over the 2^22 scalars, CPU oracle"}
        med = [round(statistics.median(x), 3) for x in zip(*runs)]
        cfg2 = {"dur_set_data_ms": med[0], "dur_wait_result_ms": med[1], "dur_full_ms": med[2], "samples": len(runs),
                "what": f"config 2: {n2} BLS12-381 elements, host points + scalars through set_data (DMA mode), one task at a time, "
                        "timers as tests/integration_msm.rs:338-355; PCIe-inclusive",
                "result_check": chk2}
        del p2, s2

    return hbm_flow, cfg2


def configs_3_4_and_small(ctx):
    """one GPU: BASELINE configs 3 and 4 (one rank's task) and lone small tasks: (`config3_bn254_pf8`, `config4_rank_task`, `lone_small_msm`)"""
    args, rank, world, multi, dist, torch, tdev, dev, gather_dev, wd = (ctx.args, ctx.rank, ctx.world, ctx.multi, ctx.dist, ctx.torch, ctx.tdev, ctx.dev,
                                                                        ctx.gather_dev, ctx.wd)
    client, params, d_sc, d_pts, queue, lay, ranged, n, n_loc, cid = (ctx.client, ctx.params, ctx.d_sc, ctx.d_pts, ctx.queue, ctx.lay, ctx.ranged, ctx.n,
                                                                      ctx.n_loc, ctx.cid)
    res, last_partial, hbm_mode, calib, L, tf = ctx.res, ctx.last_partial, ctx.hbm_mode, ctx.calib, ctx.L, ctx.tf
    CURVE, LOG_N, NTT_LOG, HBM_PEAK_GBS = ctx.CURVE, ctx.LOG_N, ctx.NTT_LOG, ctx.HBM_PEAK_GBS
    import blaze_amd
    from blaze_amd import DeviceBuffer
    from blaze_amd._lib import check
    from blaze_amd.driver_client import DriverClient
    from blaze_amd.ingo_msm import Curve, MSMClient, MSMInit, MSMInput, MSMParams, PointMemoryType
    from blaze_amd.multi_gpu import SHARD_SCALARS_FROM_HOST, shard_layout_ex, sharded_msm
    # ---- the other two BASELINE configs on this GPU (extra keys): config 3 - 2^26 BN254, precompute
    # factor 8, the 2^29 bases (32 GiB) resident in the device arena, scalars-only set_data - and one rank's task of config 4 -
    # rank 0 of 8 of a 2^26 BLS12-377 job as blz_msm_shard_layout cuts it.  Each checked against the oracle by linearity.
    cfg3 = cfg4 = lone_small = None
    if rank == 0 and world == 1 and not multi and not args.no_extras and hbm_mode and LOG_N == 26:
        def weighted_expect(curve_name, sc_bytes, count, first):
            import oracle

            kk = oracle.index_weighted_sum(curve_name, sc_bytes, count, first, threads=min(64, host_threads()))
            return oracle.result_from_affine(curve_name, oracle.generator_mul(curve_name, kk))

        wd.arm(600, "config 3 leg")
        try:
            client.close()        # the BLS12-381 handle's workspace and the arena's 14 GiB: not needed any more
            L.blz_arena_release(dev)
            n3, c3 = 1 << 26, int(Curve["BN254"])
            p3 = DeviceBuffer(dev, n3 * 8 * 64)
            s3 = DeviceBuffer(dev, n3 * 32)
            check(blaze_amd.aux().blz_synth_points(dev, c3, p3.ptr, n3, 8, 0))          # 2^(32 j) (i + 1) G, j < 8: the reference's precompute
            check(blaze_amd.aux().blz_synth_scalars_at(dev, c3, s3.ptr, n3, 0xC0F3, 0))
            cl3 = MSMClient(MSMInit(PointMemoryType.HBM, True, Curve["BN254"]), DriverClient(dev))
            cl3.load_data_to_hbm(p3, 0, 0)
            p3.free()
            prm3 = MSMParams(n3, (0, 0))
            stream(ctx, cl3, prm3, None, s3, 2)
            dt3, outs3, apis3 = stream(ctx, cl3, prm3, None, s3, 4)
            k3 = statistics.mean(a_["accumulate_kernel_ms"] for a_ in apis3)
            if os.environ.get("BENCH_DEBUG"):
                print("[bench debug] config 3 tasks:", [{k_: round(v_, 2) for k_, v_ in a_.items() if k_.endswith("_ms")} for a_ in apis3], file=sys.stderr, flush=True)
            bytes3 = n3 * (32 + 8 * 64)
            chk3 = None
            if not args.no_check:
                if outs3[-1] != weighted_expect("BN254", s3.download(), n3, 0) or any(o != outs3[-1] for o in outs3):
                    raise SystemExit("bench: the config 3 result is WRONG")
                chk3 = {"ok": True, "method": "result == (sum_i s_i (i+1) mod r) G over all 2^26 scalars (the bases are 2^(32 j) (i+1) G), CPU oracle"}
            def traffic3(key, kernel_ms):
                """counter traffic of config 3's accumulation (profiles/pmc_traffic.json), quoted while the kernel still matches the record"""
                try:
                    rec3 = json.load(open(tf)).get(key)
                    if rec3 and abs(kernel_ms - rec3["kernel_ms_at_measurement"]) <= 0.10 * kernel_ms:
                        return rec3["hbm_bytes_per_launch"]
                except Exception:   # noqa: BLE001
                    pass
                return None

            cfg3 = {"ms_per_msm": round(dt3 / 4 * 1e3, 3), "msms": 4, "tasks_in_flight": queue, "kernel_ms": round(k3, 3),
                    "window_bits": int(apis3[-1]["window_bits"]), "windows": int(apis3[-1]["windows"]),
                    "roofline": {"bound": "hbm", "kernel": "k_accumulate", "algorithmic_bytes_per_launch": bytes3,
                                 "achieved": round(bytes3 / (k3 * 1e-3) / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": round(bytes3 / (k3 * 1e-3) / 1e9 / HBM_PEAK_GBS, 6), "traffic": traffic3("k_accumulate_config3_exact", k3)},
                    "what": "config 3: 2^26 BN254 elements, precompute factor 8: 2^29 bases (32 GiB) resident in the device arena, scalars-only "
                            "set_data (device-resident scalars), tests/integration_msm_hbm.rs flow", "result_check": chk3}
            # the same tasks on the checked-table plan (opt-in, blz_msm_set_precompute_plan): the resident table is checked once
            # against precompute_base_* on the device, then every task sums the 2^28 even bases with 64-bit chunks
            try:
                cl3.set_precompute_plan(True)
                t_chk = time.perf_counter()
                ok3 = cl3.prepare_precompute_plan(n3, (0, 0))
                t_chk = (time.perf_counter() - t_chk) * 1e3
                info3 = cl3.precompute_plan_info()
                stream(ctx, cl3, prm3, None, s3, 2)
                dt3p, outs3p, apis3p = stream(ctx, cl3, prm3, None, s3, 4)
                used3 = cl3.precompute_plan_info()["used"]
                k3p = statistics.mean(a_["accumulate_kernel_ms"] for a_ in apis3p)
                if any(o != outs3[-1] for o in outs3p):
                    raise SystemExit("bench: the config 3 result on the checked-table plan differs from the exact path's")
                cfg3["checked_table_plan"] = {
                    "ms_per_msm": round(dt3p / 4 * 1e3, 3), "msms": 4, "tasks_in_flight": queue, "kernel_ms": round(k3p, 3), "plan_taken": bool(used3 and ok3),
                    "window_bits": int(apis3p[-1]["window_bits"]), "windows": int(apis3p[-1]["windows"]),
                    "table_check_ms": round(info3["check_ms"], 1), "prepare_wall_ms": round(t_chk, 1), "even_base_copy_bytes": info3["even_copy_bytes"],
                    "device_memory": apis3p[-1]["device_memory"],
                    "roofline": {"bound": "hbm", "kernel": "k_accumulate", "algorithmic_bytes_per_launch": bytes3,
                                 "achieved": round(bytes3 / (k3p * 1e-3) / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": round(bytes3 / (k3p * 1e-3) / 1e9 / HBM_PEAK_GBS, 6), "traffic": traffic3("k_accumulate_config3_plan", k3p)},
                    "what": "the same four tasks after blz_msm_set_precompute_plan(1): table checked once on the device (B_j == 2^32 B_(j-1), B_0 on the "
                            "curve), then 2^28 even bases x 64-bit chunks: 3 windows of 22 / 22 / 21 bits, 12 bucket additions per element instead of 16",
                    "result_check": {"ok": True, "method": "bytes equal the exact path's result" + (" (which the oracle checked)" if chk3 else "")}}
            except SystemExit:
                raise
            except Exception as e:   # noqa: BLE001
                cfg3["checked_table_plan"] = {"error": f"{type(e).__name__}: {e}"}
            cl3.close()
            s3.free()
            L.blz_arena_release(dev)
        except SystemExit:
            raise
        except Exception as e:   # noqa: BLE001 - an extra key, never fatal
            cfg3 = {"error": f"{type(e).__name__}: {e}"}
        wd.arm(600, "config 4 leg")
        try:
            c4 = Curve["BLS377"]
            lay4 = shard_layout_ex(c4, 1 << 26, 0, 8, 0)
            n4 = lay4["count"]
            p4 = DeviceBuffer(dev, n4 * 96)
            s4 = DeviceBuffer(dev, n4 * 32)
            check(blaze_amd.aux().blz_synth_points(dev, int(c4), p4.ptr, n4, 1, lay4["first"]))
            check(blaze_amd.aux().blz_synth_scalars_at(dev, int(c4), s4.ptr, n4, 0x377, lay4["first"]))
            cl4 = MSMClient(MSMInit(PointMemoryType.HBM, False, c4), DriverClient(dev))
            cl4.load_data_to_hbm(p4, 0, 0)
            p4.free()
            if (lay4["bit_lo"], lay4["bit_hi"]) != (0, 256):
                cl4.set_scalar_range(lay4["bit_lo"], lay4["bit_hi"])
            prm4 = MSMParams(n4, (0, 0))
            stream(ctx, cl4, prm4, None, s4, 3)
            done4, set4 = [], []
            dt4, outs4, apis4 = stream(ctx, cl4, prm4, None, s4, 10, on_set=set4.append, on_done=done4.append)
            gaps4 = [round((b_ - a_) * 1e3, 2) for a_, b_ in zip(done4, done4[1:])]
            if os.environ.get("BENCH_DEBUG"):
                print("[bench debug] config 4 result intervals (ms):", gaps4, "set_data (ms):", [round(x_, 2) for x_ in set4], file=sys.stderr, flush=True)
            chk4 = None
            if not args.no_check:
                import numpy as np

                sc4 = np.frombuffer(s4.download(), dtype=np.uint8).reshape(n4, 32).copy()
                sc4[:, : lay4["bit_lo"] // 8] = 0          # the rank sums bits [bit_lo, bit_hi) of every scalar, result weighted 2^bit_lo:
                sc4[:, lay4["bit_hi"] // 8:] = 0           # the same bytes as the scalars with every other bit cleared
                if outs4[-1] != weighted_expect("BLS377", sc4.tobytes(), n4, lay4["first"]) or any(o != outs4[-1] for o in outs4):
                    raise SystemExit("bench: the config 4 rank-task result is WRONG")
                chk4 = {"ok": True, "method": "partial == (sum_i (s_i masked to the rank's bit range) (i+1) mod r) G over the rank's elements, CPU oracle"}
                del sc4
            cfg4 = {"ms_per_task": round(dt4 / 10 * 1e3, 3), "ms_per_task_steady": round(statistics.median(gaps4), 3), "tasks": 10, "tasks_in_flight": queue, "shard_rank0_of_8": lay4,
                    "kernel_ms": round(statistics.mean(a_["accumulate_kernel_ms"] for a_ in apis4), 3),
                    "window_bits": int(apis4[-1]["window_bits"]), "windows": int(apis4[-1]["windows"]),
                    "what": "config 4, one rank's share: rank 0 of 8 of a 2^26 BLS12-377 job as blz_msm_shard_layout_ex cuts it (resident "
                            "scalars), bases in the device arena, a stream of the rank's tasks; 8 x this GPU-time is the job's compute, the "
                            "RCCL exchange of the 144-byte partials is not in it", "result_check": chk4}
            cl4.close()
            s4.free()
            L.blz_arena_release(dev)
        except SystemExit:
            raise
        except Exception as e:   # noqa: BLE001
            cfg4 = {"error": f"{type(e).__name__}: {e}"}
        # ---- lone small tasks (extra key): the reference's own tests run MSM_SIZE = 8192 elements, one task at
        # a time (tests/integration_msm.rs:149-207).  Device-resident inputs, wall clock of initialize .. result, median of 15.
        wd.arm(300, "lone small MSMs")
        try:
            lone_small = {"what": "one BLS12-381 MSM at a time over device-resident inputs (bases in the arena), wall ms of initialize -> start_process -> "
                                  "set_data -> wait_result -> result, median of 15; 2^13 = the reference's default MSM_SIZE", "ms": {}}
            c5 = Curve["BLS381"]
            for lg in (13, 16, 20):
                n5 = 1 << lg
                p5 = DeviceBuffer(dev, n5 * 96)
                s5 = DeviceBuffer(dev, n5 * 32)
                check(blaze_amd.aux().blz_synth_points(dev, int(c5), p5.ptr, n5, 1, 0))
                check(blaze_amd.aux().blz_synth_scalars_at(dev, int(c5), s5.ptr, n5, 0x5A11 + lg, 0))
                cl5 = MSMClient(MSMInit(PointMemoryType.HBM, False, c5), DriverClient(dev))
                cl5.load_data_to_hbm(p5, 0, 0)
                p5.free()
                prm5 = MSMParams(n5, (0, 0))
                inp5 = MSMInput(None, s5, prm5)
                ts5, out5 = [], None
                for k5 in range(18):
                    t5 = time.perf_counter()
                    cl5.initialize(prm5); cl5.start_process(); cl5.set_data(inp5); cl5.wait_result()
                    out5 = cl5.result().result
                    if k5 >= 3:
                        ts5.append((time.perf_counter() - t5) * 1e3)
                if not args.no_check and out5 != weighted_expect("BLS381", s5.download(), n5, 0):
                    raise SystemExit(f"bench: the lone 2^{lg} result is WRONG")
                lone_small["ms"][f"2^{lg}"] = round(statistics.median(ts5), 3)
                cl5.close()
                s5.free()
                L.blz_arena_release(dev)
            lone_small["result_check"] = None if args.no_check else {"ok": True, "method": "each size: result == (sum_i s_i (i+1) mod r) G, CPU oracle"}
        except SystemExit:
            raise
        except Exception as e:   # noqa: BLE001
            lone_small = {"error": f"{type(e).__name__}: {e}"}
        wd.disarm()

    return cfg3, cfg4, lone_small


def ntt_leg(ctx):
    """the 2^27 NTT (replica per rank; rank 0 reports): `ntt_2e27`"""
    args, rank, world, multi, dist, torch, tdev, dev, gather_dev, wd = (ctx.args, ctx.rank, ctx.world, ctx.multi, ctx.dist, ctx.torch, ctx.tdev, ctx.dev,
                                                                        ctx.gather_dev, ctx.wd)
    client, params, d_sc, d_pts, queue, lay, ranged, n, n_loc, cid = (ctx.client, ctx.params, ctx.d_sc, ctx.d_pts, ctx.queue, ctx.lay, ctx.ranged, ctx.n,
                                                                      ctx.n_loc, ctx.cid)
    res, last_partial, hbm_mode, calib, L, tf = ctx.res, ctx.last_partial, ctx.hbm_mode, ctx.calib, ctx.L, ctx.tf
    CURVE, LOG_N, NTT_LOG, HBM_PEAK_GBS = ctx.CURVE, ctx.LOG_N, ctx.NTT_LOG, ctx.HBM_PEAK_GBS
    import blaze_amd
    from blaze_amd import DeviceBuffer
    from blaze_amd._lib import check
    from blaze_amd.driver_client import DriverClient
    from blaze_amd.ingo_msm import Curve, MSMClient, MSMInit, MSMInput, MSMParams, PointMemoryType
    from blaze_amd.multi_gpu import SHARD_SCALARS_FROM_HOST, shard_layout_ex, sharded_msm
    # ---- NTT 2^27 latency (replica per rank; rank 0 reports), timed like benches/ntt_bench.rs:34-39
    # minus the 100 ms sleep of reset(): initialize + start_process + wait_result on a resident buffer
    ntt = None
    if not args.no_ntt:
        from blaze_amd.ingo_ntt import NTT, NTTClient, NTTInput, NttInit

        try:
            client.close()
        except Exception:   # noqa: BLE001 - already closed by the config 3 leg
            pass
        nn = 1 << NTT_LOG
        d_in = DeviceBuffer(dev, 32 * nn)
        check(blaze_amd.aux().blz_synth_field_elements(dev, d_in.ptr, nn, 5))
        nc = NTTClient(NTT.Ntt, DriverClient(dev), log_size=NTT_LOG)
        nc.set_data(NTTInput(0, d_in))
        d_in.free()
        kms, wall = [], []
        sclk_ntt = None
        for i in range(2 + 10):
            if i == 2:   # the shader clock while the timed transforms run (the passes are power-limited like the MSM's accumulation)
                sclk_ntt = SclkSampler(torch, dev, period=0.01)
                sclk_ntt.start()
            t1 = time.perf_counter()
            nc.initialize(NttInit())
            nc.start_process(0)
            nc.wait_result()
            w = (time.perf_counter() - t1) * 1e3
            if i >= 2:
                kms.append(nc.last_kernel_ms())
                wall.append(w)
        sclk_ntt_rec = sclk_ntt.stop() if sclk_ntt is not None else None
        nb = 2 * 32 * nn
        k = statistics.median(kms)
        ntt = {"log_size": NTT_LOG, "ms": round(statistics.median(wall), 3), "kernel_ms": round(k, 3), "samples": 10,
               "roofline": {"bound": "hbm", "achieved": round(nb / (k * 1e-3) / 1e9, 3), "peak": HBM_PEAK_GBS,
                            "unit": "GB/s", "frac": round(nb / (k * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                            "algorithmic_bytes": nb, "traffic": None},
               "sclk_mhz_timed_transforms": sclk_ntt_rec}
        # The resource the passes saturate, beside the prescribed HBM figure: 32-bit integer multiply issue.  Per lane (8 elements)
        # and pass the 512-point kernel does 37 / 37 / 29 field products: Shoup products by table twiddles at
        # 143 v_mad_u64_u32, Montgomery products (pass 2's boundary factors, read from the per-element table) at 153, and a
        # 9-multiply-add quotient reduction for each un-twiddled output: 5309 + 5389 + 4237 = 14 935 per lane
        # (tests/test_isa_counts.py counts them in the code object; round 3 and most of round 4: 16 242, pass 2 stepping its
        # factors), n / 8 lanes per pass.
        ninfo = nc.info()
        ntt["device_bytes"] = ninfo["device_bytes"]
        ntt["pass2_reads_factor_table"] = ninfo["pass2_factor_table"]
        if NTT_LOG == 27:
            # (pass 2 reading its factor table: 29 Shoup + 8 Montgomery products; stepping its factors - what a handle without
            # memory for the table runs, blz_ntt_info says which: 36 + 10)
            pass2_table, pass2_stepped = 29 * 143 + 8 * 153 + 2 * 9, 36 * 143 + 10 * 153 + 2 * 9
            ntt_mads = (37 * 143 + 2 * 9) + (pass2_table if ninfo["pass2_factor_table"] else pass2_stepped) + (29 * 143 + 10 * 9)
            mads_t = ntt_mads * (nn // 8)
            peak_i = calib["mad_lane_ops_per_s"] if calib else 3.1e13
            ntt["roofline"]["integer_issue"] = {"unit": "v_mad_u64_u32 lane-ops/s", "achieved": round(mads_t / (k * 1e-3), 0), "peak": round(peak_i, 0),
                                                "frac": round(mads_t / (k * 1e-3) / peak_i, 4), "multiply_adds_per_transform": mads_t,
                                                "multiply_adds_per_lane": ntt_mads,
                                                "peak_source": "calibration kernel on this device, this run" if calib else "constant measured on another box"}
        # The reference's double-buffered host loop (tests/integration_ntt.rs:102-136) with 4 GiB pageable host vectors: the kernel
        # hides under the transfers; result + set_data of a cycle as two calls (one direction of the link at a time) and fused
        # into blz_ntt_exchange (both at once).  PCIe-inclusive, never `ms`.
        if rank == 0 and not args.no_extras and NTT_LOG >= 20:
            try:
                import numpy as np

                hx = np.empty(32 * nn, dtype=np.uint8)
                hy = np.empty(32 * nn, dtype=np.uint8)
                # every page of both vectors first touched by THIS thread, as a host that fills its input vector would: left to the
                # runtime's device -> host staging copy, the first touch put hx wherever that copy ran, and on some boxes the two
                # directions of the exchange then overlapped badly (125 - 158 ms per cycle in this process where
                # tools/pcie_inclusive_ntt.py measured 91 on the same box)
                hx[:] = 0
                hy[:] = 0
                nc.result_into(0, hx)        # canonical field elements

                def host_loop(fused, src, dst, cycles=5, max_cycles=24):
                    # Cycles until the last four agree within 4 % (at least `cycles`, at most max_cycles): this leg runs right behind
                    # legs that freed > 100 GiB of device memory, and the driver wipes freed memory in the background ON THE COPY ENGINES
                    # the exchange's two directions need - while that drains, a cycle takes the two-call time (158 ms, the directions
                    # serialised) and then drops to 91 within a cycle or two (tools/host_loop_check.sh: the same box, the same process).
                    # Every cycle is listed in the record.
                    ts = []
                    for i in range(max_cycles):
                        t1 = time.perf_counter()
                        bh, bk = i % 2, 1 - i % 2
                        nc.start_process(bk)
                        if fused:
                            nc.exchange(bh, src, dst)
                        else:
                            nc.result_into(bh, dst)
                            nc.set_data(NTTInput(bh, src))
                        nc.wait_result()
                        ts.append((time.perf_counter() - t1) * 1e3)
                        if i + 1 >= cycles and max(ts[-4:]) <= 1.04 * min(ts[-4:]):
                            break
                    return statistics.median(ts[-4:]), [round(t_, 1) for t_ in ts]

                two_calls, _ = host_loop(False, hx, hy)
                fused, fused_cycles = host_loop(True, hx, hy)
                ntt["host_loop_ms"] = round(fused, 2)
                ntt["host_loop"] = {"exchange_ms_per_transform": round(fused, 2), "exchange_cycles_ms": fused_cycles,
                                    "result_then_set_data_ms_per_transform": round(two_calls, 2),
                                    "host_bytes_each_way": 32 * nn, "host_memory": "pageable (numpy)",
                                    "what": "tests/integration_ntt.rs:102-136: start_process on one buffer, the previous result out of / the next input "
                                            "into the other, wait_result; median of the last 4 cycles, once four in a row agree within 4 % "
                                            "(the driver's background wipe of memory freed by the legs before this one shares the copy engines)"}
                # The same cycle over PAGE-LOCKED host vectors (blz_host_malloc): every piece a true asynchronous DMA, chained by events.
                # Beside the pageable figure because that one depends on how the runtime pins a pageable range piece by piece from two
                # host threads at once: on some boxes - and then for a whole process - the two directions serialise (158 ms: the two-call
                # time; HISTORY.md section 13), where tools/pcie_inclusive_ntt.py on the same box overlaps them (91 ms)
                try:
                    from blaze_amd._lib import HostBuffer

                    px, py = HostBuffer(dev, 32 * nn), HostBuffer(dev, 32 * nn)
                    ax, ay = px.array(), py.array()
                    ax[:] = hx
                    ay[:] = 0
                    pinned, pinned_cycles = host_loop(True, ax, ay)
                    ntt["host_loop_pinned_ms"] = round(pinned, 2)
                    ntt["host_loop"]["exchange_page_locked_ms_per_transform"] = round(pinned, 2)
                    ntt["host_loop"]["exchange_page_locked_cycles_ms"] = pinned_cycles
                    del ax, ay
                    px.free(); py.free()
                except Exception as e:   # noqa: BLE001
                    ntt["host_loop"]["exchange_page_locked_error"] = f"{type(e).__name__}: {e}"
                del hx, hy
            except Exception as e:   # noqa: BLE001 - an extra key, never fatal
                ntt["host_loop"] = {"error": f"{type(e).__name__}: {e}"}
        try:   # PMC record of the three passes, quoted only while it matches the kernels being timed (see above)
            rec = json.load(open(tf)).get(f"ntt_2e{NTT_LOG}_BLS381")
            if rec and abs(k - rec.get("kernel_ms_at_measurement", k)) <= 0.10 * k:
                ntt["roofline"]["traffic"] = rec["hbm_bytes_per_transform"]
        except Exception:
            pass
        nc.close()

    return ntt


def cpu_baselines(ctx):
    """the oracle's Pippenger / reference-semantics path / threaded NTT on the box's host cores: (`cpu_baseline`, `cpu_baseline_ref_semantics`); adds ntt['cpu_baseline']"""
    args, rank, world, multi, dist, torch, tdev, dev, gather_dev, wd = (ctx.args, ctx.rank, ctx.world, ctx.multi, ctx.dist, ctx.torch, ctx.tdev, ctx.dev,
                                                                        ctx.gather_dev, ctx.wd)
    client, params, d_sc, d_pts, queue, lay, ranged, n, n_loc, cid = (ctx.client, ctx.params, ctx.d_sc, ctx.d_pts, ctx.queue, ctx.lay, ctx.ranged, ctx.n,
                                                                      ctx.n_loc, ctx.cid)
    res, last_partial, hbm_mode, calib, L, tf = ctx.res, ctx.last_partial, ctx.hbm_mode, ctx.calib, ctx.L, ctx.tf
    CURVE, LOG_N, NTT_LOG, HBM_PEAK_GBS = ctx.CURVE, ctx.LOG_N, ctx.NTT_LOG, ctx.HBM_PEAK_GBS
    import blaze_amd
    from blaze_amd import DeviceBuffer
    from blaze_amd._lib import check
    from blaze_amd.driver_client import DriverClient
    from blaze_amd.ingo_msm import Curve, MSMClient, MSMInit, MSMInput, MSMParams, PointMemoryType
    from blaze_amd.multi_gpu import SHARD_SCALARS_FROM_HOST, shard_layout_ex, sharded_msm
    ntt = ctx.ntt
    # ---- CPU baselines on this box's host cores (baseline only): the oracle's Pippenger, parallel over
    # (element chunk x window) tasks on every affinity-visible thread, on the largest prefix of the workload that
    # an estimate says finishes in ~20 s (the whole 2^26 on a 256-thread host, 2^22 on 8 cores); the
    # reference-semantics path (naive sum of double-and-add scalar multiplications, tests/msm/mod.rs:326-335, one
    # core, n = 2^10); and a threaded radix-2 NTT beside ntt_2e27.
    cpu = cpu_ref = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle  # test infrastructure, used here only as the timed CPU baseline

        cores = host_threads()
        cbits = 16
        ls = int(os.environ.get("BLAZE_BENCH_CPU_LOGN", "0"))
        if ls == 0:
            # calibrate on 2^18 elements, then take the largest power of two (up to the whole workload) that the
            # measured rate finishes in about 20 s
            nc = min(1 << 18, n_loc)
            pc, scc = d_pts.download(nc * 96), d_sc.download(nc * 32)
            t1 = time.perf_counter()
            oracle.msm_pippenger(CURVE, pc, scc, nc, 1, threads=cores, cbits=cbits)
            per_elem = (time.perf_counter() - t1) / nc
            del pc, scc
            ls = 18
            while ls < LOG_N and (2 << ls) * per_elem <= 20.0:
                ls += 1
        ns = min(1 << ls, n_loc)
        pts = d_pts.download(ns * 96)
        sc = d_sc.download(ns * 32)
        t1 = time.perf_counter()
        got = oracle.msm_pippenger(CURVE, pts, sc, ns, 1, threads=cores, cbits=cbits)
        tc = time.perf_counter() - t1
        kk = oracle.index_weighted_sum(CURVE, sc, ns, 0, threads=min(64, cores))
        assert got == oracle.result_from_affine(CURVE, oracle.generator_mul(CURVE, kk)), "CPU baseline result wrong"
        cpu = {"value": round((ns / n) / tc, 6), "unit": "MSM/s", "cores": cores, "kind": "port",
               "sample": f"first 2^{ls} of the 2^{LOG_N} elements, {cbits}-bit signed windows: {tc:.2f} s wall on {cores} threads"
                         + ("" if ns == n else f"; value scaled linearly to 2^{LOG_N}")}
        # the ~20 s prefix above is scaled linearly; its anchor is ONE run over all 2^26 elements on a box of this pool
        # (tests/probes/cpu_baseline_full.py -> profiles/r06_cpu_baseline_2e26.json: 65.7 s on 16 threads, 0.991 of the scaled prefix)
        try:
            anc = json.load(open(os.path.join(ROOT, "profiles", "r06_cpu_baseline_2e26.json")))
            if LOG_N == 26 and anc.get("result_ok"):
                cpu["full_size_run_value"] = anc["value"]
                cpu["full_size_run_seconds"] = anc["seconds"]
                cpu["full_size_run_cores"] = anc["cores"]
                cpu["full_size_run_source"] = "profiles/r06_cpu_baseline_2e26.json (all 2^26 elements, measured once, not in this run)"
        except Exception:   # noqa: BLE001 - a citation, never fatal
            pass
        nr = 1 << 10
        t1 = time.perf_counter()
        oracle.msm_naive(CURVE, pts[: nr * 96], sc[: nr * 32], nr, 1)
        tr = time.perf_counter() - t1
        cpu_ref = {"value": round(nr / tr, 1), "unit": "elements/s", "cores": 1, "kind": "port", "seconds": round(tr, 3),
                   "sample": "reference-semantics check path (tests/msm/mod.rs:326-335): sum of 2^10 double-and-add scalar "
                             "multiplications on one core (config 1's shape)"}
        del pts, sc
        if ntt is not None:
            lc = NTT_LOG if cores >= 64 else min(NTT_LOG, 24)
            import numpy as np
            xin = np.random.default_rng(1).integers(0, 256, size=32 << lc, dtype=np.uint8)
            xin[31::32] &= 0x3F
            t1 = time.perf_counter()
            oracle.ntt(CURVE, xin, lc, threads=min(cores, 64))
            tn = time.perf_counter() - t1
            ntt["cpu_baseline"] = {"value": round(tn * 1e3, 1), "unit": "ms", "cores": min(cores, 64), "kind": "port",
                                   "sample": f"threaded radix-2 NTT of 2^{lc} elements (oracle), one transform"}
            del xin

    return cpu, cpu_ref


