"""Multi-GPU host path on CPU: world_size 2 over gloo.  The per-rank device MSM and the device
combine are stubbed with the CPU oracle (this is a test, the oracle is the checker); what is under
test is blaze_amd.multi_gpu: the shard partition, the single all-gather of fixed-size partials in
rank order, and that every rank ends with identical bytes equal to the unsharded result."""
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, curve, n, pf, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    import oracle
    from blaze_amd.multi_gpu import shard_range, sharded_msm

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pts, sc, expected = oracle.input_generator(curve, n, pf, 4242)   # same on every rank
        pb = oracle.point_bytes(curve) * pf
        lo, hi = shard_range(n, rank, world)
        partial = oracle.msm_pippenger(curve, bytes(pts[lo * pb: hi * pb]), bytes(sc[lo * 32: hi * 32]), hi - lo, pf, threads=2)

        def combine(partials, count):  # what MSMClient.combine_partials does on the GPU
            rb = oracle.result_bytes(curve)
            acc = None
            for i in range(count):
                xy, on = oracle.decode_result(curve, partials[i * rb: (i + 1) * rb])
                assert on
                acc = oracle.point_add(curve, acc, xy)
            return oracle.result_from_affine(curve, acc)

        full = sharded_msm(partial, combine, dist)
        q.put((rank, full == expected, full.hex()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("curve,n,pf", [("BLS381", 777, 1), ("BN254", 300, 8)])
def test_sharded_msm_world2_gloo(curve, n, pf):
    import torch.multiprocessing as mp

    import oracle

    oracle.build()
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, curve, n, pf, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert len({h for _, _, h in res}) == 1          # every rank holds the same normalised bytes


def _gpu_worker(rank, world, port, curve, n, native, q):
    """One rank of the device path: real MSMClient on the rank's GPU (device 0 for every rank when the box has
    only one), real combine on the device.  `native`: exchange inside the library (RCCL); needs one GPU per rank."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist

    import blaze_amd
    import oracle
    from blaze_amd.driver_client import DriverClient
    from blaze_amd.ingo_msm import Curve, MSMClient, MSMInit, MSMInput, MSMParams, PointMemoryType
    from blaze_amd.multi_gpu import shard_range, sharded_msm

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ndev = blaze_amd.lib().blz_device_count()
        dev = rank if ndev >= world else 0
        pts, sc, expected = oracle.input_generator(curve, n, 1, 777)
        pb = oracle.point_bytes(curve)
        lo, hi = shard_range(n, rank, world)
        cl = MSMClient(MSMInit(PointMemoryType.DMA, False, Curve[curve]), DriverClient(dev))
        params = MSMParams(hi - lo, None)
        cl.initialize(params)
        cl.start_process()
        cl.set_data(MSMInput(bytes(pts[lo * pb: hi * pb]), bytes(sc[lo * 32: hi * 32]), params))
        cl.wait_result()
        partial = cl.result().result
        if native:
            ids = [MSMClient.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            cl.comm_init(rank, world, ids[0])
            full = cl.all_gather_combine(partial)
            cl.comm_free()
        else:
            full = sharded_msm(partial, cl.combine_partials, dist)
        cl.close()
        q.put((rank, full == expected, full.hex()))
    finally:
        dist.destroy_process_group()


def _run_gpu_world(curve, n, world, native):
    import torch.multiprocessing as mp

    import oracle

    oracle.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, world, port, curve, n, native, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert len({h for _, _, h in res}) == 1


@pytest.mark.gpu
@pytest.mark.parametrize("curve", ["BLS381", "BN254"])
def test_sharded_msm_world2_device_path(gpu, curve):
    """The same world-2 flow with nothing stubbed: each rank runs the device MSM of its shard and the device
    combine (k_combine_partials); gloo carries the 144 / 96-byte partials.  Works on a 1-GPU box (both ranks on
    device 0) and on a multi-GPU one (rank r on device r)."""
    _run_gpu_world(curve, 3001, 2, native=False)


@pytest.mark.gpu
def test_sharded_msm_world2_rccl(gpu):
    """The exchange inside the library (blz_msm_comm_init / blz_msm_all_gather_combine: ncclAllGather over xGMI on
    the handle's stream + rank-ordered add).  RCCL refuses two ranks on one device, so this needs 2 GPUs."""
    import blaze_amd

    ndev = blaze_amd.lib().blz_device_count()
    if ndev < 2:
        pytest.skip(f"NEEDS 2 GPUs: only {ndev} visible - the RCCL world-2 exchange was NOT exercised "
                    "(test_native_exchange_single_rank covers the library path with one rank)")
    _run_gpu_world("BLS381", 3001, 2, native=True)


@pytest.mark.gpu
def test_native_exchange_single_rank(gpu, orc):
    """blz_comm_unique_id / blz_msm_comm_init / blz_msm_all_gather_combine / blz_msm_comm_free with a
    one-rank communicator: RCCL is found at run time, the all-gather runs on the handle's stream, the combine
    reads the receive buffer on the device, and a partial in general projective form comes back normalised."""
    from gpu_util import msm_client, run_msm

    for curve in ("BLS381", "BN254"):
        n = 500
        pts, sc, exp = orc.input_generator(curve, n, 1, 99)
        cl = msm_client(curve, 1)
        part = run_msm(cl, pts, sc, n)
        cl.comm_init(0, 1, cl.comm_unique_id())
        assert cl.all_gather_combine(part) == exp
        assert cl.all_gather_combine(part) == exp          # the communicator is reusable
        with pytest.raises(Exception):
            cl.comm_init(0, 1, cl.comm_unique_id())        # one communicator per handle
        cl.comm_free()
        with pytest.raises(Exception):
            cl.all_gather_combine(part)
        cl.close()


@pytest.mark.gpu
def test_single_process_group_exchange(gpu, orc):
    """blz_msm_comm_init_all / blz_msm_all_gather_combine_all: one host thread, one handle per device, every rank of
    the communicator brought up inside one ncclGroupStart/End (the per-rank calls are blocking rendezvous and would
    wait for each other when issued in sequence from one thread).  Runs with as many ranks as the box has GPUs
    (1 on the single-GPU box: the group path with n = 1)."""
    import blaze_amd
    from blaze_amd.ingo_msm import MSMClient
    from gpu_util import msm_client, run_msm

    ndev = min(8, blaze_amd.lib().blz_device_count())
    curve, n = "BLS381", 900
    pts, sc, exp = orc.input_generator(curve, n, 1, 123)
    pb = orc.point_bytes(curve)
    from blaze_amd.multi_gpu import shard_range

    clients = [msm_client(curve, 1, device=d) for d in range(ndev)]
    partials = []
    for r, cl in enumerate(clients):
        lo, hi = shard_range(n, r, ndev)
        partials.append(run_msm(cl, bytes(pts[lo * pb: hi * pb]), bytes(sc[lo * 32: hi * 32]), hi - lo))
    MSMClient.comm_init_all(clients)
    for _ in range(2):                                   # the communicators are reusable
        outs = MSMClient.all_gather_combine_all(clients, partials)
        assert all(o == exp for o in outs), [o.hex()[:16] for o in outs]
    with pytest.raises(Exception):
        MSMClient.comm_init_all(clients)                 # one communicator per handle
    for cl in clients:
        cl.comm_free()
        cl.close()


@pytest.mark.gpu
def test_comm_init_has_a_deadline(gpu, monkeypatch):
    """A rank that never arrives: ncclCommInitRank(rank 0 of 2) with no peer must come back with Unknown after
    BLAZE_COMM_TIMEOUT_MS instead of blocking for ever.  Runs in a child process, because the
    abandoned bring-up thread stays parked inside RCCL until the process exits."""
    import subprocess
    import textwrap

    code = textwrap.dedent("""
        import sys, time
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        from gpu_util import msm_client
        from blaze_amd import DriverClientError
        cl = msm_client("BLS381", 1)
        t0 = time.perf_counter()
        try:
            cl.comm_init(0, 2, cl.comm_unique_id())
        except DriverClientError as e:
            dt = time.perf_counter() - t0
            print("RESULT", e.variant, round(dt, 2), "did not complete within" in str(e), flush=True)
            import os; os._exit(0)
        print("RESULT no-error", flush=True); import os; os._exit(1)
    """) % (ROOT, os.path.join(ROOT, "tests"))
    env = dict(os.environ, BLAZE_COMM_TIMEOUT_MS="3000")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
    assert p.returncode == 0 and line, (p.stdout[-500:], p.stderr[-1500:])
    _, variant, dt, msg_ok = line[-1].split()
    assert variant == "Unknown" and msg_ok == "True" and 2.5 < float(dt) < 30.0, line


def test_shard_range_partitions_exactly():
    from blaze_amd.multi_gpu import shard_range

    for n in (0, 1, 7, 8, 1 << 26, (1 << 26) + 5):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_shard_layout_tiles_elements_and_bits():
    """blz_msm_shard_layout (host-side): for every world size the ranks' (element chunk) x (scalar range) rectangles tile
    [0, n) x [0, 256) exactly - no overlap, no hole - with 32-bit aligned ranges; BLAZE_SHARD=elements is the plain element
    split of shard_range."""
    import os

    from blaze_amd.ingo_msm import Curve
    from blaze_amd.multi_gpu import shard_layout, shard_range

    for curve in (Curve.BLS381, Curve.BN254):
        for n in (1, 7, 1000, (1 << 22) + 5, 1 << 26):
            for world in range(1, 9):
                lays = [shard_layout(curve, n, r, world) for r in range(world)]
                cells = {}
                for l in lays:
                    assert l["bit_lo"] % 32 == 0 and l["bit_hi"] % 32 == 0 and 0 <= l["bit_lo"] < l["bit_hi"] <= 256
                    if l["count"]:   # (more ranks than elements: the surplus chunks are empty)
                        cells.setdefault((l["first"], l["count"]), []).append((l["bit_lo"], l["bit_hi"]))
                chunks = sorted(cells)
                pos = 0
                for first, count in chunks:
                    assert first == pos
                    pos += count
                    rs = sorted(cells[(first, count)])
                    assert rs[0][0] == 0 and rs[-1][1] == 256 and all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
                assert pos == n
    os.environ["BLAZE_SHARD"] = "elements"
    try:
        for world in (2, 8):
            for r in range(world):
                l = shard_layout(Curve.BLS381, 1 << 26, r, world)
                lo, hi = shard_range(1 << 26, r, world)
                assert (l["first"], l["first"] + l["count"], l["bit_lo"], l["bit_hi"]) == (lo, hi, 0, 256)
    finally:
        del os.environ["BLAZE_SHARD"]
    # what the planner picks at the bench size (documented in DESIGN.md section 6): bits for 2 and 4 ranks, a mix for 8
    picks = {w: shard_layout(Curve.BLS381, 1 << 26, 0, w) for w in (2, 4, 8)}
    assert picks[2]["count"] == 1 << 26 and picks[2]["bit_hi"] == 128
    assert picks[4]["count"] == 1 << 26 and picks[4]["bit_hi"] == 64
    assert picks[8]["bit_hi"] < 256
    # blz_msm_shard_layout_ex prices the flow's transfers: R scalar ranges make a rank receive R x the
    # scalar bytes and hold R x the bases.  Everything resident: the picks above.  Scalars from host memory with every task
    # (the reference's HBM flow): four ranks may not take all 2^26 scalars each (2 GiB = 38 ms over the link against a ~32 ms
    # task), and even a hidden upload costs ~15 % of its duration: the element split; bases travelling too (DMA flow): same.
    from blaze_amd.multi_gpu import SHARD_BASES_FROM_HOST, SHARD_SCALARS_FROM_HOST, shard_layout_ex
    for w in (2, 4, 8):
        res = shard_layout_ex(Curve.BLS381, 1 << 26, 0, w, 0)
        assert (res["first"], res["count"], res["bit_lo"], res["bit_hi"]) == tuple(picks[w][k] for k in ("first", "count", "bit_lo", "bit_hi"))
        assert res["est_link_ms"] == 0 and res["device_mib"] == res["count"] * (96 + 128 + 32) // (1 << 20)
        host = shard_layout_ex(Curve.BLS381, 1 << 26, 0, w, SHARD_SCALARS_FROM_HOST)
        assert host["ranges"] == 1            # measured (profiles/r04_shard_layouts.txt): the element split wins at 2, 4 and 8 ranks
        assert abs(host["est_link_ms"] - host["count"] * 32 / 56.3e6) < 0.01
        dma = shard_layout_ex(Curve.BLS381, 1 << 26, 0, w, SHARD_SCALARS_FROM_HOST | SHARD_BASES_FROM_HOST)
        assert dma["ranges"] == 1 and dma["count"] == (1 << 26) // w
        # candidates: same rectangles, R x the bytes
        for R in (1, 2, 4, 8):
            if w % R == 0:
                c = shard_layout_ex(Curve.BLS381, 1 << 26, 0, w, SHARD_SCALARS_FROM_HOST, R)
                assert c["ranges"] == R and c["count"] == (1 << 26) * R // w and c["bit_hi"] == 256 // R
    # every rank of a transfer-aware layout still tiles the job
    for flags in (SHARD_SCALARS_FROM_HOST, SHARD_SCALARS_FROM_HOST | SHARD_BASES_FROM_HOST):
        for w in (2, 3, 4, 6, 8):
            area = 0
            for r in range(w):
                l = shard_layout_ex(Curve.BLS377, (1 << 24) + 3, r, w, flags)
                area += l["count"] * (l["bit_hi"] - l["bit_lo"])
            assert area == ((1 << 24) + 3) * 256
