"""Resident-base window table (include/blaze_hip.h blz_msm_set_window_table): a pf = 1 handle whose bases live in the
arena tabulates their window multiples once and then adds every window's digit into one bucket set.  The bytes of a
result must not depend on it: every case below is checked against the CPU oracle (or, at sizes the oracle cannot
reach, through linearity over P_i = (i + 1) G) AND against the plain path."""
import os

import pytest

import blaze_amd
from blaze_amd import DeviceBuffer
from blaze_amd.ingo_msm import Curve, MSMInput, MSMParams, PointMemoryType
from gpu_util import msm_client, run_msm, synth
from oracle import pyref

pytestmark = pytest.mark.gpu
CURVES = ["BLS377", "BLS381", "BN254"]


def _release():
    blaze_amd._lib.check(blaze_amd.lib().blz_arena_release(0))


def _expected(orc, curve, ds, n):
    # P_i = (i + 1) G: the MSM is (sum s_i (i + 1)) G
    k = orc.index_weighted_sum(curve, ds.download(), n, 0, threads=8)
    return orc.result_from_affine(curve, orc.generator_mul(curve, k))


def _table_client(curve):
    cl = msm_client(curve, 1, PointMemoryType.HBM)
    cl.set_window_table(2)     # always (mode 1 leaves BN254, which does not gain from a table, on the plain path)
    return cl


@pytest.mark.parametrize("curve", CURVES)
def test_small_sizes_against_the_oracle(gpu, orc, curve):
    """The reference harness's sizes (tests/integration_msm.rs: 2, the 256-tile boundaries, 8192): repeated tiles put
    equal points into one bucket, so the doubling and cancellation branches run on table entries too."""
    _release()
    cl = _table_client(curve)
    ps = pyref.CURVES[curve]["fq_bytes"] * 2
    for k, n in enumerate((1, 2, 255, 256, 257, 1024, 8192)):
        pts, sc, exp = orc.input_generator(curve, n, 1, 300 + n)
        addr = 0x100000 * (k + 1)
        cl.load_data_to_hbm(pts, addr, 0)
        assert cl.prepare_window_table(n, (addr, 0))          # the build runs beside the tasks; here it is waited for
        assert run_msm(cl, None, sc, n, hbm=(addr, 0)) == exp, f"{curve} n={n}"
        info = cl.window_table_info()
        assert info["windows"] * info["window_bits"] >= 257 and info["bytes"] >= n * info["windows"] * ps, info
        # the same bases again (the table is reused), other scalars
        _, sc2, _ = orc.input_generator(curve, n, 1, 900 + n)
        assert run_msm(cl, None, sc2, n, hbm=(addr, 0)) == orc.msm_pippenger(curve, pts, sc2, n, 1, threads=4)
    cl.close()
    _release()


@pytest.mark.parametrize("curve", CURVES)
def test_non_canonical_scalars(gpu, orc, curve):
    """Scalars above r, up to 2^256 - 1: the W windows cover 257 bits, the top one absorbs the last digit carry."""
    import random
    rng = random.Random(78)
    n = 600
    pts, sc, _ = orc.input_generator(curve, n, 1, 4243)
    sc = bytearray(sc)
    r = pyref.CURVES[curve]["r"]
    special = [(1 << 256) - 1, (1 << 256) - 2, 1 << 255, (1 << 255) - 1, r, r + 1, r - 1, 0, 1]
    for i in range(n):
        if i < len(special):
            v = special[i]
        elif i % 3 == 0:
            v = rng.getrandbits(256)
        else:
            continue
        sc[32 * i: 32 * i + 32] = v.to_bytes(32, "little")
    exp = orc.msm_naive(curve, pts, bytes(sc), n, 1)
    _release()
    cl = _table_client(curve)
    cl.load_data_to_hbm(pts, 0, 0)
    assert cl.prepare_window_table(n)
    assert run_msm(cl, None, bytes(sc), n, hbm=(0, 0)) == exp
    assert cl.window_table_info()["bytes"] > 0
    cl.close()
    _release()


def test_table_follows_the_arena(gpu, orc):
    """A write into the extent drops the table (the next task rebuilds it from the new bytes); a sub-range of tabulated
    bases is served from the same table; another handle of the same curve shares it; a handle that did not opt in
    never sees one."""
    curve, n = "BLS381", 4096
    _release()
    pts, sc, exp = orc.input_generator(curve, n, 1, 51)
    cl = _table_client(curve)
    cl.load_data_to_hbm(pts, 0x4000, 0)
    # never inside a task: the first task enqueues the build and takes the plain path, a later one adopts the table
    assert run_msm(cl, None, sc, n, hbm=(0x4000, 0)) == exp
    assert cl.prepare_window_table(n, (0x4000, 0))
    assert run_msm(cl, None, sc, n, hbm=(0x4000, 0)) == exp
    first_build = cl.window_table_info()
    assert first_build["bytes"] > 0
    # sub-range: 1000 bases starting at base 300 (same window width at these sizes)
    sub = run_msm(cl, None, sc[: 32 * 1000], 1000, hbm=(0x4000, 96 * 300))
    assert sub == orc.msm_pippenger(curve, pts[96 * 300: 96 * 1300], sc[: 32 * 1000], 1000, 1, threads=4)
    assert cl.window_table_info()["bytes"] == first_build["bytes"]
    # overwrite 64 points in the middle: results follow the new bytes
    pts2, _, _ = orc.input_generator(curve, 64, 1, 52)
    cl.load_data_to_hbm(pts2, 0x4000, 96 * 512)
    newpts = bytes(pts[: 96 * 512]) + bytes(pts2) + bytes(pts[96 * 576:])
    assert cl.get_data_from_hbm(len(newpts), 0x4000, 0) == newpts
    # (a small rewrite keeps the table: the rows of the 64 rewritten bases are re-tabulated ahead of the next task)
    assert run_msm(cl, None, sc, n, hbm=(0x4000, 0)) == orc.msm_pippenger(curve, newpts, sc, n, 1, threads=4)
    assert cl.window_table_info()["bytes"] == first_build["bytes"]
    assert cl.prepare_window_table(n, (0x4000, 0))
    assert run_msm(cl, None, sc, n, hbm=(0x4000, 0)) == orc.msm_pippenger(curve, newpts, sc, n, 1, threads=4)
    assert cl.window_table_info()["bytes"] > 0
    # a second opted-in handle shares the table; a plain handle gives the same bytes without one
    cl2 = _table_client(curve)
    assert run_msm(cl2, None, sc, n, hbm=(0x4000, 0)) == orc.msm_pippenger(curve, newpts, sc, n, 1, threads=4)
    assert cl2.window_table_info()["bytes"] > 0
    plain = msm_client(curve, 1, PointMemoryType.HBM)
    assert run_msm(plain, None, sc, n, hbm=(0x4000, 0)) == orc.msm_pippenger(curve, newpts, sc, n, 1, threads=4)
    assert plain.window_table_info()["bytes"] == 0
    # a DMA-typed handle that opted in: tables for arena bases (msm_api.rs:41 lets it select them), none for host points
    dma = msm_client(curve, 1, PointMemoryType.DMA)
    dma.set_window_table(1)
    assert run_msm(dma, None, sc, n, hbm=(0x4000, 0)) == orc.msm_pippenger(curve, newpts, sc, n, 1, threads=4)
    assert dma.window_table_info()["bytes"] > 0
    assert run_msm(dma, pts, sc, n) == exp
    assert dma.window_table_info()["bytes"] == 0
    for c in (cl, cl2, plain, dma):
        c.close()
    _release()


def test_small_rewrites_patch_the_table_large_ones_drop_it(gpu, orc):
    """A rewrite of up to 2^18 bases has the rows of those bases re-tabulated ahead of the next task (the table stays, the result
    follows the new bytes at once); a larger one drops the table and the tasks that follow rebuild it, paced as ever.  Two
    handles share the table; a rewritten base of even order (BLS12-377's (-1, 0)... here: BLS12-381 has none, so a point at
    which the table cannot be patched is not testable on this curve) is covered by test_base_of_even_order_falls_back."""
    curve, n = "BLS381", 300000            # 27.5 MiB of bases: more than the 16 MiB a patch may cover
    _release()
    dp, ds = synth(curve, n, seed=71)
    pts, sc = bytearray(dp.download()), bytes(ds.download())
    cl, cl2 = _table_client(curve), _table_client(curve)
    cl.load_data_to_hbm(dp, 0, 0)
    assert cl.prepare_window_table(n, (0, 0))
    exp = run_msm(cl, None, ds, n, hbm=(0, 0))
    built = cl.window_table_info()
    assert built["bytes"] > 0 and exp == _expected(orc, curve, ds, n)
    # 1000 bases rewritten with OTHER bases (elements 5000.. take the points of elements 100000..)
    blk = bytes(pts[96 * 100000: 96 * 101000])
    cl.load_data_to_hbm(blk, 0, 96 * 5000)
    pts[96 * 5000: 96 * 6000] = blk
    want = orc.msm_pippenger(curve, bytes(pts), sc, n, 1, threads=os.cpu_count() or 8)
    assert run_msm(cl2, None, ds, n, hbm=(0, 0)) == want                # the OTHER handle launches first: it patches
    assert cl2.window_table_info()["bytes"] == built["bytes"]
    assert run_msm(cl, None, ds, n, hbm=(0, 0)) == want
    assert cl.window_table_info()["bytes"] == built["bytes"]
    # two small rewrites before the next task: one span
    blk2 = bytes(pts[96 * 200000: 96 * 200010])
    cl.load_data_to_hbm(blk2, 0, 96 * 10)
    cl.load_data_to_hbm(blk2, 0, 96 * 7000)
    pts[96 * 10: 96 * 20] = blk2
    pts[96 * 7000: 96 * 7010] = blk2
    want = orc.msm_pippenger(curve, bytes(pts), sc, n, 1, threads=os.cpu_count() or 8)
    assert run_msm(cl, None, ds, n, hbm=(0, 0)) == want and cl.window_table_info()["bytes"] == built["bytes"]
    # everything rewritten (27.5 MiB): the table goes, the next task takes the plain path and starts the rebuild
    cl.load_data_to_hbm(bytes(pts), 0, 0)
    assert run_msm(cl, None, ds, n, hbm=(0, 0)) == want and cl.window_table_info()["bytes"] == 0
    assert cl.prepare_window_table(n, (0, 0))
    assert run_msm(cl, None, ds, n, hbm=(0, 0)) == want and cl.window_table_info()["bytes"] == built["bytes"]
    cl.close(); cl2.close(); dp.free(); ds.free()
    _release()


def test_mode_one_skips_bn254(gpu, orc):
    """Mode 1 = "where it pays": the BLS curves get a table, BN254 (64-byte points, already gather-bound) does not."""
    n = 512
    for curve, expect in (("BLS381", True), ("BN254", False)):
        _release()
        pts, sc, exp = orc.input_generator(curve, n, 1, 71)
        cl = msm_client(curve, 1, PointMemoryType.HBM)
        cl.set_window_table(True)
        cl.load_data_to_hbm(pts, 0, 0)
        assert cl.prepare_window_table(n) == expect
        assert run_msm(cl, None, sc, n, hbm=(0, 0)) == exp
        assert (cl.window_table_info()["bytes"] > 0) == expect, curve
        with pytest.raises(blaze_amd.DriverClientError):
            cl.set_window_table(3)
        cl.close()
    _release()


def test_base_of_even_order_falls_back(gpu, orc):
    """(-1, 0) lies on BLS12-377's curve y^2 = x^3 + 1 and has order 2: its doubling is the point at infinity, which an
    affine table cannot hold.  The build notices, the extent is served by the plain path, the bytes are the plain path's."""
    curve, n = "BLS377", 300
    q = pyref.CURVES[curve]["q"]
    pts, sc, _ = orc.input_generator(curve, n, 1, 61)
    pts = bytearray(pts)
    pts[96 * 7: 96 * 8] = (q - 1).to_bytes(48, "little") + bytes(48)
    _release()
    plain = msm_client(curve, 1, PointMemoryType.HBM)
    plain.load_data_to_hbm(bytes(pts), 0, 0)
    want = run_msm(plain, None, sc, n, hbm=(0, 0))
    cl = _table_client(curve)
    assert not cl.prepare_window_table(n)       # built, found wanting, refused
    assert run_msm(cl, None, sc, n, hbm=(0, 0)) == want
    assert cl.window_table_info()["bytes"] == 0
    cl.close(); plain.close()
    _release()


@pytest.mark.parametrize("curve,logn,c", [("BLS381", 19, 0), ("BLS381", 20, 26), ("BLS377", 21, 24), ("BN254", 20, 20), ("BLS381", 22, 0),
                                          ("BLS381", 17, 16), ("BN254", 18, 16)])
def test_tasks_in_flight_and_geometries(gpu, orc, curve, logn, c, monkeypatch):
    """Synthetic bases P_i = (i + 1) G at sizes the sort's real geometry shows up (several level-1 blocks, level-2
    slices, a 17-bit level-1 remainder with forced c = 26; final bins in registers, in the two-pass LDS image - forced
    c = 16 at 2^17: 8.7 K entries per bin, the shape of c = 26 at 2^26 - and beyond it - 2^18: 17 K per bin), two
    tasks in flight so that the second task's sort runs underneath the first one's accumulation; expected value by
    linearity, and equal to the plain path's bytes."""
    n = (1 << logn) - 4321
    if c:
        monkeypatch.setenv("BLAZE_MSM_PLAN", f"table_c={c}")
    dp, ds0 = synth(curve, n, seed=21)
    ds1 = DeviceBuffer(0, n * 32)
    blaze_amd._lib.check(blaze_amd.aux().blz_synth_scalars(0, int(Curve[curve]), ds1.ptr, n, 22))
    exp = []
    for d in (ds0, ds1):
        k = orc.index_weighted_sum(curve, d.download(), n, 0, threads=8)
        exp.append(orc.result_from_affine(curve, orc.generator_mul(curve, k)))
    _release()
    cl = _table_client(curve)
    cl.load_data_to_hbm(dp, 0, 0)
    assert cl.prepare_window_table(n)
    params = MSMParams(n, (0, 0))
    got = []
    order = [0, 1, 1, 0, 1, 0]
    for k, which in enumerate(order):
        cl.initialize(params); cl.start_process(); cl.set_data(MSMInput(None, (ds0, ds1)[which], params))
        if k == 0:
            info = cl.window_table_info()
            assert info["bytes"] > 0 and (not c or info["window_bits"] == c), info
        if k >= 1:
            cl.wait_result(); got.append(cl.result().result)
    cl.wait_result(); got.append(cl.result().result)
    assert got == [exp[w] for w in order], f"{curve} 2^{logn} c={c}"
    cl.close()
    plain = msm_client(curve, 1)
    assert run_msm(plain, dp, ds0, n) == exp[0]
    plain.close()
    for b in (dp, ds0, ds1):
        b.free()
    _release()


def test_hot_buckets(gpu, orc):
    """All scalars equal: every window's entries pile into one bucket (the final level's slow path, units of one
    bucket folded by the tree)."""
    curve, n = "BLS381", 70000
    dp, ds = synth(curve, n, seed=5)
    sc = bytearray(ds.download())
    one = sc[:32]
    for i in range(n):
        sc[32 * i: 32 * i + 32] = one
    k = (int.from_bytes(one, "little") * (n * (n + 1) // 2)) % pyref.CURVES[curve]["r"]
    exp = orc.result_from_affine(curve, orc.generator_mul(curve, k))
    _release()
    cl = _table_client(curve)
    cl.load_data_to_hbm(dp, 0, 0)
    assert cl.prepare_window_table(n)
    assert run_msm(cl, None, bytes(sc), n, hbm=(0, 0)) == exp
    assert cl.window_table_info()["bytes"] > 0
    cl.close(); dp.free(); ds.free()
    _release()


def test_bench_workload_2e26_bls381(gpu, orc):
    """The bench workload with the table: 2^26 BLS12-381 bases, 10 windows of 26 bits; expected value by linearity,
    bytes equal to the plain path's; a stream of four tasks (sorts hidden from the second one on)."""
    curve, n = "BLS381", 1 << 26
    dp, ds = synth(curve, n)
    k = orc.index_weighted_sum(curve, ds.download(), n, 0, threads=8)
    exp = orc.result_from_affine(curve, orc.generator_mul(curve, k))
    _release()
    cl = _table_client(curve)
    cl.load_data_to_hbm(dp, 0, 0)
    params = MSMParams(n, (0, 0))
    # the build (3 s of the chip) never sits inside a task: the first task is submitted straight after the load, takes the
    # plain path - at the plain path's latency, give or take the build's share of the chip - and the stream switches to the
    # table at a task boundary once the build is through
    # (the table's ALLOCATION belongs with the load - prepare without a wait, as bench.py does: an 80 GiB hipMalloc takes
    # 0.3 ms on a clean device and 2.4 - 4.3 s when the driver first has to scrub memory that earlier tests freed, whoever calls it)
    import time
    assert not cl.prepare_window_table(n, (0, 0), 0)
    t0 = time.perf_counter()
    cl.initialize(params); cl.start_process(); cl.set_data(MSMInput(None, ds, params))
    assert cl.window_table_info()["bytes"] == 0
    cl.wait_result(); first = cl.result().result
    first_ms = (time.perf_counter() - t0) * 1e3
    assert first == exp and first_ms < 1500, first_ms          # (plain path ~125 ms; the synchronous build of round 3: 3240)
    assert cl.prepare_window_table(n)
    got = []
    for t in range(4):
        cl.initialize(params); cl.start_process(); cl.set_data(MSMInput(None, ds, params))
        if t == 0:
            info = cl.window_table_info()
            assert info["window_bits"] == 26 and info["windows"] == 10, info
        if t >= 1:
            cl.wait_result(); got.append(cl.result().result)
    cl.wait_result(); got.append(cl.result().result)
    assert got == [exp] * 4
    cl.close(); dp.free(); ds.free()
    _release()
