"""Sharding by scalar chunk (include/blaze_hip.h blz_msm_set_scalar_range / blz_msm_shard_layout): a handle with a scalar
range returns 2^bit_lo x the MSM over bits [bit_lo, bit_hi) of the scalars; the partials of a partition of [0, 256) - and
of the layouts the library picks for 2, 4 and 8 ranks - add up to the full result, byte for byte."""
import ctypes as C

import pytest

import blaze_amd
from blaze_amd import DeviceBuffer, DriverClientError
from blaze_amd.ingo_msm import Curve
from blaze_amd.multi_gpu import shard_layout
from gpu_util import msm_client, run_msm, synth
from oracle import pyref

pytestmark = pytest.mark.gpu
CURVES = ["BLS377", "BLS381", "BN254"]


def _masked(sc: bytes, n: int, lo: int, hi: int) -> bytes:
    """Every scalar reduced to its bits [lo, hi), left in place: the integer the range handle sums."""
    out = bytearray(len(sc))
    mask = ((1 << (hi - lo)) - 1) << lo
    for i in range(n):
        v = int.from_bytes(sc[32 * i: 32 * i + 32], "little") & mask
        out[32 * i: 32 * i + 32] = v.to_bytes(32, "little")
    return bytes(out)


@pytest.mark.parametrize("curve", CURVES)
def test_range_partials_against_the_oracle(gpu, orc, curve):
    """Every range's partial == the oracle's double-and-add over the masked scalars (non-canonical scalars included: the
    top range then holds bits above r), and the partials of 2, 4 and 8 ranges combine to the full result."""
    import random
    rng = random.Random(91)
    n = 700
    pts, sc, exp = orc.input_generator(curve, n, 1, 8100)
    sc = bytearray(sc)
    for i, v in enumerate([(1 << 256) - 1, 1 << 255, (1 << 64) - 1, 1 << 64, (1 << 128) + (1 << 127), 0, 1]):
        sc[32 * i: 32 * i + 32] = v.to_bytes(32, "little")
    for i in range(20, n, 7):
        sc[32 * i: 32 * i + 32] = rng.getrandbits(256).to_bytes(32, "little")
    sc = bytes(sc)
    full = orc.msm_naive(curve, pts, sc, n, 1)
    cl = msm_client(curve, 1)
    assert run_msm(cl, pts, sc, n) == full
    for R in (2, 4, 8):
        parts = b""
        for r in range(R):
            lo, hi = r * 256 // R, (r + 1) * 256 // R
            cl.set_scalar_range(lo, hi)
            got = run_msm(cl, pts, sc, n)
            assert got == orc.msm_naive(curve, pts, _masked(sc, n, lo, hi), n, 1), f"{curve} range [{lo}, {hi})"
            parts += got
        assert cl.combine_partials(parts, R) == full, f"{curve} R={R}"
    cl.set_scalar_range(0, 256)
    assert run_msm(cl, pts, sc, n) == full
    # uneven partitions are fine too (any 32-bit aligned cut)
    parts = b""
    for lo, hi in ((0, 96), (96, 224), (224, 256)):
        cl.set_scalar_range(lo, hi)
        parts += run_msm(cl, pts, sc, n)
    assert cl.combine_partials(parts, 3) == full
    for bad in ((0, 100), (32, 32), (64, 32), (0, 288)):
        with pytest.raises(DriverClientError):
            cl.set_scalar_range(*bad)
    cl.close()
    pf8 = msm_client(curve, 8)
    with pytest.raises(DriverClientError):
        pf8.set_scalar_range(0, 64)
    pf8.close()


@pytest.mark.parametrize("curve,logn", [("BLS381", 21), ("BLS377", 20), ("BN254", 20)])
def test_library_layouts_combine_to_the_full_result(gpu, orc, curve, logn):
    """The layouts blz_msm_shard_layout picks for 2, 4, 8 (and 3, 6) ranks, every rank's task run on this one GPU - two
    in flight, so the hidden sort runs on range tasks too - and the partials combined in rank order: the full MSM
    (expected value by linearity over P_i = (i + 1) G)."""
    n = (1 << logn) - 777
    dp, ds = synth(curve, n, seed=31)
    k = orc.index_weighted_sum(curve, ds.download(), n, 0, threads=8)
    exp = orc.result_from_affine(curve, orc.generator_mul(curve, k))
    ps = dp.nbytes // n
    cl = msm_client(curve, 1)
    from blaze_amd.ingo_msm import MSMInput, MSMParams
    for world in (2, 4, 8, 3, 6):
        lays = [shard_layout(Curve[curve], n, r, world) for r in range(world)]
        # a partition: element chunks x bit ranges tile [0, n) x [0, 256)
        area = sum(l["count"] * (l["bit_hi"] - l["bit_lo"]) for l in lays)
        assert area == n * 256, lays
        parts, pending = [], 0
        for l in lays:
            vp = DeviceBuffer.__new__(DeviceBuffer); vp.device_id = 0; vp.ptr = dp.ptr + l["first"] * ps; vp.nbytes = l["count"] * ps
            vs = DeviceBuffer.__new__(DeviceBuffer); vs.device_id = 0; vs.ptr = ds.ptr + l["first"] * 32; vs.nbytes = l["count"] * 32
            cl.set_scalar_range(l["bit_lo"], l["bit_hi"])
            params = MSMParams(l["count"], None)
            cl.initialize(params); cl.start_process(); cl.set_data(MSMInput(vp, vs, params))
            vp.ptr = None; vs.ptr = None
            pending += 1
            if pending == 2:
                cl.wait_result(); parts.append(cl.result().result); pending -= 1
        while pending:
            cl.wait_result(); parts.append(cl.result().result); pending -= 1
        assert cl.combine_partials(b"".join(parts), world) == exp, f"{curve} world={world} {lays}"
    cl.close(); dp.free(); ds.free()


@pytest.mark.parametrize("curve,logn", [("BLS381", 20), ("BN254", 19)])
def test_layouts_with_window_tables(gpu, orc, curve, logn):
    """The same layouts on handles that opted in to the resident-base window table: a ranged handle's table holds
    2^(lo + c j) P - the range's weight sits in the points, the task runs hi - lo bits' worth of windows into one bucket set
    and closes without doublings.  Partials combined in rank order = the full MSM; every partial = the plain ranged partial."""
    from blaze_amd.ingo_msm import PointMemoryType
    n = (1 << logn) - 333
    dp, ds = synth(curve, n, seed=41)
    k = orc.index_weighted_sum(curve, ds.download(), n, 0, threads=8)
    exp = orc.result_from_affine(curve, orc.generator_mul(curve, k))
    ps = dp.nbytes // n
    blaze_amd._lib.check(blaze_amd.lib().blz_arena_release(0))
    tab = msm_client(curve, 1, PointMemoryType.HBM)
    tab.set_window_table(2)
    tab.load_data_to_hbm(dp, 0, 0)
    plain = msm_client(curve, 1)
    for world in (2, 4, 8):
        parts = b""
        for r in range(world):
            l = shard_layout(Curve[curve], n, r, world)
            vs = DeviceBuffer.__new__(DeviceBuffer); vs.device_id = 0; vs.ptr = ds.ptr + l["first"] * 32; vs.nbytes = l["count"] * 32
            vp = DeviceBuffer.__new__(DeviceBuffer); vp.device_id = 0; vp.ptr = dp.ptr + l["first"] * ps; vp.nbytes = l["count"] * ps
            tab.set_scalar_range(l["bit_lo"], l["bit_hi"])
            assert tab.prepare_window_table(l["count"], (0, l["first"] * ps))     # this range's own table beside the others'
            got = run_msm(tab, None, vs, l["count"], hbm=(0, l["first"] * ps))
            info = tab.window_table_info()
            assert info["bytes"] > 0 and info["windows"] * info["window_bits"] >= l["bit_hi"] - l["bit_lo"] + 1, (l, info)
            assert info["windows"] < 17 or l["bit_hi"] - l["bit_lo"] == 256, (l, info)    # fewer windows than a whole scalar needs
            plain.set_scalar_range(l["bit_lo"], l["bit_hi"])
            assert got == run_msm(plain, vp, vs, l["count"]), f"{curve} world={world} rank={r} {l}"
            vs.ptr = None; vp.ptr = None
            parts += got
        assert tab.combine_partials(parts, world) == exp, f"{curve} world={world}"
    tab.close(); plain.close(); dp.free(); ds.free()
    blaze_amd._lib.check(blaze_amd.lib().blz_arena_release(0))


@pytest.mark.parametrize("curve", ["BLS381", "BLS377"])
def test_auto_layouts_full_size_all_ranks(gpu, orc, curve):
    """What `bench.py --gpus N` executes, at its real size, on this one GPU: for 2, 4 and 8 ranks EVERY rank's task of a 2^26 job
    as blz_msm_shard_layout_ex(flags = 0) cuts it (2 and 4 ranks split the scalars' bits of all elements, 8 ranks take 64-bit
    ranges of half of the elements: the top range with the carry window, 22-bit windows, the three-level sort on 64 / 128-bit
    widths) and as BLAZE_SHARD=elements cuts it - bases in the arena, two tasks in flight.  Each partial is checked against
    the oracle on its own (linearity over P_i = (i + 1) G with the scalars masked to the rank's bit range, as bench.py's config 4
    leg does for rank 0), and the rank-ordered combine_partials against the full result."""
    import os

    import numpy as np

    from blaze_amd.ingo_msm import MSMInput, MSMParams, PointMemoryType
    from blaze_amd.multi_gpu import shard_layout_ex

    n = 1 << 26
    dp, ds = synth(curve, n, seed=0x26)
    blaze_amd._lib.check(blaze_amd.lib().blz_arena_release(0))
    cl = msm_client(curve, 1, PointMemoryType.HBM)
    cl.load_data_to_hbm(dp, 0, 0)
    dp.free()
    sc = np.frombuffer(ds.download(), dtype=np.uint8).reshape(n, 32)

    def expect(first, count, lo, hi):
        m = np.zeros((count, 32), dtype=np.uint8)
        m[:, lo // 8: hi // 8] = sc[first: first + count, lo // 8: hi // 8]
        k = orc.index_weighted_sum(curve, m, count, first, threads=16)
        return orc.result_from_affine(curve, orc.generator_mul(curve, k))

    full = expect(0, n, 0, 256)
    seen = {}   # (first, count, lo, hi) -> checked partial: the element split's halves and quarters recur across worlds
    saved = os.environ.get("BLAZE_SHARD")
    try:
        for mode in ("auto", "elements"):
            if mode == "elements":
                os.environ["BLAZE_SHARD"] = "elements"
            else:
                os.environ.pop("BLAZE_SHARD", None)
            for world in (2, 4, 8):
                lays = [shard_layout_ex(Curve[curve], n, r, world, 0) for r in range(world)]
                assert sum(l["count"] * (l["bit_hi"] - l["bit_lo"]) for l in lays) == n * 256, lays
                if mode == "elements":
                    assert all((l["bit_lo"], l["bit_hi"]) == (0, 256) for l in lays), lays
                elif world == 8:
                    assert {l["bit_hi"] - l["bit_lo"] for l in lays} == {64} and {l["count"] for l in lays} == {n // 2}, lays
                parts, pending = [], 0

                def collect():
                    cl.wait_result()
                    parts.append(cl.result().result)

                for l in lays:
                    cl.set_scalar_range(l["bit_lo"], l["bit_hi"])
                    vs = DeviceBuffer.__new__(DeviceBuffer); vs.device_id = 0; vs.ptr = ds.ptr + l["first"] * 32; vs.nbytes = l["count"] * 32
                    params = MSMParams(l["count"], (0, l["first"] * 96))
                    cl.initialize(params); cl.start_process(); cl.set_data(MSMInput(None, vs, params))
                    vs.ptr = None
                    pending += 1
                    if pending == 2:
                        collect(); pending -= 1
                while pending:
                    collect(); pending -= 1
                for r, (l, got) in enumerate(zip(lays, parts)):
                    key = (l["first"], l["count"], l["bit_lo"], l["bit_hi"])
                    if key not in seen:
                        seen[key] = expect(*key)
                    assert got == seen[key], f"{curve} {mode} world={world} rank={r} {l}"
                assert cl.combine_partials(b"".join(parts), world) == full, f"{curve} {mode} world={world}"
    finally:
        if saved is None:
            os.environ.pop("BLAZE_SHARD", None)
        else:
            os.environ["BLAZE_SHARD"] = saved
    cl.close(); ds.free()
    blaze_amd._lib.check(blaze_amd.lib().blz_arena_release(0))
