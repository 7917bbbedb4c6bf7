"""A task fed by SEVERAL set_data calls (include/blaze_hip.h "STREAMED TASKS"; blaze_amd/csrc/msm_stage.hip stage_stream).  The
reference writes its input to the card's FIFOs in 2048-element chunks (src/ingo_msm/msm_api.rs:155-202) and the card counts
elements against NUMBER_OF_MSM_ELEMENTS (msm_hw_code.rs:18-19): however a queued task's bytes are split over calls, it is the
same task.  SURVEY.md 8(b): {armed_n, received}, launch when received == armed_n."""
import random
import time

import numpy as np
import pytest

import blaze_amd
from blaze_amd import DeviceBuffer, DriverClientError
from blaze_amd.ingo_msm import Curve, MSMInput, MSMParams, PointMemoryType
from gpu_util import msm_client, run_msm, synth

pytestmark = pytest.mark.gpu
CURVES = ["BLS377", "BLS381", "BN254"]


def _cuts(rng, n, kind):
    """Slice sizes of one task: a partition of n."""
    if kind == "cadence":            # the reference's own: 2048 elements, the remainder last
        return [2048] * (n // 2048) + ([n % 2048] if n % 2048 else [])
    if kind == "ones":               # single elements, then the rest
        k = min(n - 1, 5)
        return [1] * k + [n - k]
    out, left = [], n
    while left:
        m = min(left, rng.choice([0, 1, 7, 100, 2048, 4097, max(1, n // 3)]))
        out.append(m)
        left -= m
    return out


def _feed(cl, pts, sc, n, pf, ps, cuts, hbm=None, load=False):
    """initialize -> start_process -> one set_data per slice.  pts None: scalars only (bases in the arena at `hbm`); load: every
    slice brings its part of the table to the arena too (msm_api.rs:203-216)."""
    cl.initialize(MSMParams(n, hbm))
    cl.start_process()
    at = 0
    for m in cuts:
        s = sc[32 * at: 32 * (at + m)]
        p = None if pts is None else pts[at * pf * ps: (at + m) * pf * ps]
        addr = hbm
        if load:
            addr = (hbm[0], hbm[1] + at * pf * ps)
        assert cl.stream_progress() == ((at, n) if at else (0, n))
        cl.set_data(MSMInput(p, s, MSMParams(m, addr)))
        at += m
    assert at == n
    assert cl.stream_progress() == (0, 0)          # complete: handed to the device, nothing queued any more


@pytest.mark.parametrize("curve", CURVES)
@pytest.mark.parametrize("pf", [1, 8])
def test_random_slice_partitions_against_the_oracle(gpu, orc, curve, pf, monkeypatch):
    """Three modes x random partitions (zero-length slices, single elements, the reference's 2048 cadence with a ragged tail) x the
    engine's piece counts (whole launch; 3 and 16 pieces forced at sizes the oracle checks): the bytes of the one-call task."""
    rng = random.Random(1000 * pf + len(curve))
    n = 9000 if pf == 1 else 2600
    ps = 64 if curve == "BN254" else 96
    pts, sc, _ = orc.input_generator(curve, n, pf, 4242 + pf)
    raw = np.random.default_rng(n + pf).integers(0, 256, size=32 * n, dtype=np.uint8).reshape(n, 32)
    raw[:, 31] &= 0x0F                            # < 2^252: canonical in all three scalar fields
    sc = raw.tobytes()
    exp = orc.msm_pippenger(curve, pts, sc, n, pf, threads=8)
    blaze_amd._lib.check(blaze_amd.lib().blz_arena_release(0))
    for pieces in (None, "3", "16"):
        if pieces:
            monkeypatch.setenv("BLAZE_MSM_PIECES", pieces)
        dma = msm_client(curve, pf)
        hbm = msm_client(curve, pf, PointMemoryType.HBM)
        assert run_msm(dma, pts, sc, n) == exp
        for kind in ("random", "cadence", "ones", "random"):
            cuts = _cuts(rng, n, kind)
            # (ii) points + scalars in every slice
            _feed(dma, pts, sc, n, pf, ps, cuts)
            dma.wait_result()
            assert dma.result().result == exp, f"{curve} pf={pf} pieces={pieces} dma {kind} {cuts[:8]}"
            # (iii) every slice loads its part of the table, then its scalars
            blaze_amd._lib.check(blaze_amd.lib().blz_arena_release(0))
            _feed(hbm, pts, sc, n, pf, ps, cuts, hbm=(1 << 20, 96), load=True)
            hbm.wait_result()
            assert hbm.result().result == exp, f"{curve} pf={pf} pieces={pieces} load {kind}"
            assert hbm.get_data_from_hbm(len(pts), 1 << 20, 96) == bytes(pts)
            # (i) scalars only over the table that is there now
            _feed(hbm, None, sc, n, pf, ps, _cuts(rng, n, kind), hbm=(1 << 20, 96))
            hbm.wait_result()
            assert hbm.result().result == exp, f"{curve} pf={pf} pieces={pieces} scalars-only {kind}"
        dma.close(); hbm.close()
    blaze_amd._lib.check(blaze_amd.lib().blz_arena_release(0))


def test_stream_state_machine(gpu, orc):
    """Over-feeding is refused and changes nothing; start_process / wait_result with a half-fed task are refused; a slice in another
    mode is refused; a task in flight can be collected while the next one is half-fed; reset drops a half-fed task; a stream after
    a reset is a stream like any other; device pointers stream too."""
    curve, n, ps = "BLS381", 5000, 96
    pts, sc, exp = orc.input_generator(curve, n, 1, 99)
    cl = msm_client(curve, 1)
    p = MSMParams(n, None)
    cl.initialize(p); cl.start_process()
    half = 2048
    cl.set_data(MSMInput(pts[: half * ps], sc[: half * 32], MSMParams(half, None)))
    assert cl.stream_progress() == (half, n)
    with pytest.raises(DriverClientError) as ei:      # more than the task lacks
        cl.set_data(MSMInput(pts[half * ps:] + pts[:ps], sc[half * 32:] + sc[:32], MSMParams(n - half + 1, None)))
    assert ei.value.variant == "InvalidPrimitiveParam" and cl.stream_progress() == (half, n)
    with pytest.raises(DriverClientError) as ei:      # lengths are checked against the slice's own count
        cl.set_data(MSMInput(pts[half * ps:], sc[half * 32: -32], MSMParams(n - half, None)))
    assert ei.value.variant == "InvalidPrimitiveParam" and cl.stream_progress() == (half, n)
    with pytest.raises(DriverClientError) as ei:      # another mode
        cl.set_data(MSMInput(None, sc[half * 32:], MSMParams(n - half, (0, 0))))
    assert ei.value.variant == "InvalidPrimitiveParam" and cl.stream_progress() == (half, n)
    with pytest.raises(DriverClientError) as ei:
        cl.start_process()
    assert ei.value.variant == "InvalidPrimitiveParam" and "2048 of its 5000" in str(ei.value)
    with pytest.raises(DriverClientError) as ei:
        cl.wait_result()
    assert ei.value.variant == "InvalidPrimitiveParam" and "2048 of its 5000" in str(ei.value)
    cl.set_data(MSMInput(None, b"", MSMParams(0, None)))          # (None, None): the reference's silent no-op
    cl.set_data(MSMInput(pts[half * ps:], sc[half * 32:], MSMParams(n - half, None)))
    cl.wait_result()
    r = cl.result()
    assert r.result == exp and r.result_label == cl.task_label() == 1
    # a whole task in flight, the next one half-fed: the first can be collected, the second completes afterwards
    cl.initialize(p); cl.start_process(); cl.set_data(MSMInput(pts, sc, p))
    cl.initialize(p); cl.start_process()
    cl.set_data(MSMInput(pts[: 100 * ps], sc[: 3200], MSMParams(100, None)))
    cl.wait_result()
    assert cl.result().result == exp
    cl.set_data(MSMInput(pts[100 * ps:], sc[3200:], MSMParams(n - 100, None)))
    cl.wait_result()
    r = cl.result()
    assert r.result == exp and r.result_label == 3
    # reset drops a half-fed task
    cl.initialize(p); cl.start_process()
    cl.set_data(MSMInput(pts[: 100 * ps], sc[: 3200], MSMParams(100, None)))
    cl.reset()
    assert cl.stream_progress() == (0, 0)
    with pytest.raises(DriverClientError):
        cl.wait_result()
    assert run_msm(cl, pts, sc, n) == exp
    _feed(cl, pts, sc, n, 1, ps, [1, 4999])
    cl.wait_result()
    assert cl.result().result == exp
    # a slice without a queued task is a whole task of its own size, as ever (README order: set_data before start_process)
    cl.initialize(MSMParams(100, None))
    cl.set_data(MSMInput(pts[: 100 * ps], sc[: 3200], MSMParams(100, None)))
    cl.start_process(); cl.wait_result()
    assert cl.result().result == orc.msm_naive(curve, pts[: 100 * ps], sc[: 3200], 100, 1)
    # device pointers: slices are copied into the staging set
    dp = DeviceBuffer(0, n * ps); ds = DeviceBuffer(0, n * 32)
    dp.upload(pts); ds.upload(sc)
    cl.initialize(p); cl.start_process()
    at = 0
    for m in (1000, 1, 3999):
        vp = DeviceBuffer.__new__(DeviceBuffer); vp.device_id = 0; vp.ptr = dp.ptr + at * ps; vp.nbytes = m * ps
        vs = DeviceBuffer.__new__(DeviceBuffer); vs.device_id = 0; vs.ptr = ds.ptr + at * 32; vs.nbytes = m * 32
        cl.set_data(MSMInput(vp, vs, MSMParams(m, None)))
        vp.ptr = None; vs.ptr = None
        at += m
    dp.free(); ds.free()
    cl.wait_result()
    assert cl.result().result == exp
    cl.close()


def test_streamed_precompute_plan_and_ranges(gpu, orc, monkeypatch):
    """Scalars-only streams on a precompute handle that opted in to the checked-table plan (4n even bases, 64-bit chunks: a slice of
    m elements is 4m points of the pieces) and on a handle with a scalar range: the bytes of the one-call task."""
    monkeypatch.setenv("BLAZE_MSM_PIECES", "5")
    rng = random.Random(5)
    for curve in ("BN254", "BLS381"):
        n, pf = 3000, 8
        ps = 64 if curve == "BN254" else 96
        pts, sc, exp = orc.input_generator(curve, n, pf, 31)
        blaze_amd._lib.check(blaze_amd.lib().blz_arena_release(0))
        cl = msm_client(curve, pf, PointMemoryType.HBM)
        cl.set_precompute_plan(True)
        cl.load_data_to_hbm(pts, 0, 0)
        assert run_msm(cl, None, sc, n, hbm=(0, 0)) == exp and cl.precompute_plan_info()["used"]
        for kind in ("random", "cadence"):
            _feed(cl, None, sc, n, pf, ps, _cuts(rng, n, kind), hbm=(0, 0))
            cl.wait_result()
            assert cl.result().result == exp and cl.precompute_plan_info()["used"], f"{curve} {kind}"
        # a table rewritten (to the same bytes) under a half-fed plan task: the stream is refused, the task can be sent again
        cl.initialize(MSMParams(n, (0, 0))); cl.start_process()
        cl.set_data(MSMInput(None, sc[: 32 * 2000], MSMParams(2000, (0, 0))))
        cl.load_data_to_hbm(pts[: 8 * ps], 0, 0)
        with pytest.raises(DriverClientError) as ei:
            cl.set_data(MSMInput(None, sc[32 * 2000:], MSMParams(n - 2000, (0, 0))))
        assert ei.value.variant == "InvalidPrimitiveParam" and cl.stream_progress() == (0, n)
        _ = [cl.set_data(MSMInput(None, sc[32 * a: 32 * b], MSMParams(b - a, (0, 0)))) for a, b in ((0, 1234), (1234, n))]
        cl.wait_result()
        assert cl.result().result == exp
        cl.close()
    curve, n = "BLS377", 6000
    pts, sc, _ = orc.input_generator(curve, n, 1, 32)
    blaze_amd._lib.check(blaze_amd.lib().blz_arena_release(0))
    cl = msm_client(curve, 1, PointMemoryType.HBM)
    cl.load_data_to_hbm(pts, 0, 0)
    cl.set_scalar_range(64, 192)
    exp = run_msm(cl, None, sc, n, hbm=(0, 0))
    _feed(cl, None, sc, n, 1, 96, _cuts(rng, n, "random"), hbm=(0, 0))
    cl.wait_result()
    assert cl.result().result == exp
    cl.close()
    blaze_amd._lib.check(blaze_amd.lib().blz_arena_release(0))


def test_streamed_2e22_equals_one_call(gpu, orc):
    """Config 2's shape (2^22 BLS12-381, DMA mode, host buffers) in 64 slices of 2^16 elements and at the reference's 2048-element
    cadence: the bytes of the one-call task, which the oracle checks by linearity."""
    curve, n = "BLS381", 1 << 22
    dp, ds = synth(curve, n, seed=22)
    pts = np.frombuffer(dp.download(), dtype=np.uint8)
    sc = np.frombuffer(ds.download(), dtype=np.uint8)
    dp.free(); ds.free()
    k = orc.index_weighted_sum(curve, sc, n, 0, threads=16)
    exp = orc.result_from_affine(curve, orc.generator_mul(curve, k))
    cl = msm_client(curve, 1)
    assert run_msm(cl, pts, sc, n) == exp
    for step in (1 << 16, 2048):
        cl.initialize(MSMParams(n, None)); cl.start_process()
        t0 = time.perf_counter()
        for a in range(0, n, step):
            cl.set_data(MSMInput(pts[a * 96: (a + step) * 96], sc[a * 32: (a + step) * 32], MSMParams(step, None)))
        t1 = time.perf_counter()
        cl.wait_result()
        t2 = time.perf_counter()
        assert cl.result().result == exp, step
        print(f"[stream 2^22, slices of {step}] set_data {1e3 * (t1 - t0):.1f} ms, wait_result {1e3 * (t2 - t1):.1f} ms")
    cl.close()


def test_reference_max_shape_2e26_precompute_dma_streamed(gpu, orc):
    """msm_bls12_381_precompute_max_test (tests/integration_msm.rs:386-467) in its OWN mode - PointMemoryType::DMA, points: Some -
    at its own size: 2^26 elements x PRECOMPUTE_FACTOR 8 = 2^29 bases, 48 GiB.  The reference hands set_data one 48 GiB Vec;
    here the task is queued once and fed from 3 GiB host slices (2^22 elements each: 16 set_data calls).  Bytes equal the
    oracle's (linearity over B_ij = 2^(32 j) (i + 1) G) and the arena path's (bases resident, scalars-only set_data)."""
    curve, n, pf, ps = "BLS381", 1 << 26, 8, 96
    blaze_amd._lib.check(blaze_amd.lib().blz_arena_release(0))
    dp, ds = synth(curve, n, pf=pf, seed=0x2626)
    sc = np.frombuffer(ds.download(), dtype=np.uint8)
    k = orc.index_weighted_sum(curve, sc, n, 0, threads=16)
    exp = orc.result_from_affine(curve, orc.generator_mul(curve, k))
    cl = msm_client(curve, pf)
    step = 1 << 22
    cl.initialize(MSMParams(n, None)); cl.start_process()
    t_set = 0.0
    for a in range(0, n, step):
        host_pts = np.frombuffer(dp.download(step * pf * ps, a * pf * ps), dtype=np.uint8)     # 3 GiB
        t0 = time.perf_counter()
        cl.set_data(MSMInput(host_pts, sc[a * 32: (a + step) * 32], MSMParams(step, None)))
        t_set += time.perf_counter() - t0
        del host_pts
    t0 = time.perf_counter()
    cl.wait_result()
    t_wait = time.perf_counter() - t0
    got = cl.result().result
    print(f"[2^26 pf=8 DMA mode, 16 slices of 3 GiB] dur_set_data {1e3 * t_set:.1f} ms, dur_wait_result {1e3 * t_wait:.1f} ms")
    assert got == exp
    cl.close()
    # the arena path over the same table
    hb = msm_client(curve, pf, PointMemoryType.HBM)
    hb.load_data_to_hbm(dp, 0, 0)
    dp.free()
    assert run_msm(hb, None, ds, n, hbm=(0, 0)) == got
    hb.close(); ds.free()
    blaze_amd._lib.check(blaze_amd.lib().blz_arena_release(0))
