"""The checked-table plan of precompute handles (include/blaze_hip.h blz_msm_set_precompute_plan) on the MI355X.

The reference's precompute mode (MSMInit.is_precompute, src/ingo_msm/msm_api.rs:39-50) has the caller supply the bases
B_(i,j) = 2^(32 j) P_i (precompute_base_*: tests/msm/mod.rs:360-380) and defines the task as the sum over all 8 n of
them with the 32-bit chunks of the scalars.  The plan sums the 4 n even bases with 64-bit chunks instead - after the device has
checked that the resident table is what precompute_base_* produces.  Every test here holds the result to the oracle's literal
sum over the table AS LOADED (consistent or not) and to the exact path's bytes."""
import ctypes as C
import os

import pytest

import blaze_amd
from blaze_amd.ingo_msm import MSMInput, MSMParams, PointMemoryType
from gpu_util import msm_client, run_msm, synth
from oracle import pyref

pytestmark = pytest.mark.gpu
CURVES = ["BLS377", "BLS381", "BN254"]


def _release():
    blaze_amd._lib.check(blaze_amd.lib().blz_arena_release(0))


def _plan_client(curve):
    cl = msm_client(curve, 8, PointMemoryType.HBM)
    cl.set_precompute_plan(True)
    return cl


def test_golden_precompute_vectors_on_the_plan(gpu):
    """The committed pf = 8 golden vectors (tests/golden/msm_vectors.json, minted by the pure-Python implementation) through the
    HBM flow of a client on the checked-table plan: n = 1, 2, 5 - the plan's smallest tasks."""
    import json

    _release()
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "msm_vectors.json")) as f:
        vecs = [v for v in json.load(f) if v["pf"] == 8]
    assert len(vecs) >= 9
    for v in vecs:
        cl = _plan_client(v["curve"])
        cl.load_data_to_hbm(bytes.fromhex(v["points"]), 0, 0)
        assert run_msm(cl, None, bytes.fromhex(v["scalars"]), v["n"], hbm=(0, 0)) == bytes.fromhex(v["result"]), (v["curve"], v["name"])
        assert cl.precompute_plan_info()["used"]
        cl.close()
        _release()


@pytest.mark.parametrize("curve", CURVES)
def test_plan_matches_the_exact_path_on_the_harness_sizes(gpu, orc, curve):
    """tests/integration_msm.rs sizes through the HBM flow of a precompute client: plan on == plan off == oracle."""
    _release()
    ps = orc.point_bytes(curve)
    exact = msm_client(curve, 8, PointMemoryType.HBM)
    cl = _plan_client(curve)
    for n in (2, 255, 256, 257, 1024, 8192):
        pts, sc, exp = orc.input_generator(curve, n, 8, 300 + n)
        cl.load_data_to_hbm(pts, 0, 0)
        got = run_msm(cl, None, sc, n, hbm=(0, 0))
        info = cl.precompute_plan_info()
        assert info["used"] and info["check"] == "consistent", (curve, n, info)
        assert info["even_copy_bytes"] >= n * 4 * (64 if curve == "BN254" else 128)
        assert got == exp, f"{curve} n={n}"
        assert run_msm(exact, None, sc, n, hbm=(0, 0)) == got
        assert not exact.precompute_plan_info()["used"]
        # a second task over the same bases: no second check (same state, same cost figure)
        assert run_msm(cl, None, sc, n, hbm=(0, 0)) == exp
        assert cl.precompute_plan_info()["check_ms"] == info["check_ms"]
        # a task over a sub-range of the elements that starts on the element grid is served from the same check
        if n >= 256:
            k, m = 64, n - 100
            sub = orc.msm_pippenger(curve, pts[k * 8 * ps:], sc[32 * k:], m, 8, threads=8)
            assert run_msm(cl, None, sc[32 * k: 32 * (k + m)], m, hbm=(0, k * 8 * ps)) == sub
            assert cl.precompute_plan_info()["used"]
    cl.close(); exact.close()
    _release()


@pytest.mark.parametrize("curve", CURVES)
def test_a_corrupted_multiple_keeps_the_exact_path(gpu, orc, curve):
    """One multiple of one element replaced by ANOTHER valid curve point: the table is no longer 2^(32 j) P, the check says so,
    and the result is the oracle's literal sum over the table as loaded - what the 8 n-point task computes."""
    _release()
    n = 3000
    ps = orc.point_bytes(curve)
    pts, sc, exp = orc.input_generator(curve, n, 8, 41)
    bad = bytearray(pts)
    i, j = 1717, 3
    other = bytes(pts[(5 * 8 + 6) * ps: (5 * 8 + 7) * ps])       # B_(5,6): on the curve, in the group, not 2^96 P_1717
    bad[(i * 8 + j) * ps: (i * 8 + j + 1) * ps] = other
    exp_bad = orc.msm_pippenger(curve, bytes(bad), sc, n, 8, threads=8)
    assert exp_bad != exp
    cl = _plan_client(curve)
    cl.load_data_to_hbm(bytes(bad), 0, 0)
    assert run_msm(cl, None, sc, n, hbm=(0, 0)) == exp_bad
    info = cl.precompute_plan_info()
    assert not info["used"] and info["check"] == "refuted", info
    assert not cl.prepare_precompute_plan(n, (0, 0))
    # an ODD multiple - one the plan would never read - counts just the same
    bad2 = bytearray(pts)
    bad2[(9 * 8 + 7) * ps: (9 * 8 + 8) * ps] = other
    cl.load_data_to_hbm(bytes(bad2), 0, 0)
    assert run_msm(cl, None, sc, n, hbm=(0, 0)) == orc.msm_pippenger(curve, bytes(bad2), sc, n, 8, threads=8)
    assert cl.precompute_plan_info()["check"] == "refuted"
    # repairing the one point re-arms the check (any write into the extent does) and the plan comes back
    cl.load_data_to_hbm(pts[(9 * 8 + 7) * ps: (9 * 8 + 8) * ps], 0, (9 * 8 + 7) * ps)
    assert cl.prepare_precompute_plan(n, (0, 0))
    assert run_msm(cl, None, sc, n, hbm=(0, 0)) == exp
    assert cl.precompute_plan_info()["used"]
    # a base that is not on the curve at all (x + 1) - with ITS multiples consistent among themselves or not, the table is
    # refuted: off the curve the addition formulas are not a group law, and only on it do the two sums agree whatever the order
    # of the additions.  (No result to compare: the exact path's own bytes depend on its bucket order there.)
    bad3 = bytearray(pts)
    o = (77 * 8) * ps
    bad3[o] = (bad3[o] + 1) & 0xFF
    cl.load_data_to_hbm(bytes(bad3), 0, 0)
    assert not cl.prepare_precompute_plan(n, (0, 0))
    cl.initialize(MSMParams(n, (0, 0))); cl.start_process(); cl.set_data(MSMInput(None, sc, MSMParams(n, (0, 0)))); cl.wait_result(); cl.result()
    info = cl.precompute_plan_info()
    assert not info["used"] and info["check"] == "refuted"
    cl.close()
    _release()


def test_a_base_of_order_two_refutes_the_table(gpu, orc):
    """(-1, 0) lies on BLS12-377's curve y^2 = x^3 + 1 and has order 2: 2^32 times it is the point at infinity, which no affine
    B_1 equals - the check's chain of doublings runs into Z = 0 and stays there (ec_rr.hip.hpp ptrr_jdbl has no infinity
    branch).  Whatever stands in the element's other slots, the table is refuted and the result is the literal sum."""
    _release()
    curve, n = "BLS377", 700
    ps = orc.point_bytes(curve)
    q = pyref.CURVES[curve]["q"]
    pts, sc, _ = orc.input_generator(curve, n, 8, 77)
    bad = bytearray(pts)
    bad[(333 * 8) * ps: (333 * 8 + 1) * ps] = (q - 1).to_bytes(48, "little") + bytes(48)
    exp = orc.msm_pippenger(curve, bytes(bad), sc, n, 8, threads=8)
    cl = _plan_client(curve)
    cl.load_data_to_hbm(bytes(bad), 0, 0)
    assert not cl.prepare_precompute_plan(n, (0, 0))
    assert run_msm(cl, None, sc, n, hbm=(0, 0)) == exp
    info = cl.precompute_plan_info()
    assert not info["used"] and info["check"] == "refuted", info
    # ... and as the LAST base of an element (the chain that starts from it is never walked; the one that should end in it is)
    bad2 = bytearray(pts)
    bad2[(334 * 8 + 7) * ps: (334 * 8 + 8) * ps] = (q - 1).to_bytes(48, "little") + bytes(48)
    cl.load_data_to_hbm(bytes(bad2), 0, 0)
    assert run_msm(cl, None, sc, n, hbm=(0, 0)) == orc.msm_pippenger(curve, bytes(bad2), sc, n, 8, threads=8)
    assert cl.precompute_plan_info()["check"] == "refuted"
    cl.close()
    _release()


def test_rewrites_recheck_only_what_they_touch(gpu, orc):
    """A write into a table that was found consistent re-arms the check for the elements it touches only (an element's eight
    bases are checked against each other and nothing else): rewriting a few points of a 2^18-element table costs a check of a few
    elements, not of the table; a wrong point among them is still caught; a repair brings the plan back."""
    _release()
    curve, n = "BLS381", 1 << 18
    ps = orc.point_bytes(curve)
    dp, ds = synth(curve, n, pf=8)
    exp = _expected_synth(orc, curve, ds, n)
    cl = _plan_client(curve)
    cl.load_data_to_hbm(dp, 0, 0)
    assert run_msm(cl, None, ds, n, hbm=(0, 0)) == exp
    full = cl.precompute_plan_info()
    assert full["used"] and full["check_ms"] > 5.0                      # 2^18 elements x 224 doublings
    raw = bytes(dp.download(40 * 8 * ps, 1000 * 8 * ps))                # elements 1000 .. 1039 as loaded
    cl.load_data_to_hbm(raw[: 3 * ps], 0, 1000 * 8 * ps)                # the same bytes again: three points of element 1000
    cl.load_data_to_hbm(raw[20 * 8 * ps:], 0, 1020 * 8 * ps)            # ... and elements 1020 .. 1039
    assert run_msm(cl, None, ds, n, hbm=(0, 0)) == exp
    part = cl.precompute_plan_info()
    assert part["used"] and part["check"] == "consistent" and part["check_ms"] < full["check_ms"] / 4
    # a wrong multiple inside a rewritten span: caught by the partial check; the exact path's result is the literal sum
    wrong = raw[5 * ps: 6 * ps]                                         # B_(1000,5) where B_(1000,6) belongs
    cl.load_data_to_hbm(wrong, 0, (1000 * 8 + 6) * ps)
    pts = bytearray(dp.download())
    pts[(1000 * 8 + 6) * ps: (1000 * 8 + 7) * ps] = wrong
    sc = bytes(ds.download())
    got = run_msm(cl, None, ds, n, hbm=(0, 0))
    info = cl.precompute_plan_info()
    assert not info["used"] and info["check"] == "refuted"
    assert got == orc.msm_pippenger(curve, bytes(pts), sc, n, 8, threads=os.cpu_count() or 8)
    cl.load_data_to_hbm(raw[6 * ps: 7 * ps], 0, (1000 * 8 + 6) * ps)    # repaired: checked from scratch (what else was wrong is not known)
    assert run_msm(cl, None, ds, n, hbm=(0, 0)) == exp
    again = cl.precompute_plan_info()
    assert again["used"] and again["check_ms"] > full["check_ms"] / 2
    cl.close(); dp.free(); ds.free()
    _release()


def test_plan_needs_the_element_grid_and_a_precompute_client(gpu, orc):
    _release()
    curve, n = "BLS381", 600
    ps = orc.point_bytes(curve)
    pts, sc, exp = orc.input_generator(curve, n, 8, 43)
    plain = msm_client(curve, 1)
    with pytest.raises(blaze_amd.DriverClientError) as ei:
        plain.set_precompute_plan(True)
    assert ei.value.variant == "InvalidPrimitiveParam"
    plain.close()
    cl = _plan_client(curve)
    # the table sits 3 points into an extent that starts with something else: the task's grid is the extent's point grid
    # shifted by 3 - the check declines (unchecked), the exact path serves it
    cl.load_data_to_hbm(bytes(3 * ps) + bytes(pts), 0, 0)
    assert run_msm(cl, None, sc, n, hbm=(0, 3 * ps)) == exp
    info = cl.precompute_plan_info()
    assert not info["used"] and info["check"] == "unchecked"
    # plan off again: no check at all
    cl.set_precompute_plan(False)
    cl.load_data_to_hbm(pts, 1 << 30, 0)
    assert run_msm(cl, None, sc, n, hbm=(1 << 30, 0)) == exp
    assert cl.precompute_plan_info() == {"used": False, "check": "unchecked", "check_ms": 0.0, "even_copy_bytes": 0}
    cl.close()
    _release()


@pytest.mark.parametrize("curve", CURVES)
def test_plan_with_non_canonical_scalars_and_mixed_windows(gpu, orc, curve, monkeypatch):
    """64-bit chunks of all-ones scalars: every window's digit carries into the next, the top window of every chunk takes the
    carry out of bit 63; scalars >= r are summed as the integers they are (like the exact path).  split_ns = 0 lets the
    planner pick mixed window widths at this size."""
    _release()
    monkeypatch.setenv("BLAZE_MSM_PLAN", "split_ns=0")
    n = 1500
    pts, sc, _ = orc.input_generator(curve, n, 8, 47)
    sc = bytearray(sc)
    for i in range(0, n, 3):
        sc[32 * i: 32 * i + 32] = b"\xff" * 32
    for i in range(1, n, 7):
        sc[32 * i: 32 * i + 32] = (b"\xff" * 7 + b"\x7f") * 4
    for i in range(2, n, 11):
        sc[32 * i: 32 * i + 32] = (b"\x00" * 7 + b"\x80") * 4
    sc = bytes(sc)
    exp = orc.msm_pippenger(curve, pts, sc, n, 8, threads=8)
    cl = _plan_client(curve)
    cl.load_data_to_hbm(pts, 0, 0)
    assert run_msm(cl, None, sc, n, hbm=(0, 0)) == exp
    assert cl.precompute_plan_info()["used"]
    cl.close()
    _release()


@pytest.mark.parametrize("curve,mode", [("BN254", "hide2"), ("BLS381", "hide2"), ("BN254", "pieces"), ("BLS377", "pieces"), ("BN254", "host_pieces")])
def test_plan_through_every_task_shape(gpu, orc, curve, mode, monkeypatch):
    """The plan's 4 n-point task of 64-bit scalars through the three-level sort (forced), piecewise accumulation (forced), the
    host-scalar HBM flow in pieces, and two tasks in flight."""
    _release()
    n = 20011
    pts, sc, exp = orc.input_generator(curve, n, 8, 53)
    sc2 = bytes(sc[160:]) + bytes(sc[:160])
    exp2 = orc.msm_pippenger(curve, pts, sc2, n, 8, threads=8)
    if mode == "hide2":
        monkeypatch.setenv("BLAZE_SORT_HIDE", "2")
    else:
        monkeypatch.setenv("BLAZE_MSM_PIECES", "5")
    cl = _plan_client(curve)
    cl.load_data_to_hbm(pts, 0x2000, 0)
    assert cl.prepare_precompute_plan(n, (0x2000, 0))
    p = MSMParams(n, (0x2000, 0))
    if mode == "host_pieces":
        a, b = sc, sc2                           # host buffers: an idle handle enqueues the task piece by piece
    else:
        ds = blaze_amd.DeviceBuffer(0, n * 32); ds.upload(sc)
        ds2 = blaze_amd.DeviceBuffer(0, n * 32); ds2.upload(sc2)
        a, b = ds, ds2
    assert run_msm(cl, None, a, n, hbm=(0x2000, 0)) == exp
    assert cl.precompute_plan_info()["used"]
    cl.initialize(p); cl.start_process(); cl.set_data(MSMInput(None, a, p))
    cl.initialize(p); cl.start_process(); cl.set_data(MSMInput(None, b, p))
    cl.wait_result(); assert cl.result().result == exp
    cl.wait_result(); assert cl.result().result == exp2
    assert cl.precompute_plan_info()["used"]
    cl.close()
    _release()


def test_plan_survives_a_table_loaded_in_pieces_and_mode_iii(gpu, orc):
    """The table arrives in ragged pieces (load_data_to_hbm, msm_api.rs:299-313) - some of them cutting elements in two; a task
    of set_data mode iii (points + hbm address: load, then scalars; msm_api.rs:203-216) brings its own table and takes the exact
    path like a DMA-mode task, and its write re-arms the check for the tasks that follow."""
    _release()
    curve, n = "BN254", 5000
    ps = orc.point_bytes(curve)
    pts, sc, exp = orc.input_generator(curve, n, 8, 59)
    cl = _plan_client(curve)
    cuts = [0, 1000 * 8 * ps + 3 * ps, 2500 * 8 * ps + 40, 4999 * 8 * ps + 7 * ps, len(pts)]
    for a, b in zip(cuts, cuts[1:]):
        cl.load_data_to_hbm(pts[a:b], 0, a)
    assert run_msm(cl, None, sc, n, hbm=(0, 0)) == exp
    assert cl.precompute_plan_info()["used"]
    pts2, sc2, exp2 = orc.input_generator(curve, n, 8, 61)
    assert run_msm(cl, pts2, sc2, n, hbm=(0, 0)) == exp2          # mode iii: the task brings its own table - exact path, no check
    info = cl.precompute_plan_info()
    assert not info["used"] and info["check"] == "unchecked"
    assert run_msm(cl, None, sc2, n, hbm=(0, 0)) == exp2          # ... which the next scalars-only task over it pays, once
    info = cl.precompute_plan_info()
    assert info["used"] and info["check"] == "consistent"
    cl.close()
    _release()


def _expected_synth(orc, curve, ds, n, start=0):
    # P_i = (start + i + 1) G: the MSM is (sum s_i (start + i + 1)) G
    sc = ds.download(n * 32)
    k = orc.index_weighted_sum(curve, sc, n, start, threads=os.cpu_count() or 1)
    return orc.result_from_affine(curve, orc.generator_mul(curve, k))


def test_config3_2e26_bn254_with_the_plan(gpu, orc):
    """BASELINE config 3 (2^26 BN254, pf = 8, 32 GiB of bases resident) on the checked-table plan: the device-built table
    (blz_synth_points: 2^(32 j) (i + 1) G) passes the check, the task sums 2^28 even bases, the result equals the linearity
    answer and the exact path's bytes."""
    _release()
    curve, n = "BN254", 1 << 26
    dp, ds = synth(curve, n, pf=8)
    exp = _expected_synth(orc, curve, ds, n)
    cl = _plan_client(curve)
    cl.load_data_to_hbm(dp, 0, 0)
    dp.free()
    assert cl.prepare_precompute_plan(n, (0, 0))
    info0 = cl.precompute_plan_info()
    assert run_msm(cl, None, ds, n, hbm=(0, 0)) == exp
    info = cl.precompute_plan_info()
    assert info["used"] and info["check"] == "consistent"
    assert info["even_copy_bytes"] <= (n * 4 * 64) * 1.6          # half of the full Montgomery copy (+ the extent's growth room)
    print("config 3 plan: check %.1f ms, copy %.1f GiB" % (info0["check_ms"], info["even_copy_bytes"] / 2**30))
    cl.set_precompute_plan(False)
    assert run_msm(cl, None, ds, n, hbm=(0, 0)) == exp
    assert not cl.precompute_plan_info()["used"]
    cl.close(); ds.free()
    _release()


@pytest.mark.parametrize("curve", ["BLS377", "BLS381"])
def test_reference_max_shape_2e26_precompute_with_the_plan(gpu, orc, curve):
    """msm_bls12_37{7,81}_precompute_max_test (tests/integration_msm.rs:386-467) - n = 2^26, the generator's 256-element tile
    repeated, 48 GiB of bases loaded in 3 GiB pieces - on the checked-table plan: every bucket that is hit holds 2^18 copies of
    the same even base."""
    _release()
    n, pf = 1 << 26, 8
    tp, ts, exp = orc.input_tile(curve, n, pf, 77)
    reps = 1 << 14
    chunk = bytes(tp) * reps                   # 3 GiB
    cl = _plan_client(curve)
    for k in range((n // 256) // reps):
        cl.load_data_to_hbm(chunk, 0, k * len(chunk))
    del chunk
    assert run_msm(cl, None, bytes(ts) * (n // 256), n, hbm=(0, 0)) == exp
    info = cl.precompute_plan_info()
    assert info["used"] and info["check"] == "consistent"
    print("%s max shape: check %.1f ms" % (curve, info["check_ms"]))
    cl.close()
    _release()
