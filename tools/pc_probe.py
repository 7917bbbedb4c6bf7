"""Checked-table plan vs the exact path over synthetic precompute tables: python3 tools/pc_probe.py CURVE logn [logn ...]"""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import blaze_amd
from blaze_amd.ingo_msm import MSMInput, MSMParams, PointMemoryType
from gpu_util import msm_client, run_msm, synth

curve = sys.argv[1]
for lg in [int(a) for a in sys.argv[2:]]:
    n = 1 << lg
    blaze_amd._lib.check(blaze_amd.lib().blz_arena_release(0))
    dp, ds = synth(curve, n, pf=8)
    ex = msm_client(curve, 8, PointMemoryType.HBM)
    ex.load_data_to_hbm(dp, 0, 0)
    dp.free()
    p = MSMParams(n, (0, 0))

    def stream(cl, k):
        outs, t0 = [], time.perf_counter()
        cl.initialize(p); cl.start_process(); cl.set_data(MSMInput(None, ds, p))
        for i in range(k):
            if i + 1 < k:
                cl.initialize(p); cl.start_process(); cl.set_data(MSMInput(None, ds, p))
            cl.wait_result(); outs.append(cl.result().result)
        return (time.perf_counter() - t0) / k * 1e3, outs

    stream(ex, 2)
    t_ex, o_ex = stream(ex, 4)
    api_ex = ex.get_api()
    ex.close()
    pl = msm_client(curve, 8, PointMemoryType.HBM)
    pl.set_precompute_plan(True)
    ok = pl.prepare_precompute_plan(n, (0, 0))
    info = pl.precompute_plan_info()
    lone = run_msm(pl, None, ds, n, hbm=(0, 0))
    stream(pl, 2)
    t_pl, o_pl = stream(pl, 4)
    api = pl.get_api()
    print(f"{curve} 2^{lg}: exact {t_ex:.2f} ms (acc {api_ex['accumulate_kernel_ms']:.2f}, c={api_ex['window_bits']:.0f} W={api_ex['windows']:.0f}) | plan ok={ok} check {info['check_ms']:.1f} ms "
          f"{t_pl:.2f} ms (acc {api['accumulate_kernel_ms']:.2f}, sort {api['sort_ms']:.2f} hidden={api['sort_hidden']}, reduce {api['phase2_reduce_ms']:.2f}, c={api['window_bits']:.0f} W={api['windows']:.0f}) "
          f"lone_equal={lone == o_ex[0]} stream_equal={[o == o_ex[0] for o in o_pl]} mem={ {k: round(v / 2**30, 2) for k, v in api['device_memory'].items()} }", flush=True)
    pl.close(); ds.free()
blaze_amd._lib.check(blaze_amd.lib().blz_arena_release(0))
