import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import oracle, blaze_amd
from blaze_amd.ingo_msm import *
from gpu_util import msm_client, synth
class View(blaze_amd.DeviceBuffer):
    def __init__(self, base, nbytes): self.device_id, self.nbytes, self.ptr = base.device_id, nbytes, base.ptr
    def free(self): pass
rng = random.Random(9)
bad = 0
for c in ("BLS381", "BLS377", "BN254"):
    cl = msm_client(c, 1)
    dp, ds = synth(c, 1 << 23, seed=21)
    ps = 64 if c == "BN254" else 96
    for n in (3 * (1 << 20) + 5, (1 << 22) - 1, (1 << 21) + 12345, 5 * (1 << 20), (1 << 23) - 7, 1500001):
        p = MSMParams(n, None)
        cl.initialize(p); cl.start_process(); cl.set_data(MSMInput(View(dp, n * ps), View(ds, n * 32), p))
        cl.wait_result(); got = cl.result().result
        k = oracle.index_weighted_sum(c, ds.download(n * 32), n, 0)
        exp = oracle.result_from_affine(c, oracle.generator_mul(c, k))
        ok = got == exp
        bad += not ok
        print(c, n, cl.get_api()["window_bits"], cl.get_api()["windows"], "OK" if ok else "MISMATCH", flush=True)
    cl.close(); dp.free(); ds.free()
print("mismatches", bad)
