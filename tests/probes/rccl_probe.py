#!/usr/bin/env python3
"""Dev tool: bring up a one-rank RCCL communicator through the library (blz_msm_comm_init) with NCCL_DEBUG=INFO."""
import os, sys
os.environ.setdefault("NCCL_DEBUG", "INFO")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
if len(sys.argv) > 1 and sys.argv[1] == "torch":
    import torch  # loads torch's own librccl first
import oracle
from gpu_util import msm_client, run_msm
n = 500
pts, sc, exp = oracle.input_generator("BLS381", n, 1, 99)
cl = msm_client("BLS381", 1)
part = run_msm(cl, pts, sc, n)
cl.comm_init(0, 1, cl.comm_unique_id())
print("all_gather_combine ok:", cl.all_gather_combine(part) == exp)
cl.comm_free(); cl.close()
os.system("grep -i rccl /proc/%d/maps | awk '{print $6}' | sort -u" % os.getpid())
