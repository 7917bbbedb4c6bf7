# round 3, GPU call 2: BN254 on the 9 x 29 reduced radix: field / ec / MSM parity, then timings; translation counters of the old claim
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_field.py -m gpu -x -q 2>&1 | tail -5
timeout 1500 python -m pytest tests/test_gpu_msm.py tests/test_dist.py -m gpu -x -q 2>&1 | tail -5
export CHECK=1 REPS=3
f() { "$@" 2>&1 | grep -E "rep 2|^B" | cut -c1-330; }
echo "== BN254 2^26 pf=1"; CURVE=BN254 f timeout 600 python tools/gpu_big.py 26
echo "== BN254 2^26 pf=8"; CURVE=BN254 PF=8 f timeout 900 python tools/gpu_big.py 26
echo "== BN254 2^22"; CURVE=BN254 f timeout 600 python tools/gpu_big.py 22
export CHECK=0 REPS=2
rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum -d gpurun_out/pmc_utcl1_bn254_pf8 -- python3 tools/gpu_big.py 26 > gpurun_out/pmc_utcl1.log 2>&1
python3 tools/pmc_generic.py gpurun_out/pmc_utcl1_bn254_pf8/*/*_results.db k_accumulate | tee gpurun_out/r03_pmc_utcl1_bls381_2e26.txt
CURVE=BN254 PF=8 rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum -d gpurun_out/pmc_utcl1_b -- python3 tools/gpu_big.py 26 > gpurun_out/pmc_utcl1b.log 2>&1
python3 tools/pmc_generic.py gpurun_out/pmc_utcl1_b/*/*_results.db k_accumulate | tee gpurun_out/r03_pmc_utcl1_bn254_pf8_2e26.txt
rm -rf gpurun_out/pmc_utcl1_bn254_pf8 gpurun_out/pmc_utcl1_b
