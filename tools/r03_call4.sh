# round 3, GPU call 4: slice-major accumulation (parity + config 3 timing), index prefetch, SEG 64
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_msm.py -m gpu -x -q -k "slice_major or config3 or randomised or harness_sizes or golden or two_clients or queue" 2>&1 | tail -4
export CHECK=1 REPS=3
f() { "$@" 2>&1 | grep -E "rep 2|^B" | cut -c1-330; }
echo "== BN254 2^26 pf=8 (auto slices)"; CURVE=BN254 PF=8 f timeout 900 python tools/gpu_big.py 26
echo "== BN254 2^26 pf=8 (8 slices)"; BLAZE_MSM_SLICES=8 CURVE=BN254 PF=8 f timeout 900 python tools/gpu_big.py 26
echo "== BN254 2^26 pf=8 (32 slices)"; BLAZE_MSM_SLICES=32 CURVE=BN254 PF=8 f timeout 900 python tools/gpu_big.py 26
echo "== BN254 2^26 pf=8 (1 slice)"; BLAZE_MSM_SLICES=1 CURVE=BN254 PF=8 f timeout 900 python tools/gpu_big.py 26
echo "== BN254 2^26 pf=1"; CURVE=BN254 f timeout 900 python tools/gpu_big.py 26
echo "== BLS381 2^26"; f timeout 600 python tools/gpu_big.py 26
echo "== BLS377 2^26"; CURVE=BLS377 f timeout 600 python tools/gpu_big.py 26
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r03b_bench_line.json 2> gpurun_out/r03b_bench.err; echo "bench rc $?"; cut -c1-900 gpurun_out/r03b_bench_line.json; python3 -c "
import json;j=json.loads(open('gpurun_out/r03b_bench_line.json').read().strip().splitlines()[-1]);print(j['hbm_flow']);print(j['phases_ms'])"
