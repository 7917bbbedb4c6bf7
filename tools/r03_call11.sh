# round 3, GPU call 11: paired independent products in the mixed add: parity + same-box A/B
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_field.py tests/test_gpu_msm.py -m gpu -x -q -k "ec_ops or golden or harness_sizes or randomised or known_answers or bench_workload or config3 or config4" 2>&1 | tail -3
export CHECK=1 REPS=3
f() { "$@" 2>&1 | grep -E "rep 2|^B" | cut -c1-330; }
N=$GRAFT_REPO_ROOT/blaze_amd/lib/libblaze_hip_nopair.so
for i in 1 2; do
echo "== BLS381 2^26 paired"; f timeout 600 python tools/gpu_big.py 26
echo "== BLS381 2^26 unpaired"; BLAZE_HIP_LIB=$N f timeout 600 python tools/gpu_big.py 26
done
echo "== BN254 2^26 paired"; CURVE=BN254 f timeout 600 python tools/gpu_big.py 26
echo "== BN254 2^26 unpaired"; BLAZE_HIP_LIB=$N CURVE=BN254 f timeout 600 python tools/gpu_big.py 26
b() { timeout 600 python bench.py --no-cpu-baseline --no-ntt --no-extras | python3 -c "
import json,sys;j=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(j['ms_per_step'], j['roofline']['kernel_ms'], j['phases_ms']['sort_ms'], j['result_check']['ok'])"; }
echo "== bench paired"; b; echo "== bench unpaired"; BLAZE_HIP_LIB=$N b; echo "== bench paired"; b
