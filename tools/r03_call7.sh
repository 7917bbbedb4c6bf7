# round 3, GPU call 7: three-level sort: forced everywhere it qualifies (parity), then hidden under the accumulation (bench A/B)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== parity with BLAZE_SORT_HIDE=2 (three-level sort wherever the plan qualifies)"
BLAZE_SORT_HIDE=2 timeout 1500 python -m pytest tests/test_gpu_msm.py -m gpu -x -q 2>&1 | tail -4
export CHECK=1 REPS=3
f() { "$@" 2>&1 | grep -E "rep 2|^B" | cut -c1-330; }
echo "== BLS381 2^26 standard sort alone"; BLAZE_SORT_HIDE=0 f timeout 600 python tools/gpu_big.py 26
echo "== BLS381 2^26 three-level sort alone"; BLAZE_SORT_HIDE=2 f timeout 600 python tools/gpu_big.py 26
echo "== BLS381 2^22 2^24 three-level sort alone"; BLAZE_SORT_HIDE=2 f timeout 600 python tools/gpu_big.py 22 24
for i in 1 2; do
echo "== bench, sort never hidden"; BLAZE_SORT_HIDE=0 timeout 600 python bench.py --no-cpu-baseline --no-ntt --no-extras | python3 -c "
import json,sys;j=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(j['ms_per_step'], j['roofline']['kernel_ms'], j['phases_ms'], j['result_check'])"
echo "== bench, sort hidden (default)"; timeout 600 python bench.py --no-cpu-baseline --no-ntt --no-extras | python3 -c "
import json,sys;j=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(j['ms_per_step'], j['roofline']['kernel_ms'], j['phases_ms'], j['result_check'])"
done
