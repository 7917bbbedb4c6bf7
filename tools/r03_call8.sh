# round 3, GPU call 8: full suite with the hidden sort on by default; priority sweep of the hidden sort
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r03d_pytest.log 2>&1; echo "pytest rc $?"; grep -E "passed|failed|error" gpurun_out/r03d_pytest.log | tail -3
b() { timeout 600 python bench.py --no-cpu-baseline --no-ntt --no-extras | python3 -c "
import json,sys;j=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(j['ms_per_step'], j['roofline']['kernel_ms'], j['phases_ms']['sort_ms'], j['result_check']['ok'])"; }
for p in 3 2 1 0 3; do echo "== hidden sort, BLAZE_SORT_PRIO=$p"; BLAZE_SORT_PRIO=$p b; done
echo "== never hidden"; BLAZE_SORT_HIDE=0 b
