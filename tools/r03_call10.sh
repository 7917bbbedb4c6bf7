# round 3, GPU call 10: single-pass level 3: parity, A/B, kernel stats of the shipped pipeline (no extras: every k_accumulate launch is the 2^26 workload)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
BLAZE_SORT_HIDE=2 timeout 1500 python -m pytest tests/test_gpu_msm.py -m gpu -x -q -k "hidden or bench_workload or config2 or config4 or harness_2e24 or overlap_large or randomised" 2>&1 | tail -3
b() { timeout 600 python bench.py --no-cpu-baseline --no-ntt --no-extras | python3 -c "
import json,sys;j=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(j['ms_per_step'], j['roofline']['kernel_ms'], j['phases_ms']['sort_ms'], j['result_check']['ok'])"; }
for i in 1 2; do echo "== hidden"; b; echo "== never hidden"; BLAZE_SORT_HIDE=0 b; done
T=r03f
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$T -- python3 bench.py --no-cpu-baseline --no-extras > gpurun_out/${T}_bench_line_under_rocprof.json 2> gpurun_out/prof_$T.err
python3 tools/rocpd_summary.py gpurun_out/prof_$T/*/*_results.db > gpurun_out/${T}_bench_kernel_stats.txt
python3 tools/rocpd_timeline.py gpurun_out/prof_$T/*/*_results.db "k_accumulate" > gpurun_out/${T}_step_timeline.txt
rm -rf gpurun_out/prof_$T
head -22 gpurun_out/${T}_bench_kernel_stats.txt | cut -c1-150
cat gpurun_out/${T}_step_timeline.txt | cut -c1-120
python3 -c "
import json;j=json.loads(open('gpurun_out/r03f_bench_line_under_rocprof.json').read().strip().splitlines()[-1]);print(j['ms_per_step'], j['roofline']['kernel_ms'])"
