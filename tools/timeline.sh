# Dev tool: kernel timeline of one steady-state bench step.  usage: [BLAZE_BENCH_LOGN=23] bash tools/timeline.sh label
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tl_$1
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/tl_$1 -- python3 bench.py --no-cpu-baseline --no-check --steps 6 --warmup 2 > gpurun_out/tl_$1.json 2> gpurun_out/tl_$1.err < /dev/null
python3 tools/rocpd_timeline.py gpurun_out/tl_$1/*/*_results.db < /dev/null
