# Regenerates the round's measurement artefacts under gpurun_out/ (copied to profiles/ afterwards).
# The kernel-trace run shows the pipeline as shipped (the next task's sort hidden underneath the accumulation); the two PMC passes
# run with BLAZE_SORT_HIDE=0 so that a dispatch's counters are its own (concurrent kernels share the counters).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=${1:-r03_final}
python3 bench.py > gpurun_out/${T}_bench_line.json 2> gpurun_out/${T}_bench.err
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$T -- python3 bench.py --no-cpu-baseline --no-extras > gpurun_out/${T}_bench_line_under_rocprof.json 2> gpurun_out/prof_$T.err
python3 tools/rocpd_summary.py gpurun_out/prof_$T/*/*_results.db > gpurun_out/${T}_bench_kernel_stats.txt
python3 tools/rocpd_timeline.py gpurun_out/prof_$T/*/*_results.db > gpurun_out/${T}_step_timeline.txt
BLAZE_SORT_HIDE=0 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch_$T -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/pmc_fetch_$T.log 2>&1
BLAZE_SORT_HIDE=0 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write_$T -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/pmc_write_$T.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_fetch_$T/*/*_results.db gpurun_out/pmc_write_$T/*/*_results.db > gpurun_out/${T}_pmc_hbm_traffic.txt
rm -rf gpurun_out/prof_$T gpurun_out/pmc_fetch_$T gpurun_out/pmc_write_$T
cut -c1-600 gpurun_out/${T}_bench_line.json
head -24 gpurun_out/${T}_bench_kernel_stats.txt | cut -c1-160
cat gpurun_out/${T}_step_timeline.txt | cut -c1-140
head -14 gpurun_out/${T}_pmc_hbm_traffic.txt
