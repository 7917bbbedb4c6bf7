# Regenerates the round's measurement artefacts under gpurun_out/ (copied to profiles/ afterwards).
# The kernel-trace run shows the pipeline as shipped (the next task's sort hidden underneath the accumulation); the two PMC passes
# run with BLAZE_SORT_HIDE=0 so that a dispatch's counters are its own (concurrent kernels share the counters).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=${1:-r05a_final}
python3 bench.py > gpurun_out/${T}_bench_line.json 2> gpurun_out/${T}_bench.err
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$T -- python3 bench.py --no-cpu-baseline --no-extras > gpurun_out/${T}_bench_line_under_rocprof.json 2> gpurun_out/prof_$T.err
python3 tools/rocpd_summary.py gpurun_out/prof_$T/*/*_results.db > gpurun_out/${T}_bench_kernel_stats.txt
python3 tools/rocpd_timeline.py gpurun_out/prof_$T/*/*_results.db > gpurun_out/${T}_step_timeline.txt
BLAZE_SORT_HIDE=0 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch_$T -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/pmc_fetch_$T.log 2>&1
BLAZE_SORT_HIDE=0 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write_$T -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/pmc_write_$T.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_fetch_$T/*/*_results.db gpurun_out/pmc_write_$T/*/*_results.db > gpurun_out/${T}_pmc_hbm_traffic.txt
# ---- the other BASELINE configs, each as the program itself behind `--`: kernel-trace stats of config 3 (exact
# path, then the checked-table plan), config 2's DMA flow and config 4's rank task; FETCH_SIZE / WRITE_SIZE of config 3's accumulation
# (sorts in the open: BLAZE_SORT_HIDE=0, so that a dispatch's counters are its own)
R=${T%%_*}
for v in exact plan; do
  export PC_PLAN=$([ $v = plan ] && echo 1 || echo 0)
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c3$v -- python3 tools/config3_probe.py 26 > gpurun_out/${R}_config3_${v}_probe.json 2> gpurun_out/prof_c3$v.err
  python3 tools/rocpd_summary.py gpurun_out/prof_c3$v/*/*_results.db > gpurun_out/${R}_config3_${v}_kernel_stats.txt
  BLAZE_SORT_HIDE=0 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch_c3$v -- python3 tools/config3_probe.py 26 > gpurun_out/pmc_fetch_c3$v.log 2>&1
  BLAZE_SORT_HIDE=0 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write_c3$v -- python3 tools/config3_probe.py 26 > gpurun_out/pmc_write_c3$v.log 2>&1
  python3 tools/pmc_summary.py gpurun_out/pmc_fetch_c3$v/*/*_results.db gpurun_out/pmc_write_c3$v/*/*_results.db > gpurun_out/${R}_config3_${v}_pmc_hbm_traffic.txt
  rm -rf gpurun_out/prof_c3$v gpurun_out/pmc_fetch_c3$v gpurun_out/pmc_write_c3$v
done
unset PC_PLAN
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c2 -- python3 tools/pcie_inclusive.py 22 > gpurun_out/${R}_config2_probe.json 2> gpurun_out/prof_c2.err
python3 tools/rocpd_summary.py gpurun_out/prof_c2/*/*_results.db > gpurun_out/${R}_config2_kernel_stats.txt
CURVE=BLS377 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c4 -- python3 tools/shard_probe.py 26 8 0 10 > gpurun_out/${R}_config4_probe.txt 2> gpurun_out/prof_c4.err
python3 tools/rocpd_summary.py gpurun_out/prof_c4/*/*_results.db > gpurun_out/${R}_config4_kernel_stats.txt
rm -rf gpurun_out/prof_c2 gpurun_out/prof_c4
# PCIe-inclusive flows (never the bench value): the NTT's host loop (two calls / blz_ntt_exchange, pageable and pinned), config 2 and
# the 2^26 DMA flow
python3 tools/pcie_inclusive_ntt.py 27 > gpurun_out/${R}_pcie_inclusive_ntt_2e27.json 2>> gpurun_out/prof_$T.err
python3 tools/pcie_inclusive.py 22 > gpurun_out/${R}_pcie_inclusive_cfg2_2e22.json 2>> gpurun_out/prof_$T.err
python3 tools/pcie_inclusive.py 26 > gpurun_out/${R}_pcie_inclusive_2e26.json 2>> gpurun_out/prof_$T.err
rm -rf gpurun_out/prof_$T gpurun_out/pmc_fetch_$T gpurun_out/pmc_write_$T
cut -c1-600 gpurun_out/${T}_bench_line.json
head -24 gpurun_out/${T}_bench_kernel_stats.txt | cut -c1-160
cat gpurun_out/${T}_step_timeline.txt | cut -c1-140
head -14 gpurun_out/${T}_pmc_hbm_traffic.txt
for f in config3_exact config3_plan config2 config4; do echo "== $f"; head -8 gpurun_out/${R}_${f}_kernel_stats.txt | cut -c1-150; done
head -6 gpurun_out/${R}_config3_exact_pmc_hbm_traffic.txt gpurun_out/${R}_config3_plan_pmc_hbm_traffic.txt
