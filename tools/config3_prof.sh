# Dev tool: per-kernel times of config 3 (tools/config3_probe.py) under rocprofv3.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_c3
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c3 -- python3 tools/config3_probe.py 26 > gpurun_out/c3.log 2>&1
tail -1 gpurun_out/c3.log | cut -c1-300
python3 tools/rocpd_summary.py gpurun_out/prof_c3/*/*_results.db | head -24 | cut -c1-150
