# Dev tool: per-pass kernel times of the 2^27 NTT under rocprofv3 (gpurun_out/nttprof*).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/nttprof -- python3 tools/ntt_only.py 27 8 > gpurun_out/nttprof.log 2>&1 < /dev/null
grep "kernel ms" gpurun_out/nttprof.log | tail -3
python3 tools/rocpd_summary.py gpurun_out/nttprof/*/*_results.db < /dev/null | grep -i "ntt" | cut -c1-150
