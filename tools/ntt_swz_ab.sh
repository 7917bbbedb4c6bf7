cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for m in 0 1 2; do
  echo "== swz $m"
  BLAZE_NTT_SWZ=$m timeout 200 rocprofv3 --kernel-trace --stats -d gpurun_out/swz$m -- python3 tools/ntt_only.py 27 5 > gpurun_out/swz$m.log 2>&1 < /dev/null
  grep "kernel ms" gpurun_out/swz$m.log | tail -2
  python3 tools/rocpd_summary.py gpurun_out/swz$m/*/*_results.db < /dev/null | grep -i ntt512 | cut -c1-150
done
BLAZE_NTT_SWZ=1 timeout 300 python -m pytest tests/test_gpu_ntt.py -m gpu -x -q < /dev/null 2>&1 | tail -2
