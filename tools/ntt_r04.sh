# (historical: BLAZE_NTT_HALF existed only while the two kernels were compared in round 4; the whole-tile kernel is gone)
# Round 4: per-pass kernel times and SQ / traffic counters of the 2^27 NTT, whole-tile exchange (BLAZE_NTT_HALF=0, the round-3
# kernel) against the half-tile exchange (=1).  Output: gpurun_out/ntt_r04_half{0,1}.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for h in 0 1; do
  export BLAZE_NTT_HALF=$h
  out=gpurun_out/ntt_r04_half$h.txt
  echo "# BLAZE_NTT_HALF=$h" > $out
  rm -rf gpurun_out/nttprof; timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/nttprof -- python3 tools/ntt_only.py 27 8 > gpurun_out/nttprof.log 2>&1 < /dev/null
  grep "kernel ms" gpurun_out/nttprof.log | tail -3 >> $out
  python3 tools/rocpd_summary.py gpurun_out/nttprof/*/*_results.db < /dev/null | grep -i "ntt" | cut -c1-150 >> $out
  i=0; rm -rf gpurun_out/nttpmc*
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_IFETCH" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $set -d gpurun_out/nttpmc$i -- python3 tools/ntt_only.py 27 2 > gpurun_out/nttpmc$i.log 2>&1 < /dev/null
  done
  python3 tools/pmc_dump.py ntt512 gpurun_out/nttpmc*/*/*_results.db < /dev/null >> $out
  cat $out
done
