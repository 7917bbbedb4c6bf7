# Round 6: the SQ set of profiles/r04_ntt_half_tile.txt on the SHIPPED 2^27 NTT kernels - per-pass kernel
# times (rocprofv3 --kernel-trace --stats) and SQ / traffic counters per launch (separate --pmc passes) - then ONE experiment on the
# passes' memory side: non-temporal loads / stores (A/B builds blaze_amd/lib/libblaze_hip_nt<mask>.so, tools/ntt_r06_build.sh).
# Output: gpurun_out/r06_ntt_sq.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_ntt_sq.txt
echo "# shipped kernels (BLZ_NTT_NT=0)" > $out
rm -rf gpurun_out/nttprof; timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/nttprof -- python3 tools/ntt_only.py 27 8 > gpurun_out/nttprof.log 2>&1 < /dev/null
grep "kernel ms" gpurun_out/nttprof.log | tail -3 >> $out
python3 tools/rocpd_summary.py gpurun_out/nttprof/*/*_results.db < /dev/null | grep -i "ntt" | cut -c1-150 >> $out
i=0; rm -rf gpurun_out/nttpmc*
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_IFETCH" "SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set -d gpurun_out/nttpmc$i -- python3 tools/ntt_only.py 27 2 > gpurun_out/nttpmc$i.log 2>&1 < /dev/null
done
python3 tools/pmc_dump.py ntt512 gpurun_out/nttpmc*/*/*_results.db < /dev/null >> $out
echo "# A/B on the same box: kernel ms of 12 transforms per build (HIP events), shipped first and last" >> $out
for v in "" _nt1 _nt3 _nt5 _nt21 _nt63 _occ4 _occ6 _occ7 ""; do
  lib=blaze_amd/lib/libblaze_hip$v.so
  [ -f $lib ] || continue
  BLAZE_HIP_LIB=$PWD/$lib timeout 200 python3 tools/ntt_only.py 27 14 2>&1 < /dev/null | grep "kernel ms" | tail -12 | python3 -c "
import sys, statistics
v = [float(l.split()[2]) for l in sys.stdin]
print('%-24s median %.3f  min %.3f  max %.3f ms  (n=%d)' % ('libblaze_hip$v', statistics.median(v), min(v), max(v), len(v)))" >> $out
done
cat $out
