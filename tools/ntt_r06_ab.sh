# the A/B part of tools/ntt_r06.sh alone (appends to gpurun_out/r06_ntt_ab.txt)
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_ntt_ab.txt
echo "# A/B on the same box: kernel ms of 12 transforms per build (HIP events), shipped first and last" > $out
for v in "" _nt1 _nt3 _nt5 _nt21 _nt63 _occ4 _occ6 _occ7 ""; do
  lib=blaze_amd/lib/libblaze_hip$v.so
  [ -f $lib ] || continue
  BLAZE_HIP_LIB=$PWD/$lib timeout 200 python3 tools/ntt_only.py 27 14 2>&1 < /dev/null | grep "kernel ms" | tail -12 | python3 -c "
import sys, statistics
v = [float(l.split()[2]) for l in sys.stdin]
print('%-24s median %.3f  min %.3f  max %.3f ms  (n=%d)' % ('libblaze_hip$v', statistics.median(v), min(v), max(v), len(v)))" >> $out
done
cat $out
