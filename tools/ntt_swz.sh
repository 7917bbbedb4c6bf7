# (needs an experiment build: make -C blaze_amd/csrc OUT=../lib/libblaze_hip_x.so OBJDIR=../../build/obj_x EXTRA=-DBLZ_EXPERIMENT_KNOBS; BLAZE_HIP_LIB=...)
# Dev tool: pass-1 tile order sweep.  BLAZE_NTT_SWZ = (1 + s) + 16 b: 2^s adjacent column groups back to back, then b bits of i1.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in "$@"; do
  export BLAZE_NTT_SWZ=$v
  rm -rf gpurun_out/nttprof; timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/nttprof -- python3 tools/ntt_only.py 27 8 > gpurun_out/nttprof.log 2>&1 < /dev/null
  echo "== SWZ $v (s=$(( (v & 15) - 1 )) b=$(( v >> 4 )))  $(grep 'kernel ms' gpurun_out/nttprof.log | tail -2 | tr '\n' ' ')"
  python3 tools/rocpd_summary.py gpurun_out/nttprof/*/*_results.db < /dev/null | grep "ntt512" | cut -c30-150
done
