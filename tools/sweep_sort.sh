# (needs an experiment build: make -C blaze_amd/csrc OUT=../lib/libblaze_hip_x.so OBJDIR=../../build/obj_x EXTRA=-DBLZ_EXPERIMENT_KNOBS; BLAZE_HIP_LIB=...)
for cl in 10 11 12; do for epb in 16 32 64 128; do
  echo -n "cl=$cl epb=$epb: "
  BLAZE_SORT_CL=$cl BLAZE_SORT_EPB=$epb CHECK=0 REPS=3 timeout 200 python tests/probes/gpu_big.py 26 2>&1 | grep "rep 2" | sed "s/.*'sort_ms': \([0-9.]*\).*'total_ms'.*/\1/; s/.*total_ms': \([0-9.]*\).*sort_ms': \([0-9.]*\).*/total \1 sort \2/"
done; done
