# A/B builds of the NTT translation units only (everything else is linked from the shipped objects): BLZ_NTT_NT masks, see ntt_rr.hip.hpp
set -e
cd "$(dirname "$0")/../blaze_amd/csrc"
for m in ${1:-1 3 5 21 63}; do
  d=../../build/obj_nt$m
  rm -rf $d; mkdir -p $d; cp -p ../../build/obj/*.o $d/
  rm -f $d/ntt_bls377.o $d/ntt_bls381.o $d/ntt_bn254.o
  make -j8 OUT=../lib/libblaze_hip_nt$m.so OBJDIR=$d EXTRA=-DBLZ_NTT_NT=$m 2>&1 | grep -E "error|Error" || true
  ls -la ../lib/libblaze_hip_nt$m.so
done
