# A/B builds of the NTT translation units only (everything else is linked from the shipped objects):
#   tools/ntt_r06_build.sh nt  1 3 5 21 63     -DBLZ_NTT_NT=<mask>    non-temporal loads / stores (ntt_rr.hip.hpp)
#   tools/ntt_r06_build.sh occ 4 6 7           -DBLZ_NTT_OCC4=<mask>  four blocks per CU for the passes of the mask
set -e
cd "$(dirname "$0")/../blaze_amd/csrc"
kind=${1:-nt}; shift || true
flag=$([ "$kind" = occ ] && echo BLZ_NTT_OCC4 || echo BLZ_NTT_NT)
for m in "$@"; do
  d=../../build/obj_$kind$m
  rm -rf $d; mkdir -p $d; cp -p ../../build/obj/*.o $d/
  rm -f $d/ntt_bls377.o $d/ntt_bls381.o $d/ntt_bn254.o
  make -j8 OUT=../lib/libblaze_hip_$kind$m.so OBJDIR=$d EXTRA=-D$flag=$m 2>&1 | grep -E "error|Error" || true
  ls -la ../lib/libblaze_hip_$kind$m.so
done
