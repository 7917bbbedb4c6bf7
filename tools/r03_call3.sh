# round 3, GPU call 3: (a) is the 9 x 29 BN254 accumulation memory-bound at pf = 8?  (b) hybrid XYZZ / batched-affine lower bound  (c) level-0 segment sweep
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export CHECK=0 REPS=3
f() { "$@" 2>&1 | grep -E "rep 2|^B" | cut -c1-330; }
M=$GRAFT_REPO_ROOT/blaze_amd/lib/libblaze_hip_mask.so
echo "== BN254 (9x29) 2^26 pf=8 normal"; CURVE=BN254 PF=8 f timeout 900 python tools/gpu_big.py 26
echo "== BN254 (9x29) 2^26 pf=8 confined"; BLAZE_HIP_LIB=$M CURVE=BN254 PF=8 f timeout 900 python tools/gpu_big.py 26
echo "== BN254 (9x29) 2^26 pf=1 confined"; BLAZE_HIP_LIB=$M CURVE=BN254 f timeout 900 python tools/gpu_big.py 26
echo "== hybrid probe"; timeout 900 python tools/hybrid_probe.py 2>&1 | tee gpurun_out/r03_hybrid_probe.txt
for s in 8 16 32 64; do echo "== BLS381 2^26 BLAZE_MSM_SEG=$s"; BLAZE_MSM_SEG=$s f timeout 600 python tools/gpu_big.py 26; done
