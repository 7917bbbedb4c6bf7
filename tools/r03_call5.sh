# round 3, GPU call 5: slice-major BN254 pf=8: slices x unit length
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export CHECK=0 REPS=3 CURVE=BN254 PF=8
f() { "$@" 2>&1 | grep -E "rep 2" | cut -c1-260; }
for S in 4 8 16; do for L in 32 64 128; do echo "== slices $S L $L"; BLAZE_MSM_SLICES=$S BLAZE_MSM_L=$L f timeout 600 python tools/gpu_big.py 26; done; done
for L in 32 64 128; do echo "== slices 1 L $L"; BLAZE_MSM_SLICES=1 BLAZE_MSM_L=$L f timeout 600 python tools/gpu_big.py 26; done
