cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=$GRAFT_REPO_ROOT/blaze_amd/lib/libblaze_hip_base.so
for i in 1 2 3; do echo "== shoup"; python tools/ntt_only.py 27 6 2>&1 | tail -3 | tr '\n' ' '; echo; echo "== montgomery"; BLAZE_HIP_LIB=$B python tools/ntt_only.py 27 6 2>&1 | tail -3 | tr '\n' ' '; echo; done
