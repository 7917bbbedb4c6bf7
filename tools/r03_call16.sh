cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests/test_gpu_ntt.py tests/test_gpu_field.py -m gpu -x -q 2>&1 | tail -4
timeout 300 python tools/ntt_only.py 2>&1 | tail -6
