cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r03h_pytest.log 2>&1; echo "pytest rc $?"; grep -E "passed|failed|error" gpurun_out/r03h_pytest.log | tail -3
bash tools/final_profiles.sh r03h 2>&1 | tail -60
