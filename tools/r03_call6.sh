# round 3, GPU call 6: full -m gpu suite on the final BN254 arrangement + the config table
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r03c_pytest.log 2>&1; echo "pytest rc $?"; grep -E "passed|failed|error" gpurun_out/r03c_pytest.log | tail -3
bash tools/configs_run.sh 2>&1 | tee gpurun_out/r03_configs_raw.txt | cut -c1-300
