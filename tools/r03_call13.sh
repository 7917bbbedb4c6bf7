cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_x -- python3 bench.py --no-cpu-baseline --no-extras --no-ntt --steps 4 --warmup 2 > /dev/null 2>&1
python3 tools/rocpd_timeline.py gpurun_out/prof_x/*/*_results.db "k_accumulate" | cut -c1-120
rm -rf gpurun_out/prof_x
