# usage: bash tools/prof_msm.sh <tag> <logn> ; kernel-trace summary of a few MSMs
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export CHECK=0 REPS=3
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$1 -- python3 tests/probes/gpu_big.py $2 > gpurun_out/prof_$1.log 2>&1
python3 tools/rocpd_summary.py gpurun_out/prof_$1/*/*_results.db | cut -c1-44,70-130 | head -${3:-14}
