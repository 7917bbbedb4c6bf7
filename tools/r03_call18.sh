cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=$GRAFT_REPO_ROOT/blaze_amd/lib/libblaze_hip_base.so
rocprofv3 --kernel-trace --stats -d gpurun_out/pa -- python3 tools/ntt_only.py 27 6 > /dev/null 2>&1
python3 tools/rocpd_summary.py gpurun_out/pa/*/*_results.db | grep -E "k_ntt512|kernel " | cut -c1-150
BLAZE_HIP_LIB=$B rocprofv3 --kernel-trace --stats -d gpurun_out/pb -- python3 tools/ntt_only.py 27 6 > /dev/null 2>&1
python3 tools/rocpd_summary.py gpurun_out/pb/*/*_results.db | grep -E "k_ntt512" | cut -c1-150
rm -rf gpurun_out/pa gpurun_out/pb
