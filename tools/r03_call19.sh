cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for L in "" blaze_amd/lib/libblaze_hip_tas.so blaze_amd/lib/libblaze_hip_base.so; do
echo "== lib: ${L:-default (tA Montgomery, rest Shoup)}"
BLAZE_HIP_LIB=${L:+$GRAFT_REPO_ROOT/$L} rocprofv3 --kernel-trace --stats -d gpurun_out/pa -- python3 tools/ntt_only.py 27 8 > /dev/null 2>&1
python3 tools/rocpd_summary.py gpurun_out/pa/*/*_results.db | grep -E "k_ntt512" | cut -c30-150
rm -rf gpurun_out/pa
done
timeout 900 python -m pytest tests/test_gpu_ntt.py -m gpu -x -q -k "every_output or against_oracle or inverse or random_sizes" 2>&1 | tail -2
