cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export CHECK=0 REPS=2 BLAZE_SORT_HIDE=2
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d gpurun_out/pmc_i -- python3 tools/gpu_big.py 26 > /dev/null 2>&1
python3 tools/pmc_generic.py gpurun_out/pmc_i/*/*_results.db k3_ | tee gpurun_out/r03_sort3_insts.txt
python3 tools/pmc_generic.py gpurun_out/pmc_i/*/*_results.db k_accumulate | tee -a gpurun_out/r03_sort3_insts.txt
python3 tools/pmc_generic.py gpurun_out/pmc_i/*/*_results.db k_reduce_level0 | tee -a gpurun_out/r03_sort3_insts.txt
rm -rf gpurun_out/pmc_i
