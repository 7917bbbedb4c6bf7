# round 3, GPU call 1: the whole -m gpu suite, the bench line, the RCCL probe under torch, counter names, BN254 compute-vs-memory split
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03a_pytest.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r03a_pytest.log
timeout 600 python bench.py > gpurun_out/r03a_bench_line.json 2> gpurun_out/r03a_bench.err; echo "bench rc $?"; cut -c1-1500 gpurun_out/r03a_bench_line.json; tail -3 gpurun_out/r03a_bench.err
timeout 300 python tools/rccl_probe.py torch > gpurun_out/r03_rccl_probe_torch.txt 2>&1; echo "probe rc $?"; tail -4 gpurun_out/r03_rccl_probe_torch.txt
(rocprofv3 --list-avail || rocprofv3 -L) > gpurun_out/r03_counters_avail.txt 2>&1; grep -ic "utcl\|tlb" gpurun_out/r03_counters_avail.txt
export CHECK=0 REPS=3
f() { "$@" 2>&1 | grep -E "rep 2|^B" | cut -c1-330; }
M=$GRAFT_REPO_ROOT/blaze_amd/lib/libblaze_hip_mask.so
echo "== BN254 2^26 pf=1 normal"; CURVE=BN254 f timeout 600 python tools/gpu_big.py 26
echo "== BN254 2^26 pf=1 gathers confined to 64 MiB"; BLAZE_HIP_LIB=$M CURVE=BN254 f timeout 600 python tools/gpu_big.py 26
echo "== BN254 2^26 pf=8 normal"; CURVE=BN254 PF=8 f timeout 900 python tools/gpu_big.py 26
echo "== BN254 2^26 pf=8 confined"; BLAZE_HIP_LIB=$M CURVE=BN254 PF=8 f timeout 900 python tools/gpu_big.py 26
echo "== BLS381 2^26 normal"; f timeout 600 python tools/gpu_big.py 26
echo "== BLS381 2^26 confined"; BLAZE_HIP_LIB=$M f timeout 600 python tools/gpu_big.py 26
