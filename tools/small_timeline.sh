# Dev tool: lone small MSMs (the reference's default MSM_SIZE is 8192): wall / device time per size, then the kernel timeline of one 2^13 task.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for l in 10 13 16 18 20; do python3 tools/latency_probe.py $l 15 2>&1 | tail -1; done
rm -rf gpurun_out/prof_small
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_small -- python3 tools/latency_probe.py ${1:-13} 6 > gpurun_out/small.log 2>&1
tail -1 gpurun_out/small.log
python3 tools/rocpd_timeline.py gpurun_out/prof_small/*/*_results.db ${2:-k_finish} | cut -c1-130
