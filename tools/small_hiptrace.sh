# Dev tool: host-side cost of one small task: HIP API calls (count, total, average) of tools/latency_probe.py at 2^13.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_hip
rocprofv3 --hip-trace --stats -d gpurun_out/prof_hip -f csv -- python3 tools/latency_probe.py ${1:-13} 40 > gpurun_out/hip.log 2>&1
tail -1 gpurun_out/hip.log
f=$(ls gpurun_out/prof_hip/*/*hip_api_stats.csv | head -1)
head -40 $f | tr -d '"' | awk -F, '{printf "%-32s %8s %14s %14s %8s\n", $1, $2, $3, $4, $5}'
