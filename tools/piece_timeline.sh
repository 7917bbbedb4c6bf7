# Dev tool: kernel timeline of a piecewise task (the reference's HBM flow, lone 2^26 task: tools/hbm_flow_probe.py) between two
# consecutive k_accumulate_cont launches, and the per-kernel totals.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_piece
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_piece -- python3 tools/hbm_flow_probe.py ${1:-26} > gpurun_out/piece.log 2>&1
tail -1 gpurun_out/piece.log
python3 tools/rocpd_timeline.py gpurun_out/prof_piece/*/*_results.db k_accumulate_cont | cut -c1-120
python3 tools/rocpd_summary.py gpurun_out/prof_piece/*/*_results.db | head -30 | cut -c1-150
