cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for s in 1 2 3 4 5 6 0; do echo "== BLAZE_NTT_SWZ=$s"; BLAZE_NTT_SWZ=$s python tools/ntt_only.py 27 6 2>&1 | tail -3 | tr '\n' ' '; echo; done
for s in 1 3; do BLAZE_NTT_SWZ=$s rocprofv3 --kernel-trace --stats -d gpurun_out/pa -- python3 tools/ntt_only.py 27 8 > /dev/null 2>&1; echo "swz $s"; python3 tools/rocpd_summary.py gpurun_out/pa/*/*_results.db | grep -E "k_ntt512" | cut -c30-150; rm -rf gpurun_out/pa; done
