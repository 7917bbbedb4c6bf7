cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --list-avail < /dev/null 2>/dev/null | grep -oE "\b(SQ|SQC|TCP|TCC|TA)_[A-Z0-9_]+" | sort -u > gpurun_out/avail_counters.txt
wc -l gpurun_out/avail_counters.txt
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_IFETCH" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set -d gpurun_out/nttpmc$i -- python3 tools/ntt_only.py 27 2 > gpurun_out/nttpmc$i.log 2>&1 < /dev/null
  tail -1 gpurun_out/nttpmc$i.log
done
python3 tools/pmc_dump.py ntt512 gpurun_out/nttpmc*/*/*_results.db < /dev/null
