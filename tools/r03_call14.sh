cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r03g_pytest.log 2>&1; echo "pytest rc $?"; grep -E "passed|failed|error" gpurun_out/r03g_pytest.log | tail -3
python __graft_entry__.py smoke 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/r03g_bench_line.json 2> gpurun_out/r03g_bench.err; echo "bench rc $?"; python3 -c "
import json;j=json.loads(open('gpurun_out/r03g_bench_line.json').read().strip().splitlines()[-1]);print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms'], j['roofline']['frac'], j['roofline']['integer_issue']['frac'], j['ntt_2e27']['ms'], j['hbm_flow']['ms_per_msm_steady'], j['config2_dma']['dur_full_ms'], j['clock']['sclk_mhz_timed_steps'])"
