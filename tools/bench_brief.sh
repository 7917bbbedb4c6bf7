# Dev tool: one bench line, reduced to the numbers that matter while tuning.  usage: bash tools/bench_brief.sh [label]
python bench.py --no-cpu-baseline --no-check 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
p = d['phases_ms']
print('$1', 'MSM/s', d['value'], 'ms/step', d['ms_per_step'], 'acc', p['accumulate_kernel_ms'], 'sort', p['sort_ms'], 'reduce', p['phase2_reduce_ms'])"
