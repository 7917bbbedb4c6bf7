# bench.py's NTT host loop beside the probe's, on one box (the exchange lost its overlap in bench.py's process on some boxes)
pick='import json,sys
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(d.get("exchange_loop_ms_per_transform_pageable"), d.get("exchange_cycles_ms_pageable"), "pinned", d.get("exchange_loop_ms_per_transform_pinned"))'
echo "probe";              python3 tools/pcie_inclusive_ntt.py 27 2>/dev/null | tail -1 | python3 -c "$pick"
for i in 1 2 3; do echo "bench $i"; python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1])
h=d['ntt_2e27'].get('host_loop'); print({k: v for k, v in h.items() if k not in ('what',)})"; done
