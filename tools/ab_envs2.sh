#!/bin/bash
# environment sweep on the in-tree library: tools/ab_envs2.sh "ENV=1" "X=" ...
for E in "$@"; do
  env $E python bench.py --no-cpu-baseline --no-parity --steps 40 --warmup 8 2>/dev/null | tail -1 | \
    python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$E', 'ms/step', r['ms_per_step'], 'k_sinc_ms', r['roofline']['kernel_ms'], 'alone', r['roofline'].get('kernel_ms_alone'))"
done
