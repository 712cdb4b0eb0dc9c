#!/bin/bash
# default bench line (short) under environment settings, in-tree library, two rounds: tools/ab_envs3.sh "A=1" "B=2" ...
for rep in 1 2; do
for E in "$@"; do
  env $E python bench.py --no-cpu-baseline --no-parity --steps 40 --warmup 3 2>/dev/null | tail -1 | \
    python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$E', 'ms/step', r['ms_per_step'], 'k_sinc_ms', r['roofline']['kernel_ms'], 'alone', r['roofline'].get('kernel_ms_alone'))"
done
done
