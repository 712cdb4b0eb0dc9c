#!/bin/bash
# A/B of environment settings on ONE library in one gpurun session:  tools/ab_env.sh "ENV_A=1" "ENV_B=1" [bench args]
# (use "X=" for the plain run)
A=$1; B=$2; shift 2
for rep in 1 2; do
  for E in "$A" "$B"; do
    env $E python bench.py --no-cpu-baseline --steps 40 --warmup 8 "$@" 2>/dev/null | tail -1 | \
      python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$E', 'ms/step', r['ms_per_step'], 'k_sinc_ms', r['roofline']['kernel_ms'], 'alone', r['roofline'].get('kernel_ms_alone'), 'archive', r.get('archive_value'))"
  done
done
