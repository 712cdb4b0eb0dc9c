#!/bin/bash
# Sweep of environment settings on ONE library in one gpurun session:  tools/ab_envs.sh LIB "ENV=1 ENV2=2" "X=" ...   ("X=" = the plain run)
L=$1; shift
for E in "$@"; do
  env $E PAR_HIP_LIB=$PWD/$L python bench.py --no-cpu-baseline --no-parity --steps 30 --warmup 6 2>/dev/null | tail -1 | \
    python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$E', 'ms/step', r['ms_per_step'], 'k_sinc_ms', r['roofline']['kernel_ms'], 'alone', r['roofline'].get('kernel_ms_alone'))"
done
