#!/bin/bash
# A/B of several builds of libpar_hip.so in ONE gpurun session: tools/ab_n.sh A.so B.so C.so ... (two rounds, K_sinc alone + step)
for rep in 1 2; do
  for L in "$@"; do
    PAR_HIP_LIB=$PWD/$L python bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | tail -1 | \
      python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$L', 'ms/step', r['ms_per_step'], 'k_sinc_ms', r['roofline']['kernel_ms'], 'alone', r['roofline'].get('kernel_ms_alone'))"
  done
done
