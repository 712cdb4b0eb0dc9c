#!/bin/bash
# A/B several library builds in ONE gpurun session: tools/ab_libs.sh lib1.so lib2.so ...   (two rounds, alternating)
for rep in 1 2; do
  for L in "$@"; do
    PAR_HIP_LIB=$PWD/$L python bench.py --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | tail -1 | \
      python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$L', 'ms/step', r['ms_per_step'], 'k_sinc_ms', r['roofline']['kernel_ms'], 'alone', r['roofline'].get('kernel_ms_alone'))"
  done
done
