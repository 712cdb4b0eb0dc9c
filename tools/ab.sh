#!/bin/bash
# A/B two builds of libpar_hip.so in ONE gpurun session (boxes differ by several % in clocks):
#   tools/ab.sh A.so B.so [bench args]    -> alternates A,B,A,B and prints ms_per_step + K_sinc kernel ms
A=$1; B=$2; shift 2
for rep in 1 2; do
  for L in "$A" "$B"; do
    PAR_HIP_LIB=$PWD/$L python bench.py --no-cpu-baseline --steps 20 --warmup 3 "$@" 2>/dev/null | tail -1 | \
      python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$L', 'ms/step', r['ms_per_step'], 'k_sinc_ms', r['roofline']['kernel_ms'], 'alone', r['roofline'].get('kernel_ms_alone'))"
  done
done
