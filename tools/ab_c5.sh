#!/bin/bash
# config-5 archive rate (one GPU, 96 files) for several library builds in one session: tools/ab_c5.sh A.so B.so ...
for rep in 1 2; do
  for L in "$@"; do
    PAR_HIP_LIB=$PWD/$L python bench.py --config5 --files 96 --n1-files 16 --no-e2e --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | \
      python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$L', 'archive G/s', r['value'], 'K_sinc ms/file alone', r['roofline']['kernel_ms_per_file_alone_min_max'][0])"
  done
done
