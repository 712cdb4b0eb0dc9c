#!/bin/bash
# config-5 archive rate (one GPU, 96 files) under several environment settings: tools/ab_c5_env.sh LIB "ENV=1" ...
L=$1; shift
for E in "$@"; do
  env $E PAR_HIP_LIB=$PWD/$L python bench.py --config5 --files 96 --n1-files 16 --no-e2e --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | \
    python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$E', 'archive G/s', r['value'], 'K_sinc ms/file alone', r['roofline']['kernel_ms_per_file_alone_min_max'][0])"
done
