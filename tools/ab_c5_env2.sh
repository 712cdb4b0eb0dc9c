#!/bin/bash
# config-5 archive rate (one GPU, 96 files) under environment settings, in-tree library
for E in "$@"; do
  env $E python bench.py --config5 --files 96 --n1-files 16 --no-e2e --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | \
    python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$E', 'archive G/s', r['value'], 'ms/step', r['ms_per_step'])"
done
