#!/bin/bash
# A/B builds of libpar_hip.so on the config-4 chain (tools/bench_heal.py) in ONE gpurun session
for rep in 1 2; do
  for L in "$@"; do
    PAR_HIP_LIB=$PWD/$L python tools/bench_heal.py 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$L', round(r['ms'],3), {k: round(v,3) for k,v in r['parts_ms'].items()})"
  done
done
