#!/bin/bash
# Times several builds of libpar_hip.so in ONE gpurun session: tools/time_libs.sh A.so B.so ...  (2 rounds, interleaved)
for rep in 1 2; do
  for L in "$@"; do
    PAR_HIP_LIB=$PWD/$L python bench.py --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | tail -1 | \
      python -c "import json,sys; r=json.loads(sys.stdin.read()); q=r['roofline']; print('%-44s ms/step %.3f  k_sinc %.3f  alone %.3f  stereo %s' % ('$L', r['ms_per_step'], q['kernel_ms'], q.get('kernel_ms_alone', 0), r.get('secondary_config5', {}).get('ms_per_file')))"
  done
done
