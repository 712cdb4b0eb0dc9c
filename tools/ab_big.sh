#!/bin/bash
# kernel trace of tools/trace_stft_big.py for each library variant given (one gpurun session): per-kernel averages
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  tag=$(basename $L .so)
  rm -rf /tmp/abbig_$tag
  PAR_HIP_LIB=$R/$L rocprofv3 --kernel-trace --stats -d /tmp/abbig_$tag -o t -- python $R/tools/trace_stft_big.py > /dev/null 2>&1
  echo "== $L"
  python $R/tools/rocpd_stats.py /tmp/abbig_$tag/t_results.db 2>&1 | grep bigfft | cut -c1-48,72-130
done
