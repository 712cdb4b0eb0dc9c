#!/bin/bash
# per-kernel durations of the mono K_sinc launch on a 10-min file, for several library builds (kernel trace)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for L in "$@"; do
  echo "== $L"
  PAR_HIP_LIB=$PWD/$L python tools/exp/pick_speed.py 600 2>/dev/null | grep streaming
  PAR_HIP_LIB=$PWD/$L rocprofv3 --kernel-trace --stats -d gpurun_out/sf -o t -- python tools/exp/pick_speed.py 600 > /dev/null 2>&1
  python tools/rocpd_stats.py gpurun_out/sf/t_results.db 2>/dev/null | grep -i "k_sinc_pipe<1\|k_sinc_fused_list(" | cut -c1-150
  rm -rf gpurun_out/sf
done
