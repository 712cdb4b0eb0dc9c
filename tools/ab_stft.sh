#!/bin/bash
# A/B builds of libpar_hip.so on the get_mag 1024/256 kernel in ONE gpurun session:
#   tools/ab_stft.sh A.so B.so ...    -> two rounds over the list, prints kernel ms (HIP events, best of 20)
for rep in 1 2; do
  for L in "$@"; do
    PAR_HIP_LIB=$PWD/$L python tools/stft_mag_only.py --time 2>/dev/null | tail -1 | sed "s|^|$L  |"
  done
done
