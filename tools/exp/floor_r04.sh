#!/bin/bash
# what the r03 kernel shape costs with phases compiled out (PAR_SINC_EXP timing builds, see sinc.hip):
#   1 taps skipped, 2 placement replaced by identity positions, 16 staging loads skipped, 32 stores skipped
# on the slow-only / fast-only / benchmark tapes of tools/exp/unity_only.py (10-min mono file, 115 M outputs).
# Builds (here, no GPU needed):  for v in 1 33 3 17 51 35; do python tools/build_variant.py tools/ab/libpar_exp$v.so -DPAR_SINC_EXP=$v; done
mkdir -p gpurun_out
{
echo "# tools/exp/floor_r04.sh  (K_sinc alone, 10 launches back to back, 115.2 M outputs per launch; ps/output = ms / 115.2e6 * 1e9)"
python tools/exp/unity_only.py 2>&1 | grep -v Warning
for v in 1 33 3 17 51 35; do
  PAR_HIP_LIB=$PWD/tools/ab/libpar_exp$v.so python tools/exp/unity_only.py 2>&1 | grep -v Warning
done
} | tee gpurun_out/r04_sinc_floor_raw.txt
