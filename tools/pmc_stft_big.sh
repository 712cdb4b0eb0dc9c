#!/bin/bash
# PMC passes over tools/trace_stft_big.py (four-step STFT kernels), each in its own run with --kernel-trace only:
#   gpurun -- 'bash tools/pmc_stft_big.sh'   -> gpurun_out/bigpmc/*.db ; summarise with tools/rocpd_pmc.py
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/bigpmc
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT" -o fetch -- python $R/tools/trace_stft_big.py > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT" -o write -- python $R/tools/trace_stft_big.py > "$OUT/write.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d "$OUT" -o sq -- python $R/tools/trace_stft_big.py > "$OUT/sq.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d "$OUT" -o lds -- python $R/tools/trace_stft_big.py > "$OUT/lds.log" 2>&1
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum -d "$OUT" -o tcc -- python $R/tools/trace_stft_big.py > "$OUT/tcc.log" 2>&1
ls "$OUT"
