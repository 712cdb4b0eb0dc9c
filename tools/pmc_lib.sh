#!/bin/bash
# K_sinc under the counters, for one or more library builds (separate rocprofv3 --pmc passes, kernel trace only):
#   tools/pmc_lib.sh OUTDIR lib1.so [lib2.so ...]   -> per-kernel durations and counter averages of the k_sinc_* kernels
OUT=$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for L in "$@"; do
  tag=$(basename "$L" .so)
  echo "== $tag"
  PAR_HIP_LIB=$PWD/$L rocprofv3 --kernel-trace --stats -d "$OUT" -o "tr_$tag" -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > "$OUT/tr_$tag.log" 2>&1
  python tools/rocpd_stats.py "$OUT/tr_${tag}_results.db" 2>/dev/null | grep -i "k_sinc\|Name" | cut -c1-200
  i=0
  for SET in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
             "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" \
             "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS" \
             "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    PAR_HIP_LIB=$PWD/$L rocprofv3 --kernel-trace --pmc $SET -d "$OUT" -o "p${i}_$tag" -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/p${i}_$tag.log" 2>&1
    python tools/rocpd_pmc.py "$OUT/p${i}_${tag}_results.db" k_sinc 2>/dev/null || tail -3 "$OUT/p${i}_$tag.log"
  done
  rm -f "$OUT"/*_results.db
done
