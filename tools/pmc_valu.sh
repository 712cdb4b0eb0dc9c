#!/bin/bash
# VALU / SALU / LDS instruction counts of K_sinc for one or more library builds (rocprofv3 --pmc pass each):
#   tools/pmc_valu.sh OUTDIR lib1.so [lib2.so ...]
OUT=$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for L in "$@"; do
  tag=$(basename "$L" .so)
  PAR_HIP_LIB=$PWD/$L rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d "$OUT" -o "pmc_$tag" -- \
    python bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/pmc_$tag.log" 2>&1
  python - "$OUT/pmc_${tag}_results.db" "$tag" <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%k_sinc_fused<1%' group by 1, 2").fetchall()
for k, n, cnt, a in rows:
    print(sys.argv[2], k[:40], n, cnt, f"{a:.4g}", f"per output x64: {a * 64 / 691199999:.1f}" if n.startswith("SQ_INSTS") else "")
PY
done
