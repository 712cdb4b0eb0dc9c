#!/bin/bash
# per-kernel time of the sparse-curve plan (tools/bench_sparse_curve.py <seconds>), longest launch per kernel
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sp && timeout 600 rocprofv3 --kernel-trace -d /tmp/sp -o sp --output-format csv -- python "$GRAFT_REPO_ROOT/tools/bench_sparse_curve.py" "${1:-3600}" > /dev/null 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob("/tmp/sp/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r["Kernel_Name"][:58]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, v in sorted(d.items(), key=lambda kv: -max(kv[1]))[:16]:
    print(f"{k:58s} n={len(v):4d} max={max(v):8.3f} ms  sum={sum(v):9.3f}")
PY
