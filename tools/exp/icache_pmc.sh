cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/ic
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_HITS -d gpurun_out/ic -o icov -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/ic/log_ov.txt 2>&1
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_HITS -d gpurun_out/ic -o icno -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-overlap > gpurun_out/ic/log_no.txt 2>&1
python - <<PY
import sqlite3,glob
for f in sorted(glob.glob("gpurun_out/ic/*_results.db")):
    c=sqlite3.connect(f)
    rows=c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by 1,2").fetchall()
    print(f)
    for k,n,cnt,a in rows:
        if "k_sinc_fused<1" in k: print("  ", n, cnt, a)
PY
rm -f gpurun_out/ic/*.db
