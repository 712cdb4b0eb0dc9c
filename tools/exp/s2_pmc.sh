#!/bin/bash
# instruction counts / busy cycles of the streaming K_sinc per tape (rocprofv3 --pmc; one pass per counter group)
OUT=gpurun_out/s2pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp -d $OUT -o pmc_$tag -- python tools/exp/unity_only.py > $OUT/log_$tag.txt 2>&1
  python - $OUT/pmc_${tag}_results.db <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%" + __import__('os').environ.get('KPAT','k_sinc_stream') + "%' group by 1, 2").fetchall()
# per-dispatch values in order: 3 tapes x 11 launches
for k, n, cnt, a in rows:
    vals = [r[0] for r in c.execute("select value from counters_collection where kernel_name like '%" + __import__('os').environ.get('KPAT','k_sinc_stream') + "%' and counter_name=? order by dispatch_id", (n,)).fetchall()]
    per = len(vals) // 3
    t = [sum(vals[i * per:(i + 1) * per]) / per for i in range(3)]
    print(f"{n:28s} slow {t[0]:.4g}  fast {t[1]:.4g}  mix {t[2]:.4g}   per-output x64 (INSTS): slow {t[0]*64/115.2e6:.1f} fast {t[1]*64/115.2e6:.1f}")
PY
done
