#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box (run through gpurun):
#   tools/profile_round.sh r01         -> gpurun_out/r01/summary/*  (tools/summarise_profiles.py run on the box;
#                                         copy them into profiles/)
# Counters are collected in separate passes with --kernel-trace only (never with sys/hip/hsa traces).
set -u
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
rocprofv3 --kernel-trace --stats -d "$OUT" -o trace -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > "$OUT/trace.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT" -o fetch -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT" -o write -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/write.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d "$OUT" -o sq -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/sq.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM -d "$OUT" -o lds -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/lds.log" 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d "$OUT" -o grbm -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/grbm.log" 2>&1
python bench.py --config5 --steps 2 --warmup 1 > "$OUT/bench_config5_n1.json" 2> "$OUT/bench_config5_n1.err"
PAR_OVERSUBSCRIBE=1 python bench.py --gpus 2 --files 64 --n1-files 24 --n1-e2e-files 12 --steps 1 --warmup 1 > "$OUT/bench_config5_2ranks_1gpu.json" 2> "$OUT/bench_config5_2ranks_1gpu.err"
# summarise on the box and keep only the summaries: the six sqlite files outgrew the 64 MiB that gpurun copies back
python tools/summarise_profiles.py "$OUT" "$TAG" > "$OUT/summarise.log" 2>&1
# the default bench once more, now that profiles/pmc_*.json carry this build's digest: the committed line then says
# `stale: false` about the very counters it scales
python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
python - "$OUT" "$TAG" <<'PY'
import json, sys
out, tag = sys.argv[1:3]
bench = json.loads(open(f"{out}/bench_default.json").read().strip().splitlines()[-1])
json.dump(bench, open(f"profiles/{tag}_bench_default.json", "w"), indent=1)
PY
mkdir -p "$OUT/summary"
cp profiles/${TAG}_kernel_stats.txt profiles/${TAG}_pmc.txt profiles/${TAG}_bench_default.json profiles/pmc_traffic.json profiles/pmc_valu.json "$OUT/summary/"
cp "$OUT/bench_config5_n1.json" "$OUT/summary/${TAG}_bench_config5_n1.json"
cp "$OUT/bench_config5_2ranks_1gpu.json" "$OUT/summary/${TAG}_bench_config5_2ranks_1gpu.json"
rm -f "$OUT"/*_results.db
cat "$OUT/bench_default.json" "$OUT/bench_config5_n1.json"
