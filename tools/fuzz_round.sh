#!/bin/bash
# The round's fuzz campaigns in one gpurun session:  tools/fuzz_round.sh r05 [resampler seconds]  -> gpurun_out/<tag>_fuzz.txt
TAG=${1:-r05}; T=${2:-300}
OUT=gpurun_out/${TAG}_fuzz.txt
{
echo "# $TAG fuzz campaigns on the final sources (one MI355X, one gpurun session; tools/fuzz_*.py)"
echo "## resampler (default kernels: streaming kernel for mono NT = 32, lazy plans where the curve allows)"; python tools/fuzz_resampler.py $T 500000 2>&1 | tail -2
echo "## resampler, block kernel forced for mono NT = 32 (seeds 600000..)"; PAR_FUZZ_BLOCK=1 python tools/fuzz_resampler.py $((T/3)) 600000 2>&1 | tail -2
echo "## slot"; python tools/fuzz_operator_slot.py 60 2>&1 | tail -1
echo "## stft"; python tools/fuzz_stft.py 60 2>&1 | tail -1
echo "## filters"; python tools/fuzz_filters.py 40 2>&1 | tail -1
echo "## trackers"; python tools/fuzz_trackers.py 40 2>&1 | tail -1
echo "## heal"; python tools/fuzz_heal.py 60 2>&1 | tail -1
} > $OUT 2>&1
cat $OUT
