#!/bin/bash
# per-kernel times of the plan alone for a library variant:  tools/exp/plan_kernel_time.sh lib.so [kernel-substring]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pk && PAR_HIP_LIB=$R/$1 rocprofv3 --kernel-trace --stats -d /tmp/pk -o plan -- python $R/tools/exp/plan_only.py > /dev/null 2>&1
python $R/tools/rocpd_stats.py /tmp/pk/plan_results.db | grep "${2:-k_}" | cut -c1-120 | head -${3:-3}
