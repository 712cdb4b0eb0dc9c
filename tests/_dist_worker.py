"""Worker for tests/test_multi_gpu_cpu.py: world_size-2 gloo run of the sharding/timing plumbing
that bench.py uses on GPUs (no GPU, no HIP calls here)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyaudiorestoration_amd import multi_gpu  # noqa: E402


def main():
    ctx = multi_gpu.RankContext(backend="gloo")
    items = multi_gpu.work_items(512, 2)                   # config 5: 512 stereo files
    mine = multi_gpu.shard_items(len(items), ctx.world, ctx.rank)
    done = []

    def step():
        time.sleep(0.01 * (ctx.rank + 1))                  # rank 1 is the slow one
        done.append(len(mine))
    dt = ctx.timed(step, 3)
    total = ctx.reduce_sum(len(mine))
    first = ctx.reduce_sum(mine[0])
    # the shared queue bench.py's batch mode pulls from: every item exactly once across the ranks, uneven paces
    pulled = []
    for it in multi_gpu.WorkQueue(ctx, range(40), "t0"):
        pulled.append(it)
        time.sleep(0.002 * (1 + 3 * ctx.rank))
    ctx.barrier()
    q_total = ctx.reduce_sum(len(pulled))
    q_sum = ctx.reduce_sum(sum(pulled))
    q_min = -ctx.reduce_max(-len(pulled))
    if ctx.rank == 0:
        print(json.dumps({"world": ctx.world, "dt": dt, "total": total, "first_sum": first, "per_rank": len(mine),
                          "q_total": q_total, "q_sum": q_sum, "q_min": q_min}))
    ctx.close()


if __name__ == "__main__":
    main()
