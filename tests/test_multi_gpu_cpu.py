"""N > 1 path on CPU: two gloo processes run the same sharding + barrier + MAX/SUM reduction code
bench.py uses with one rank per GPU."""
import json
import os
import subprocess
import sys

from pyaudiorestoration_amd import multi_gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_items_partition_is_disjoint_and_complete():
    items = multi_gpu.work_items(512, 2)
    assert len(items) == 1024 and items[0] == (0, 0) and items[3] == (1, 1)
    for world in (1, 2, 4, 8, 3):
        parts = [multi_gpu.shard_items(len(items), world, r) for r in range(world)]
        flat = sorted(i for p in parts for i in p)
        assert flat == list(range(1024))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_world2_gloo_timing_and_reduction():
    import socket
    blocker = socket.socket()                     # the port right above the rendezvous port is taken: the queue's
    blocker.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)     # store must move on to the next free one
    try:
        blocker.bind(("127.0.0.1", 29542))
        blocker.listen(1)
    except OSError:
        pass
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29541", os.path.join(ROOT, "tests", "_dist_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    r = json.loads(line)
    assert r["world"] == 2 and r["total"] == 1024 and r["per_rank"] == 512 and r["first_sum"] == 1
    assert r["dt"] >= 0.055            # MAX over ranks: the slow rank (3 x 20 ms) sets the time
    # shared work queue: 40 items pulled exactly once in total, the faster rank took more of them
    assert r["q_total"] == 40 and r["q_sum"] == sum(range(40)) and 1 <= r["q_min"] < 20
    blocker.close()


def test_world8_gloo_queue_and_reductions():
    """The driver's largest launch shape (8 ranks of one node) over gloo on the CPU: rendezvous, the TCPStore work queue,
    MAX-over-ranks timing and the sum reductions."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29571")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr",
           "127.0.0.1", "--master-port", "29571", os.path.join(ROOT, "tests", "_dist_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert r["world"] == 8 and r["total"] == 1024 and r["per_rank"] == 128 and r["first_sum"] == sum(range(8))
    assert r["q_total"] == 40 and r["q_sum"] == sum(range(40))          # every item pulled exactly once across the 8 ranks


def test_work_queue_single_process():
    ctx = multi_gpu.RankContext()
    assert list(multi_gpu.WorkQueue(ctx, [5, 3, 9], "solo")) == [5, 3, 9]
