"""Multi-GPU sharding of the varispeed path: independent (file, channel) work items, one process
per GPU, NO data-path collective.

The reference processes files and channels independently (`for filename ...` util/resampling.py:168,
`for out_channel, in_channel ...` :225), so the work shards embarrassingly: every rank owns a
disjoint slice of the work-item list, keeps its results (the host gathers them by writing files),
and the only communication is host-side: the benchmark's barrier, a MAX (time) / SUM (sample count) reduction and
the shared work queue's fetch-add -- all over gloo / the c10d TCP store on the loopback interface.  RCCL is never
initialised: nothing on this path touches another GPU's memory (BASELINE north_star: "no RCCL collectives").
"""
import os
import time

import torch


def shard_items(n_items, world, rank):
    """Static longest-first style assignment for equally sized items: rank r takes items
    r, r+world, r+2*world, ...  (every rank gets ceil or floor of n_items/world)."""
    return list(range(rank, n_items, world))


def work_items(n_files, n_channels):
    """(file, channel) pairs in the order the reference would visit them."""
    return [(f, c) for f in range(n_files) for c in range(n_channels)]


def parse_cpulist(text):
    """'0-3,8,10-11' (a sysfs cpulist) -> {0, 1, 2, 3, 8, 10, 11}"""
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def bind_to_device_node(dev, sysfs="/sys"):
    """Best effort: restrict this process to the CPUs of the NUMA node its GPU hangs off, BEFORE it allocates pinned host
    buffers (first touch then places them on that node).  The host-gather leg of the archive moves 57 GB/s per GPU into host
    memory; eight ranks whose buffers all sit on one socket share that socket's DRAM and the inter-socket links.
    Returns the node number, or None when the topology cannot be read (then nothing changes).  Never widens the affinity
    the launcher granted."""
    import os
    try:
        import torch
        pr = torch.cuda.get_device_properties(dev)
        bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        node = int(open(os.path.join(sysfs, "bus", "pci", "devices", bdf, "numa_node")).read())
        if node < 0:
            return None
        cpus = parse_cpulist(open(os.path.join(sysfs, "devices", "system", "node", f"node{node}", "cpulist")).read())
        mine = os.sched_getaffinity(0) & cpus
        if not mine:
            return None
        os.sched_setaffinity(0, mine)
        return node
    except Exception:
        return None


class RankContext:
    """Process-group plumbing shared by bench.py and the CPU tests."""

    def __init__(self, backend=None):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        self.device = "cpu"
        self.distinct_devices = 1 if torch.cuda.is_available() else 0
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            if backend is None:
                backend = os.environ.get("PAR_DIST_BACKEND") or "gloo"      # host-side coordination only
            # gloo announces its connections on STDOUT ("[Gloo] Rank 0 is connected to ..."): the benchmark's stdout is ONE
            # JSON line, so file descriptor 1 points at stderr while the group forms (and for the first collective)
            import sys
            sys.stdout.flush()
            saved = os.dup(1)
            os.dup2(2, 1)
            try:
                dist.init_process_group(backend)
                dist.barrier()
            finally:
                sys.stdout.flush()
                os.dup2(saved, 1)
                os.close(saved)
            self.distinct_devices = 0
            if torch.cuda.is_available():
                n_dev = torch.cuda.device_count()
                local_world = int(os.environ.get("LOCAL_WORLD_SIZE", self.world))
                if local_world > n_dev and os.environ.get("PAR_OVERSUBSCRIBE") != "1":
                    # one rank per GPU is the contract: folding ranks onto fewer devices would still print n_gpus = world
                    raise RuntimeError(f"{local_world} ranks on this node but only {n_dev} GPU(s) visible; set "
                                       "PAR_OVERSUBSCRIBE=1 to share devices on purpose (flow tests on a 1-GPU box)")
                self.distinct_devices = min(local_world, n_dev)
                self.local %= n_dev
                torch.cuda.set_device(self.local)
                self.device = f"cuda:{self.local}"
            self.reduce_device = "cpu"
            self.dist = dist
        elif torch.cuda.is_available():
            torch.cuda.set_device(self.local)
            self.device = f"cuda:{self.local}"

    def barrier(self):
        if self.device != "cpu":
            torch.cuda.synchronize()
        if self.dist:
            self.dist.barrier()

    def timed(self, fn, steps):
        """barrier + sync, run fn() `steps` times, sync + barrier; returns MAX-over-ranks seconds."""
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        self.barrier()
        return self.reduce_max(time.perf_counter() - t0)

    def _reduce(self, value, op):
        if not self.dist:
            return float(value)
        t = torch.tensor([float(value)], dtype=torch.float64, device=self.reduce_device)
        self.dist.all_reduce(t, op=op)
        return float(t.item())

    def reduce_max(self, value):
        return self._reduce(value, self.dist.ReduceOp.MAX if self.dist else None)

    def reduce_sum(self, value):
        return self._reduce(value, self.dist.ReduceOp.SUM if self.dist else None)

    def close(self):
        if self.dist:
            self.dist.destroy_process_group()


class WorkQueue:
    """Shared longest-first queue of work items for the ranks of one node: every rank pulls the next
    `grab` items with ONE fetch-add on a key of a c10d TCP store -- host-side, no collective, no GPU traffic.  `order`
    is the item list, longest first (equal-sized items: any order); `tag` separates queues (one per benchmark step).
    grab = 4: a rank asks rank 0's store thread once per four files (8 ranks x 1.8 ms per file would otherwise be
    ~4400 requests/s) -- except for the LAST grab x world items, which go out one by one (r06): the requests are tickets (one
    fetch-add of 1 each) that every rank maps to the same ranges, so the tail imbalance is one item per rank, not grab - 1 (at 8 GPUs
    a rank's share of the archive is 64 files: three files of imbalance were 5 % of the step)."""
    _store = None

    def __init__(self, ctx, order, tag, grab=4):
        self.ctx, self.order, self.key, self._next = ctx, list(order), f"par_queue_{tag}", 0
        self.grab, self._have = max(1, int(grab)), []
        # tickets 0 .. n_big - 1 are chunks of `grab` items, every later ticket one item of the tail
        self._n_big = max(0, (len(self.order) - self.grab * max(1, int(ctx.world))) // self.grab)
        if ctx.dist and WorkQueue._store is None:
            # rank 0 binds the first free port above the rendezvous port and tells the others through the gloo group
            addr, base = os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("MASTER_PORT", "29533"))
            port, store = 0, None
            if ctx.rank == 0:
                for cand in range(base + 1, base + 33):
                    try:
                        store = ctx.dist.TCPStore(addr, cand, ctx.world, is_master=True, wait_for_workers=False)
                        port = cand
                        break
                    except (RuntimeError, OSError):
                        continue
            port = int(ctx.reduce_max(port))
            if port == 0:
                raise RuntimeError(f"WorkQueue: no free port in {base + 1}..{base + 32} for the queue's TCP store")
            if ctx.rank != 0:
                store = ctx.dist.TCPStore(addr, port, ctx.world, is_master=False)
            WorkQueue._store = store

    def pull(self):
        """Next item, or None when the queue is empty."""
        if not self._have:
            if self.ctx.dist:
                t = WorkQueue._store.add(self.key, 1) - 1
            else:
                t, self._next = self._next, self._next + 1
            if t < self._n_big:
                lo, hi = t * self.grab, (t + 1) * self.grab
            else:
                lo = self._n_big * self.grab + (t - self._n_big)
                hi = lo + 1
            self._have = list(range(lo, min(hi, len(self.order))))
            if not self._have:
                return None
        return self.order[self._have.pop(0)]

    def __iter__(self):
        while True:
            item = self.pull()
            if item is None:
                return
            yield item
