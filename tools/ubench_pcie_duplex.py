#!/usr/bin/env python3
"""Does this box move H2D and D2H at the same time?  Pinned 460 MB buffers, two streams: one direction alone, both
directions back to back on one stream, both directions on two streams."""
import json
import time

import torch

n = 115_200_000
h_in = torch.empty(n, dtype=torch.float32).pin_memory()
h_out = torch.empty(n, dtype=torch.float32).pin_memory()
d_in = torch.empty(n, dtype=torch.float32, device="cuda")
d_out = torch.empty(n, dtype=torch.float32, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def up():
    with torch.cuda.stream(s1):
        d_in.copy_(h_in, non_blocking=True)


def down(stream=s2):
    with torch.cuda.stream(stream):
        h_out.copy_(d_out, non_blocking=True)


def both_one_stream():
    up()
    down(s1)


def both_two_streams():
    up()
    down(s2)


gb = n * 4 / 1e9
res = {"H2D alone GB/s": round(gb / timed(up), 1), "D2H alone GB/s": round(gb / timed(down), 1),
       "both, one stream: ms": round(timed(both_one_stream) * 1e3, 2),
       "both, two streams: ms": round(timed(both_two_streams) * 1e3, 2)}
print(json.dumps(res, indent=1))
