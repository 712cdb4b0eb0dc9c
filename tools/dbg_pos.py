import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, inputs
from oracle import oracle_c as C
from pyaudiorestoration_amd import resampling
sc = inputs.bench_speed_curve(2.0, 48000)
st, sp = sc[:, 0] * 48000, sc[:, 1]
ref, _ = C.speed_to_pos(st, sp, 96000)
pos = resampling.speed_to_pos(st, sp, 96000)
print(len(ref), len(pos))
d = np.nonzero(pos != ref)[0]
print("mismatches", len(d), d[:20])
if len(d):
    i = d[0]
    print(i, repr(pos[i]), repr(ref[i]), pos[i] - ref[i], "max abs", np.max(np.abs(pos - ref)))
    print("diff runs:", np.diff(d)[:30])
