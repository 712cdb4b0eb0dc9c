"""Pipelined streaming K_sinc (k_sinc_pipe): how many passes take the loop and why the others leave it.
Needs a library whose sinc2.hip was built with -DPAR_S2_EXP=128 (tools/exp/s2_variant.sh cnt -DPAR_S2_EXP=128)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from pyaudiorestoration_amd import _dev, _lib
L = _lib.lib()
dev, sr, seconds, nt = 0, 192000, 60.0, 32
s = _dev.stream_ptr(dev)
n = int(sr * seconds); m = int(seconds * sr / 256)
sig = torch.empty(n, dtype=torch.float32, device="cuda")
_lib.check(L.par_synth_signal_f32(dev, _dev.ptr(sig), 0, n, float(sr), 0x5EED, s))
t = np.linspace(0, seconds, m)
names = {0: "tiles to the block kernel", 1: "cold passes", 2: "loop iterations", 3: "exit: flag / bad block", 4: "exit: end of range",
         5: "exit: short pass", 6: "exit: tap regime", 7: "exit: conversion window", 8: "exit: g0", 9: "exit: file end", 10: "exit: float16",
         11: "exit: record base"}
for name, sp in (("slow", 0.995 + 0.005 * np.sin(2 * np.pi * 0.55 * t + 0.7)), ("fast", 1.005 + 0.005 * np.sin(2 * np.pi * 0.55 * t + 0.7)),
                 ("mix", 1.0 + 0.01 * np.sin(2 * np.pi * 0.55 * t + 0.7))):
    st_t = torch.from_numpy(t * sr).cuda(); sp_t = torch.from_numpy(sp).cuda()
    cap = int(n * 1.02) + 1024
    nb, ab = int(L.par_speed_plan_bytes(m)), int(L.par_fused_aux_bytes(cap, m))
    work = torch.empty(nb, dtype=torch.uint8, device="cuda"); aux = torch.empty(ab, dtype=torch.uint8, device="cuda")
    out = torch.empty(cap, dtype=torch.float32, device="cuda")
    lo, tr, ok = ctypes.c_int64(0), ctypes.c_int(0), ctypes.c_int(0)
    _lib.check(L.par_speed_to_pos_plan_fused(dev, _dev.ptr(st_t), _dev.ptr(sp_t), m, n, _dev.ptr(work), nb, _dev.ptr(aux), ab, cap,
                                             ctypes.byref(lo), ctypes.byref(tr), 0, None, ctypes.byref(ok), s))
    _lib.check(L.par_varispeed_fused_f32(dev, _dev.ptr(sp_t), m, _dev.ptr(work), _dev.ptr(aux), cap, lo.value, _dev.ptr(sig), 1, n, nt, _dev.ptr(out), 1, s))
    w = (ctypes.c_int * 16)()
    L.par_debug_fused_counters.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
    _lib.check(L.par_debug_fused_counters(dev, _dev.ptr(aux), cap, m, w, s))
    print(f"{name} tape: {lo.value // 1024} tiles, {lo.value / 127.0:.0f} passes expected")
    for k in range(12):
        if w[k]: print(f"   {names[k]:28s} {w[k]}")
