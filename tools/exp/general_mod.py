"""The fc < 1 far-tap sum through modulated images (NOTES r3): float64 check of the identity and of what the first-order
treatment of the per-output deviation eps = fc - fc0 leaves, on the benchmark's curve (worst wave: steepest speed ramp)."""
import numpy as np
NT = 32
n = np.arange(-NT, NT + 1)
win = np.hanning(2 * NT + 1).astype(np.float32).astype(np.float64)
rng = np.random.default_rng(1)
sr = 192000.0
def wave(speed0, dspeed, sig):
    # 256 outputs, speed ramps linearly speed0 -> speed0 + dspeed*255; period = 1/speed
    sp = speed0 + dspeed * np.arange(257)
    per = 1.0 / sp
    p = 100.0 + 0.37 + np.concatenate(([0.0], np.cumsum(per[:-1])))[:256]
    period = per[:256]
    fc = np.minimum(1.0 / period, 1.0)
    ind = np.rint(p).astype(int); s = p - ind
    far = np.abs(n) >= 5
    ref = np.zeros(256); near = np.zeros(256)
    for i in range(256):
        w = win * np.sinc((n - s[i]) * fc[i]) * fc[i]
        x = sig[ind[i] + n]
        ref[i] = np.sum((w * x)[far])
    # modulated images
    m0 = ind[0] - NT - 8
    fc0 = fc[128]; g0 = 1.0 - fc0
    k = np.arange(len(sig)) - m0
    A = sig * np.sin(np.pi * g0 * k); B = sig * np.cos(np.pi * g0 * k)
    out0 = np.zeros(256); out1 = np.zeros(256); out2 = np.zeros(256)
    sgn = (-1.0) ** np.abs(n)
    for i in range(256):
        K = ind[i] - m0
        psi = np.pi * (s[i] - g0 * (K + s[i]))
        wU = sgn * win / (np.pi * (n - s[i]))         # the unity path's operator U
        UA = np.sum((wU * A[ind[i] + n])[far]); UB = np.sum((wU * B[ind[i] + n])[far])
        HA = np.sum((sgn * win * A[ind[i] + n])[far]); HB = np.sum((sgn * win * B[ind[i] + n])[far])
        H1A = np.sum((sgn * win * (n - s[i]) * A[ind[i] + n])[far]); H1B = np.sum((sgn * win * (n - s[i]) * B[ind[i] + n])[far])
        eps = fc[i] - fc0
        t0 = -np.cos(psi) * UA - np.sin(psi) * UB
        t1 = eps * (np.cos(psi) * HB - np.sin(psi) * HA)
        t2 = -(np.pi * eps * eps / 2) * (-np.cos(psi) * H1A - np.sin(psi) * H1B)
        out0[i] = t0; out1[i] = t0 + t1; out2[i] = t0 + t1 + t2
    pk = np.max(np.abs(sig))
    return [np.max(np.abs(o - ref)) / pk for o in (out0, out1, out2)], np.max(np.abs(fc - fc0))
t = np.arange(4000)
sigs = {"noise": rng.standard_normal(4000), "nyquist": np.cos(np.pi * t), "0.45fs": np.cos(0.9 * np.pi * t + 0.2), "fs/4": np.cos(0.5 * np.pi * t + .1)}
# benchmark: speed 1 + 0.01 sin(2 pi 0.55 t): steepest ramp 0.01*2pi*0.55/192000 per sample
ds = 0.01 * 2 * np.pi * 0.55 / sr
for name, sig in sigs.items():
    for sp0, d in ((0.995, ds), (0.990, 1e-9), (0.97, 3 * ds), (0.9995, ds)):
        errs, de = wave(sp0, d, sig)
        print(f"{name:8s} speed {sp0} ramp {d:.2e}/sample  max|eps| {de:.1e}: zeroth {errs[0]:.2e}  first {errs[1]:.2e}  second {errs[2]:.2e}")
