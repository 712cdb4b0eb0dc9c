"""Does a pitched complex spectrogram (rows on 32- / 128-byte boundaries) write faster?  par_stft_f32 mode 0, 512 / 32 (the healer's
transform) and 1024 / 256, packed rows against out_pitch = bins rounded up to 4 / 8 / 16 complex values."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, scipy.signal, torch
from pyaudiorestoration_amd import _dev, _lib
L = _lib.lib()
dev = 0
for n, n_fft, hop in ((82567936, 512, 32), (57600000, 1024, 256), (23040000, 2048, 128)):
    x = torch.randn(n, dtype=torch.float32, device="cuda")
    win = torch.from_numpy(scipy.signal.get_window("hann", n_fft).astype(np.float32)).cuda()
    bins = n_fft // 2 + 1
    frames = int(L.par_stft_frames(n, n_fft, hop))
    for rnd in (0, 4, 8, 16):
        pitch = bins if rnd == 0 else (bins + rnd - 1) // rnd * rnd
        buf = torch.empty((frames, pitch), dtype=torch.complex64, device="cuda")
        def run():
            _lib.check(L.par_stft_f32(dev, _dev.ptr(x), n, 1, n_fft, hop, 1, _dev.ptr(win), _dev.ptr(buf), 0, 0 if rnd == 0 else pitch, _dev.stream_ptr(dev)))
        run(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        print(f"stft {n_fft}/{hop} complex, {frames} frames, pitch {pitch} ({pitch * 8} B rows): {best:.3f} ms  {frames * bins * 8 / best / 1e6:.0f} GB/s of payload written")
    del buf, x
