"""k_inpaint_gain / k_apply_gain_boxes alone on the config-4 x256 sparse spectrogram (event timing)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pyaudiorestoration_amd import pipeline, _dev, _lib
rng = np.random.default_rng(4)
n1, sr, n_fft, hop, tiles = 322531, 44100, 512, 32, 256
marks = []
for k in range(tiles):
    for t in np.sort(rng.uniform(0.2, n1 / sr - 0.2, 32)):
        w = rng.uniform(0.004, 0.02)
        marks.append((k * n1 / sr + t - w / 2, 500.0, k * n1 / sr + t + w / 2, 9000.0, 0.5))
geo = np.array([pipeline.marker_geometry(m, sr, hop, n_fft) for m in marks], dtype=np.int32)
frames = (n1 * tiles + 256) // hop + 1
spec = torch.randn(frames, 257, 2, device="cuda") * 0.1
gain = torch.zeros(frames, 257, device="cuda")
g = torch.from_numpy(geo).cuda()
L = _lib.lib(); s = _dev.stream_ptr(0)
def run():
    _lib.check(L.par_inpaint_gain_db_c64(0, _dev.ptr(spec), frames, 257, _dev.ptr(g), len(geo), _dev.ptr(gain), s))
def app():
    _lib.check(L.par_spec_apply_gain_boxes_c64(0, _dev.ptr(spec), frames, 257, _dev.ptr(g), len(geo), _dev.ptr(gain), s))
for f, name in ((run, "inpaint_gain"), (app, "apply_gain_boxes")):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    print(os.path.basename(os.environ.get("PAR_HIP_LIB", "default")), name, f"{e0.elapsed_time(e1) / 10 * 1000:.1f} us")
