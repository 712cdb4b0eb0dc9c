"""Where does resampling.run() spend its time on a 10-min 192 kHz file (WAV float in, WAV float out)?"""
import logging, os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import inputs
from pyaudiorestoration_amd import io_ops, resampling
logging.basicConfig(level=logging.DEBUG, format="%(message)s")
sr, seconds = 192000, float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
ch = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n = int(sr * seconds)
sig = inputs.bench_signal(0, n, sr)
sig = np.stack([sig] * ch, axis=1)
curve = inputs.bench_speed_curve(seconds, sr)
d = tempfile.mkdtemp()
path = os.path.join(d, "in.wav")
t0 = time.perf_counter(); io_ops.write_wav_float(path, sig, sr); print(f"write_wav_float of the input: {time.perf_counter() - t0:.3f} s")
for rep in range(3):
    t0 = time.perf_counter()
    s2, sr2 = io_ops.read_file(path)[:2]
    t1 = time.perf_counter()
    resampling.run([path], signal_data=[(s2, sr2)], speed_curve=curve, resampling_mode="Sinc", sinc_quality=32)
    t2 = time.perf_counter()
    print(f"rep {rep}: read_file {t1 - t0:.3f} s, run() {t2 - t1:.3f} s  ({n * ch / (t2 - t0) / 1e6:.0f} M channel-samples/s file to file)", flush=True)
