"""Per-call latency of the streaming Demodulator (one 8192-sample block = 30.5 ms of air time)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rtldavis_amd import dsp, synth
cfg = dsp.PacketConfig(19200, 14, 16, 80, "1100101110001001", 8192)
raw = synth.synth_stream(0)
dem = dsp.Demodulator(cfg)
ts = []
npk = 0
for rep in range(3):
    dem.reset()
    for b in range(33):
        blk = raw[2 * 8192 * b: 2 * 8192 * (b + 1)]
        t0 = time.perf_counter()
        pk = dem.demodulate(blk)
        ts.append(time.perf_counter() - t0)
        npk += len(pk)
ts = np.array(ts[5:]) * 1e3
print(f"demodulate(): median {np.median(ts):.3f} ms, p99 {np.percentile(ts, 99):.3f} ms, max {ts.max():.3f} ms over {ts.size} calls; {npk} packets")
t0 = time.perf_counter(); d = dem.discriminated; t1 = time.perf_counter()
print(f".discriminated materialisation {1e3*(t1-t0):.3f} ms")
for _ in range(3):
    t0 = time.perf_counter(); d = dem.discriminated; t1 = time.perf_counter(); f = dem.filtered; t2 = time.perf_counter(); q = dem.quantized; t3 = time.perf_counter()
    print(f"again: discriminated {1e3*(t1-t0):.3f} ms, filtered {1e3*(t2-t1):.3f} ms, quantized {1e3*(t3-t2):.3f} ms")
