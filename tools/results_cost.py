import sys, time; sys.path.insert(0,".")
import numpy as np
from rtldavis_amd import batch, dsp, synth
cfg = dsp.PacketConfig(19200, 14, 16, 80, "1100101110001001", 8192)
host = np.tile(synth.synth_streams(range(64)), (64, 1))
bd = batch.BatchDemodulator(cfg, 4096, 33); bd.upload(host)
for _ in range(3):
    bd.run(); bd.results()
bd.run(); r = bd.results()
ts = []
for _ in range(20):
    t = time.perf_counter(); r = bd.results(); ts.append(time.perf_counter() - t)
print("results() on a finished run: median %.3f ms min %.3f ms, %d records" % (1e3 * sorted(ts)[10], 1e3 * min(ts), len(r)))
