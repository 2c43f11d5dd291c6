"""Kernel time of the one-launch streaming blocks (run under rocprofv3 --kernel-trace --stats): 33-block fixture stream,
uint8 and complex128 input, six passes each."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from rtldavis_amd import dsp, synth
cfg = dsp.PacketConfig(19200, 14, 16, 80, "1100101110001001", 8192)
B = 8192
raw = synth.synth_stream(0)
blocks = [raw[2 * B * b: 2 * B * (b + 1)] for b in range(33)]
cblocks = [((b[0::2].astype(np.float64) - 127.5) / 127.5 + 1j * (b[1::2].astype(np.float64) - 127.5) / 127.5) for b in blocks]
for name, blks in (("uint8", blocks), ("complex128", cblocks)):
    dem = dsp.Demodulator(cfg)
    ts, n = [], 0
    for rep in range(12):
        dem.reset()
        for blk in blks:
            t0 = time.perf_counter(); n += len(dem.demodulate(blk)); ts.append(time.perf_counter() - t0)
    ts = np.array(ts[5:]) * 1e6
    print(f"{name}: demodulate() median {np.median(ts):.1f} us, p10 {np.percentile(ts, 10):.1f}, p90 {np.percentile(ts, 90):.1f}; {n} packets")
