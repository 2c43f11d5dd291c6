import sys; sys.path.insert(0,".")
import numpy as np
from rtldavis_amd import batch, dsp, synth
cfg = dsp.PacketConfig(19200, 14, 16, 80, "1100101110001001", 8192)
host = np.tile(synth.synth_streams(range(64)), (64, 1))
bd = batch.BatchDemodulator(cfg, 4096, 33); bd.upload(host)
for _ in range(35): bd.run()
bd.results()
