import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from rtldavis_amd import batch, dsp, synth
cfg = dsp.PacketConfig(19200, 14, 16, 80, "1100101110001001", 8192)
uniq = synth.synth_streams(range(64))
host = np.tile(uniq, (8, 1))
bd = batch.BatchDemodulator(cfg, 512, 33)
bd.upload(host)
bd.run(); bd.results()
print(bd.counters())
