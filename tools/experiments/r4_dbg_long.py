import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle import c_oracle as CO
from rtldavis_amd import batch, dsp, synth
cfg = dsp.PacketConfig(19200, 14, 16, 80, "1100101110001001", 8192)
ns, parts = 3, 24
raw = np.stack([np.concatenate([synth.synth_stream(100 + 17 * s + k) for k in range(parts)]) for s in range(ns)])
nb = parts * 33
want, wbits = CO.demod_batch(raw, CO.make_cfg(), threads=4, want_bits=True, cap_per_stream=4096)
bd = batch.BatchDemodulator(cfg, ns, nb)
for rep in range(2):
    bd.upload(raw); bd.run(); res = bd.packets()
    print(os.environ.get("RD_TAIL_IMPL"), "rep", rep, bd.last_run_forms(), bd.counters())
    for i in range(ns):
        b = bd.bits(i)
        d = np.nonzero(b != wbits[i])[0]
        print(" stream", i, "differing bytes", d.size, d[:12], [(int(b[j]), int(wbits[i][j])) for j in d[:6]])
