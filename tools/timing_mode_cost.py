import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from rtldavis_amd import batch, dsp, synth
cfg = dsp.PacketConfig(19200, 14, 16, 80, "1100101110001001", 8192)
host = np.tile(synth.synth_streams(range(64)), (64, 1))
bds = [batch.BatchDemodulator(cfg, 4096, 33) for _ in range(2)]
for b in bds: b.upload(host)
def loop(tm, steps=600):
    for b in bds:
        b.set_timing(tm); b.run(); b.results()
        if tm: b.timing()
    t0 = time.perf_counter()
    bds[0].run()
    for i in range(steps):
        bds[(i + 1) % 2].run()
        bds[i % 2].results()
    bds[steps % 2].results()
    print(f"timing mode {tm}: {(time.perf_counter() - t0) / (steps + 1) * 1e3:.4f} ms/step", flush=True)
    if tm:
        for b in bds: b.timing()
for r in range(3):
    loop(0); loop(1)
