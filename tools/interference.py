"""What slows the demod kernel inside the bench loop (0.52-0.53 ms) compared with back-to-back launches of one
batch (0.49-0.50)?  Variants: one batch / two resident batches alternating, with and without fetching the
results (the SDMA readback then runs beside the next batch's kernel)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rtldavis_amd import batch, dsp, synth

cfg = dsp.PacketConfig(19200, 14, 16, 80, "1100101110001001", 8192)
host = np.tile(synth.synth_streams(range(64)), (64, 1))
bds = [batch.BatchDemodulator(cfg, 4096, 33) for _ in range(2)]
for b in bds:
    b.upload(host)
    b.set_timing(1)


def loop(nb, fetch, steps=200):
    use = bds[:nb]
    for b in use:
        b.run(); b.results(); b.timing()
    t0 = time.perf_counter()
    use[0].run()
    for i in range(steps):
        nxt = use[(i + 1) % nb]
        if nb == 1:
            if fetch:
                use[0].results()
            use[0].run()
        else:
            nxt.run()
            if fetch:
                use[i % nb].results()
    for b in use:
        b.results()
    wall = (time.perf_counter() - t0) / (steps + 1) * 1e3
    d = [b.timing() for b in use]
    dem = sum(x["demod_ms"] * x["runs"] for x in d) / sum(x["runs"] for x in d)
    print(f"batches {nb} fetch {int(fetch)}: demod {dem:.4f} ms, wall {wall:.4f} ms/step", flush=True)


for rnd in range(2):
    loop(1, False); loop(2, False); loop(2, True); loop(1, True)
