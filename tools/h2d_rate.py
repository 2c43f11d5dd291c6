"""PCIe-inclusive rate of the batch path: upload + run + results per batch, from
pageable and from pinned host memory.  Never bench.py's `value` (inputs resident in HBM there)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.cuda.init()
from rtldavis_amd import batch, dsp, synth

cfg = dsp.PacketConfig(19200, 14, 16, 80, "1100101110001001", 8192)
ns, nb = 4096, 33
uniq = synth.synth_streams(range(64))
host = np.tile(uniq, (ns // 64, 1))
pinned = torch.empty(host.shape, dtype=torch.uint8, pin_memory=True)
pinned.numpy()[:] = host
bd = batch.BatchDemodulator(cfg, ns, nb)
for name, src in (("pageable", host), ("pinned", pinned.numpy())):
    bd.upload(src); bd.run(); bd.results()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        bd.upload(src)
        bd.run()
        recs = bd.results()
    dt = (time.perf_counter() - t0) / reps
    t1 = time.perf_counter(); bd.upload(src); torch.cuda.synchronize(); tu = time.perf_counter() - t1
    print(f"{name}: upload {1e3 * tu:.1f} ms ({host.nbytes / tu / 1e9:.1f} GB/s), upload+run+results {1e3 * dt:.1f} ms "
          f"-> {ns * nb * 8192 / dt / 1e6:.0f} MS/s PCIe-inclusive, {len(recs)} packets")
