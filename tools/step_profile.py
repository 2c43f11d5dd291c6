"""Where does one bench step spend its host time?  (diagnostic)
usage: step_profile.py [n_streams] [torch] [notiming]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
use_torch = "torch" in sys.argv
if use_torch:
    import torch
    torch.cuda.set_device(0)
    torch.cuda.synchronize()
from rtldavis_amd import batch, dsp, synth
cfg = dsp.PacketConfig(19200, 14, 16, 80, "1100101110001001", 8192)
S = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 4096
uniq = synth.synth_streams(range(64))
host = np.tile(uniq, (S // 64, 1))
bd = batch.BatchDemodulator(cfg, S, 33)
bd.upload(host)
timing = "notiming" not in sys.argv
bd.set_timing(2 if timing else 0)
out = []
for it in range(8):
    t0 = time.perf_counter(); bd.run(); t1 = time.perf_counter()
    r = bd.results(); t2 = time.perf_counter()
    out.append(f"timing={timing} torch={use_torch} run {1e3*(t1-t0):.3f} ms  results {1e3*(t2-t1):.3f} ms  n={len(r)}")
print("\n".join(out))
if timing:
    print(bd.timing())
print(bd.counters())
