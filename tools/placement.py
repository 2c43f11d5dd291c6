"""Does the demod kernel's duration depend on where the buffers of a batch happen to be placed?
Creates the batch handle several times in ONE process (fresh allocations each time, other allocations of
varying size in between) and prints the kernel's mean duration for each instance."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rtldavis_amd import batch, dsp, synth

cfg = dsp.PacketConfig(19200, 14, 16, 80, "1100101110001001", 8192)
host = np.tile(synth.synth_streams(range(64)), (64, 1))
junk = []
for inst in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    bd = batch.BatchDemodulator(cfg, 4096, 33)
    bd.upload(host)
    bd.set_timing(1)
    for _ in range(10): bd.run()
    bd.results(); bd.timing()
    for _ in range(40): bd.run()
    bd.results()
    t = bd.timing()
    ptr = bd.input_ptr()[0]
    print(f"instance {inst}: demod {t['demod_ms']:.4f} ms  input at {ptr:#x}", flush=True)
    del bd
    junk.append(torch.empty((inst + 1) * 37 * 2**20 + 4096 * inst, dtype=torch.uint8, device="cuda"))
