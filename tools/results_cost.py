"""Host cost of rd_batch_results on a finished run, and the latency of a one-shot BatchDemodulator run
(run() + results(), nothing else queued) - per tail form when RD_TAIL_IMPL is set in the environment."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rtldavis_amd import batch, dsp, synth
cfg = dsp.PacketConfig(19200, 14, 16, 80, "1100101110001001", 8192)
host = np.tile(synth.synth_streams(range(64)), (64, 1))
bd = batch.BatchDemodulator(cfg, 4096, 33); bd.upload(host)
for _ in range(20):
    bd.run(); bd.results()
bd.run(); r = bd.results()
ts = []
for _ in range(20):
    t = time.perf_counter(); r = bd.results(); ts.append(time.perf_counter() - t)
one = []
for _ in range(50):
    t = time.perf_counter(); bd.run(); r = bd.results(); one.append(time.perf_counter() - t)
print("tail=%s: results() on a finished run: median %.3f ms min %.3f ms, %d records; one-shot run()+results(): median %.3f ms min %.3f ms"
      % (os.environ.get("RD_TAIL_IMPL", "k_tail (one launch)"), 1e3 * sorted(ts)[10], 1e3 * min(ts), len(r), 1e3 * sorted(one)[25], 1e3 * min(one)))
