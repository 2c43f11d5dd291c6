"""How the batch path behaves on inputs other than the bench's: the noise level (in units of 1/127.6, i.e. LSB of the
uint8 samples) from half an LSB to the bench's 6.4, a loud signal, a constant.  Per class: demod kernel and whole-run
time (HIP events), the fraction of 32-sample runs the guard band sent to k_fixup, packets, and whether the packet
lists equal the C oracle's on the unique streams (bits exact on stream 0).  The guard band's fast path compares
|numerator| with a constant threshold: a quieter input has smaller numerators."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rtldavis_amd import batch, dsp, synth
from oracle import c_oracle as CO

cfg = dsp.PacketConfig(19200, 14, 16, 80, "1100101110001001", 8192)
ocfg = CO.make_cfg(19200, 14, 16, 80, "1100101110001001", 8192)
NU, NS, NB = 16, 4096, 33
classes = [("noise 6.4 LSB (bench)", 0.5, 0.05), ("noise 3 LSB", 0.5, 3 / 127.6), ("noise 2 LSB", 0.5, 2 / 127.6),
           ("noise 1 LSB", 0.25, 1 / 127.6), ("noise 0.5 LSB", 0.1, 0.5 / 127.6), ("loud: amplitude 0.9, noise 12 LSB", 0.9, 12 / 127.6)]
bd = batch.BatchDemodulator(cfg, NS, NB)
for name, amp, noise in classes + [("constant 127", None, None)]:
    if amp is None:
        uniq = np.full((NU, 2 * NB * 8192), 127, dtype=np.uint8)
    else:
        uniq = np.stack([synth.synth_stream(s, amplitude=amp, noise=noise) for s in range(NU)])
    host = np.tile(uniq, (NS // NU, 1))
    bd.upload(host)
    bd.set_timing(0)
    for _ in range(3):
        bd.run(); bd.results()
    bd.set_timing(1)
    t0 = time.perf_counter()
    for _ in range(10):
        bd.run(); recs = bd.results()
    wall = (time.perf_counter() - t0) / 10
    tm = bd.timing(); cnt = bd.counters()
    want, wbits = CO.demod_batch(uniq, ocfg, threads=8, want_bits=True, cap_per_stream=4096)
    res = bd.packets()
    ok = all([(c, p.index, bytes(p.data)) for c, ps in enumerate(res[i]) for p in ps] ==
             [(p.call, p.index, bytes(p.data)) for p in want[i]] for i in range(NU))
    ok = ok and np.array_equal(bd.bits(0), wbits[0])
    print("%-36s demod %.3f ms  run %.3f ms  wall %.3f ms  fix-up runs %.4f %%  packets %6d  %s" % (
        name, tm["demod_ms"], tm["total_ms"], 1e3 * wall, 100.0 * cnt["fixup_runs"] * 32 / (NS * NB * 8192) / 32 * 32,
        len(recs), "== C oracle" if ok else "MISMATCH"))
    sys.stdout.flush()
