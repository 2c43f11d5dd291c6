import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from rtldavis_amd import dsp, synth
from rtldavis_amd.ring import BlockRing
cfg = dsp.PacketConfig(19200, 14, 16, 80, "1100101110001001", 8192)
B = 8192
raw = synth.synth_stream(0)
blocks = [raw[2 * B * b: 2 * B * (b + 1)] for b in range(33)]
cblocks = [((b[0::2].astype(np.float64) - 127.5) / 127.5 + 1j * (b[1::2].astype(np.float64) - 127.5) / 127.5) for b in blocks]
for name, blks in (("uint8", blocks), ("complex128", cblocks)):
    dem = dsp.Demodulator(cfg)
    ts, tf = [], []
    for rep in range(6):
        dem.reset()
        for blk in blks:
            t0 = time.perf_counter(); dem.submit(blk); t1 = time.perf_counter(); dem.fetch(); t2 = time.perf_counter()
            ts.append(t1 - t0); tf.append(t2 - t1)
    ts, tf = np.array(ts[5:]) * 1e6, np.array(tf[5:]) * 1e6
    print(f"{name}: submit() median {np.median(ts):.1f} us, fetch() median {np.median(tf):.1f} us")
    # zero-copy: blocks already in a registered ring
    ring = BlockRing.create(n_slots=4, block_size=B)
    dem.register_input(ring.data)
    ts, tf = [], []
    for rep in range(6):
        dem.reset()
        for blk in blks:
            ring.put(blk)
            slot, off, kind, count = ring.get()
            t0 = time.perf_counter(); dem.submit_from(off, count, kind == 1); t1 = time.perf_counter(); dem.fetch(); t2 = time.perf_counter()
            ring.release()
            ts.append(t1 - t0); tf.append(t2 - t1)
    ts, tf = np.array(ts[5:]) * 1e6, np.array(tf[5:]) * 1e6
    print(f"{name} from the ring: submit_from() median {np.median(ts):.1f} us, fetch() median {np.median(tf):.1f} us, sum {np.median(ts + tf):.1f}")
    dem.register_input(None)
    ring.close()
