"""Blocks per second through the DSP worker, producer in another process: the reference's multiprocessing.Queue hop
(runners/rtlsdr.py:100-103 -> worker.py:37: a pickle, a pipe, an unpickle and a copy into the pinned slot per block)
against the shared-memory ring (rtldavis_amd/ring.py: one copy on the producer's side, none on the worker's - the GPU
reads the slot in place).  uint8 blocks (16 KB) and complex128 blocks (128 KB, what pyrtlsdr yields).  The parser is
a stand-in that keeps the demodulator and drops the packets: the protocol layer is not what is measured."""
import multiprocessing as mp
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
B, N = 8192, 2000


def _blocks(kind):
    from rtldavis_amd import synth
    raw = synth.synth_stream(0)
    out = [raw[2 * B * b: 2 * B * (b + 1)] for b in range(33)]
    if kind == "complex128":
        out = [((x[0::2].astype(np.float64) - 127.5) / 127.5 + 1j * (x[1::2].astype(np.float64) - 127.5) / 127.5) for x in out]
    return out


def produce_queue(q, kind, n):
    blocks = _blocks(kind)
    for i in range(n):
        q.put(blocks[i % 33])
    q.put(None)


def produce_ring(name, kind, n):
    from rtldavis_amd.ring import BlockRing
    ring = BlockRing.attach(name)
    blocks = _blocks(kind)
    for i in range(n):
        ring.put(blocks[i % 33])
    ring.stop()
    ring.close()


class Stub:
    def __init__(self):
        from rtldavis_amd import dsp
        self.cfg = dsp.PacketConfig(19200, 14, 16, 80, "1100101110001001", B)
        self.demodulator = dsp.Demodulator(self.cfg)
        self.packets = 0

    def parse(self, packets):
        self.packets += len(packets)
        return []


def main():
    from rtldavis_amd import worker
    from rtldavis_amd.ring import BlockRing
    ctx = mp.get_context("spawn")
    rq = ctx.Queue()
    for kind in ("uint8", "complex128"):
        for form in ("queue", "ring"):
            stub = Stub()
            stub.demodulator.demodulate(_blocks("uint8")[0])   # device state allocated outside the timing
            stub.demodulator.reset()
            if form == "queue":
                dq = ctx.Queue(maxsize=8)
                prod = ctx.Process(target=produce_queue, args=(dq, kind, N))
                prod.start()
                first = dq.get()                                # (the producer's start-up is not the hop)
                t0 = time.perf_counter()
                stub.demodulator.submit(first)
                stub.parse(stub.demodulator.fetch())
                worker.worker_loop(dq, rq, lambda: stub, poll_s=0.5)
            else:
                ring = BlockRing.create(n_slots=8, block_size=B)
                prod = ctx.Process(target=produce_ring, args=(ring.name, kind, N))
                prod.start()
                while ring.backlog[0] == 0:
                    time.sleep(0.001)
                t0 = time.perf_counter()
                worker.ring_worker_loop(ring, rq, lambda: stub, poll_s=0.5)
                ring.close()
            dt = time.perf_counter() - t0
            prod.join(30)
            print(f"{kind:10s} blocks through the {form:5s}: {N / dt:8.0f} blocks/s ({1e3 * dt / N:.3f} ms per block, real time needs 32.8 blocks/s); "
                  f"{stub.packets} packets")


if __name__ == "__main__":
    main()
