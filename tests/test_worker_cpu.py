"""rtldavis_amd.worker's loop logic without a GPU: a fake demodulator stands in (the loop's contract
with the reference's worker, worker.py:34-58: order, stop sentinel, a failing block is dropped)."""
import queue
import threading

import numpy as np


class FakeDem:
    def __init__(self):
        self.flight = []
        self.log = []

    def submit(self, block):
        if block.size != 4:
            raise ValueError("Incompatible array sizes")
        assert len(self.flight) < 2
        self.flight.append(int(block[0]))
        self.log.append(("submit", int(block[0])))

    def fetch(self):
        v = self.flight.pop(0)
        self.log.append(("fetch", v))
        return [v]


class FakeParser:
    def __init__(self):
        self.cfg = None
        self.demodulator = FakeDem()

    def parse(self, packets):
        if packets == [3]:
            raise RuntimeError("parse blew up")  # logged, block dropped, loop goes on
        return [("msg", p) for p in packets]


def test_worker_loop_order_stop_and_errors():
    from rtldavis_amd import worker
    dq, rq = queue.Queue(), queue.Queue()
    for i in range(6):
        dq.put(np.full(4, i, np.uint8))
    dq.put(np.zeros(3, np.uint8))  # wrong size: submit raises, block dropped
    dq.put(np.full(4, 9, np.uint8))
    dq.put(None)
    parsers = []

    def factory():
        parsers.append(FakeParser())
        return parsers[-1]

    t = threading.Thread(target=worker.worker_loop, args=(dq, rq, factory), kwargs=dict(poll_s=0.02))
    t.start(); t.join(20)
    assert not t.is_alive()
    got = []
    while not rq.empty():
        got.append(rq.get())
    assert got == [("msg", 0), ("msg", 1), ("msg", 2), ("msg", 4), ("msg", 5), ("msg", 9)]
    log = parsers[0].demodulator.log
    # every block is fetched before the next one is submitted (parse() reads the demodulator's state)
    assert [op for op, _ in log] == ["submit", "fetch"] * 7
    assert parsers[0].demodulator.flight == []


def test_worker_loop_survives_a_failing_factory_and_idle_polls():
    from rtldavis_amd import worker

    def bad():
        raise RuntimeError("no parser")

    worker.worker_loop(queue.Queue(), queue.Queue(), bad, poll_s=0.01)  # returns, like worker.py:30-32
    dq, rq = queue.Queue(), queue.Queue()
    t = threading.Thread(target=worker.worker_loop, args=(dq, rq, FakeParser), kwargs=dict(poll_s=0.01))
    t.start()
    import time
    time.sleep(0.1)           # idle polling
    dq.put(np.full(4, 7, np.uint8))
    time.sleep(0.1)
    assert rq.get(timeout=2) == ("msg", 7)  # delivered without waiting for a further block
    dq.put(None)
    t.join(10)
    assert not t.is_alive()


class FlakyDem(FakeDem):
    """fetch() fails on the device side once: the block stays in flight (rd_demod_fetch returned an error)."""

    def __init__(self):
        super().__init__()
        self.fail_next = True

    @property
    def inflight(self):
        return len(self.flight)

    def fetch(self):
        if self.fail_next and self.flight and self.flight[0] == 1:
            self.fail_next = False
            raise RuntimeError("device lost")
        return super().fetch()


def test_worker_loop_drains_the_handle_after_a_failed_fetch():
    from rtldavis_amd import worker

    class P(FakeParser):
        def __init__(self):
            super().__init__()
            self.demodulator = FlakyDem()

    dq, rq = queue.Queue(), queue.Queue()
    for i in (0, 1, 2, 4):
        dq.put(np.full(4, i, np.uint8))
    dq.put(None)
    t = threading.Thread(target=worker.worker_loop, args=(dq, rq, P), kwargs=dict(poll_s=0.02))
    t.start(); t.join(20)
    assert not t.is_alive()
    got = []
    while not rq.empty():
        got.append(rq.get())
    # block 1 is lost with its failed fetch; blocks 2 and 4 come back as themselves, not shifted by one
    assert got == [("msg", 0), ("msg", 2), ("msg", 4)]


def test_worker_main_has_the_reference_signature_and_builds_the_reference_parser(tmp_path, monkeypatch):
    """worker.worker_main(data_queue, result_queue, station_id, symbol_length, log_level) - the reference's own
    positional arguments (src/rtldavis/worker.py:10-16, called at runners/rtlsdr.py:61-65) - on a stand-in package
    whose protocol module does `from . import dsp` like the reference's (protocol.py:1-20)."""
    import inspect
    import logging
    import sys
    from rtldavis_amd import dsp as hip_dsp, worker

    assert list(inspect.signature(worker.worker_main).parameters) == [
        "data_queue", "result_queue", "station_id", "symbol_length", "log_level"]
    pkg = tmp_path / "fakeref"
    pkg.mkdir()
    (pkg / "__init__.py").write_text("")
    (pkg / "dsp.py").write_text("raise ImportError('the CPU dsp must not be imported once the swap is in place')\n")
    (pkg / "protocol.py").write_text(
        "from . import dsp\n"
        "class _Dem:\n"
        "    def __init__(self): self.flight = []\n"
        "    def submit(self, b): self.flight.append(int(b[0]))\n"
        "    def fetch(self): return [self.flight.pop(0)]\n"
        "class Parser:\n"
        "    def __init__(self, symbol_length, station_id=None):\n"
        "        self.args = (symbol_length, station_id)\n"
        "        self.dsp_module = dsp\n"
        "        self.cfg = dsp.PacketConfig(19200, symbol_length, 16, 80, '1100101110001001', 8192)\n"
        "        self.demodulator = _Dem()\n"
        "    def parse(self, packets):\n"
        "        return [(self.args, self.dsp_module.__name__, p) for p in packets]\n")
    monkeypatch.syspath_prepend(str(tmp_path))
    monkeypatch.setattr(worker, "REFERENCE_PACKAGE", "fakeref")
    for k in [k for k in sys.modules if k == "fakeref" or k.startswith("fakeref.")]:
        del sys.modules[k]
    dq, rq = queue.Queue(), queue.Queue()
    dq.put(np.full(4, 5, np.uint8))
    dq.put(None)
    t = threading.Thread(target=worker.worker_main, args=(dq, rq, 3, 14, logging.WARNING))
    t.start(); t.join(20)
    assert not t.is_alive()
    assert rq.get(timeout=2) == ((14, 3), hip_dsp.__name__, 5)
    assert sys.modules["fakeref.dsp"] is hip_dsp
    for k in [k for k in sys.modules if k == "fakeref" or k.startswith("fakeref.")]:
        del sys.modules[k]


def test_multi_worker_interrupt_on_the_first_get_stops_cleanly(monkeypatch):
    """a KeyboardInterrupt raised by the very first queue read used to leave `samples` unbound."""
    from rtldavis_amd import worker

    class Q:
        def get(self, timeout=None):
            raise KeyboardInterrupt

    class M:
        def __init__(self, cfg, n):
            self.cfg = cfg

    class Cfg:
        block_size = 2

    class P:
        cfg = Cfg()
        demodulator = None

    monkeypatch.setattr(worker.dsp, "MultiDemodulator", M)
    worker.multi_worker_main([Q()], queue.Queue(), lambda k: P(), poll_s=0.01)  # returns instead of raising


# ---------------------------------------------------------------- round 4: shared-memory ring, pre-imported protocol
def _ring_producer(name, n_blocks, block_size):
    """Child process: what the SDR side does - one put() per block, then stop()."""
    from rtldavis_amd.ring import BlockRing
    ring = BlockRing.attach(name)
    for i in range(n_blocks):
        if i % 3 == 2:   # every third block arrives as complex samples (pyrtlsdr's form)
            blk = (np.arange(block_size) + i).astype(np.complex64) * (1 + 0.5j)
        else:
            blk = ((np.arange(2 * block_size) + 7 * i) % 251).astype(np.uint8)
        assert ring.put(blk, timeout=20.0)
    ring.stop()
    ring.close()


def test_block_ring_two_processes_wrap_around_and_stop():
    """rtldavis_amd.ring.BlockRing between two processes: 23 blocks through 4 slots (several wrap-arounds, the producer
    blocked on a full ring), uint8 and complex blocks mixed, a slot released only after the consumer is done with it,
    STOP after the last block (the data queue hop of runners/rtlsdr.py:100-103 -> worker.py:37 without pickle)."""
    import multiprocessing as mp
    from rtldavis_amd.ring import BlockRing, KIND_C128, KIND_U8, STOP
    bs, n_blocks = 64, 23
    ring = BlockRing.create(n_slots=4, block_size=bs)
    try:
        assert ring.slot_bytes % 64 == 0 and ring.data_offset % 4096 == 0 and ring.data.size == 4 * ring.slot_bytes
        ctx = mp.get_context("spawn")
        prod = ctx.Process(target=_ring_producer, args=(ring.name, n_blocks, bs))
        prod.start()
        seen = 0
        held = []   # the consumer keeps up to two blocks (one on the GPU, one being parsed) before releasing the oldest
        while True:
            item = ring.get(taken=len(held), timeout=20.0)
            assert item is not None, "timed out"
            if item is STOP:
                break
            slot, off, kind, count = item
            assert off == slot * ring.slot_bytes
            if seen % 3 == 2:
                assert kind == KIND_C128 and count == bs
                want = ((np.arange(bs) + seen).astype(np.complex64) * (1 + 0.5j)).astype(np.complex128)
                assert np.array_equal(ring.slot_view(slot, kind, count), want)
            else:
                assert kind == KIND_U8 and count == 2 * bs
                assert np.array_equal(ring.slot_view(slot, kind, count), ((np.arange(2 * bs) + 7 * seen) % 251).astype(np.uint8))
            held.append(slot)
            seen += 1
            if len(held) == 2:
                held.pop(0)
                ring.release()
        while held:
            held.pop(0)
            ring.release()
        prod.join(20)
        assert prod.exitcode == 0 and seen == n_blocks and ring.backlog == (n_blocks, n_blocks)
        assert ring.claim(timeout=0.01) is not None         # free again
        import pytest
        with pytest.raises(ValueError):
            ring.put(np.zeros(2 * 16 * bs, np.uint8))       # larger than a slot: the reference's size error
    finally:
        ring.close()


class RingDem(FakeDem):
    """submit_from / register_input on top of FakeDem: the block's first byte identifies it."""

    def __init__(self):
        super().__init__()
        self.buf = None

    def register_input(self, buf):
        self.log.append(("register", None if buf is None else int(buf.size)))
        self.buf = buf

    def submit_from(self, offset, count, is_complex=False):
        if count != 4:
            raise ValueError("Incompatible array sizes")
        assert len(self.flight) < 2
        v = int(self.buf[offset])
        self.flight.append(v)
        self.log.append(("submit", v))


def test_ring_worker_loop_order_stop_and_errors():
    """ring_worker_loop = worker_loop's contract (worker.py:34-58) on the ring: order, a failing block is dropped and
    its slot given back, blocks committed before stop() are still delivered, the buffer is registered and unregistered."""
    from rtldavis_amd import worker
    from rtldavis_amd.ring import BlockRing
    ring = BlockRing.create(n_slots=3, block_size=2)
    parsers = []

    class P(FakeParser):
        def __init__(self):
            super().__init__()
            self.demodulator = RingDem()
            parsers.append(self)

    rq = queue.Queue()
    t = threading.Thread(target=worker.ring_worker_loop, args=(ring, rq, P), kwargs=dict(poll_s=0.02))
    t.start()
    try:
        for i in (0, 1, 2, 3, 4, 5):
            assert ring.put(np.full(4, i, np.uint8), timeout=10.0)
        assert ring.put(np.zeros(3, np.uint8), timeout=10.0)      # wrong size: submit_from raises, block dropped
        assert ring.put(np.full(4, 9, np.uint8), timeout=10.0)
        ring.stop()
        t.join(20)
        assert not t.is_alive()
        got = []
        while not rq.empty():
            got.append(rq.get())
        assert got == [("msg", 0), ("msg", 1), ("msg", 2), ("msg", 4), ("msg", 5), ("msg", 9)]   # 3: parse() blew up
        log = parsers[0].demodulator.log
        assert log[0] == ("register", ring.data.size) and log[-1] == ("register", None)
        assert [op for op, _ in log[1:-1]] == ["submit", "fetch"] * 7
        assert ring.backlog == (8, 8)      # every slot went back to the producer, the failed ones included
    finally:
        ring.close()


def test_worker_main_when_protocol_was_imported_before_the_swap(tmp_path, monkeypatch):
    """runners/rtlsdr.py:6 imports `protocol` before it starts the worker, and a forked child inherits the
    module bound to the reference's own dsp.  reference_parser_factory rebinds `protocol.dsp` (the module uses it through
    global lookups at call time only) instead of refusing."""
    import logging
    import sys
    from rtldavis_amd import dsp as hip_dsp, worker
    pkg = tmp_path / "fakeref2"
    pkg.mkdir()
    (pkg / "__init__.py").write_text("")
    (pkg / "dsp.py").write_text("class PacketConfig:\n    def __init__(self, *a): raise RuntimeError('the CPU dsp was used')\n")
    (pkg / "protocol.py").write_text(
        "from . import dsp\n"
        "class _Dem:\n"
        "    def __init__(self): self.flight = []\n"
        "    def submit(self, b): self.flight.append(int(b[0]))\n"
        "    def fetch(self): return [self.flight.pop(0)]\n"
        "class Parser:\n"
        "    def __init__(self, symbol_length, station_id=None):\n"
        "        self.cfg = dsp.PacketConfig(19200, symbol_length, 16, 80, '1100101110001001', 8192)\n"
        "        self.demodulator = _Dem()\n"
        "        self.dsp_name = dsp.__name__\n"
        "    def parse(self, packets):\n"
        "        return [(self.dsp_name, p) for p in packets]\n")
    monkeypatch.syspath_prepend(str(tmp_path))
    monkeypatch.setattr(worker, "REFERENCE_PACKAGE", "fakeref2")
    import importlib
    proto = importlib.import_module("fakeref2.protocol")       # what the runner did before the fork
    assert proto.dsp.__name__ == "fakeref2.dsp"
    try:
        dq, rq = queue.Queue(), queue.Queue()
        dq.put(np.full(4, 6, np.uint8))
        dq.put(None)
        t = threading.Thread(target=worker.worker_main, args=(dq, rq, None, 14, logging.WARNING))
        t.start(); t.join(20)
        assert not t.is_alive()
        assert rq.get(timeout=2) == (hip_dsp.__name__, 6)
        assert proto.dsp is hip_dsp
    finally:
        for k in [k for k in sys.modules if k == "fakeref2" or k.startswith("fakeref2.")]:
            del sys.modules[k]
