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


def test_worker_main_order_stop_and_errors():
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

    t = threading.Thread(target=worker.worker_main, args=(dq, rq, factory), kwargs=dict(poll_s=0.02))
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


def test_worker_main_survives_a_failing_factory_and_idle_polls():
    from rtldavis_amd import worker

    def bad():
        raise RuntimeError("no parser")

    worker.worker_main(queue.Queue(), queue.Queue(), bad, poll_s=0.01)  # returns, like worker.py:30-32
    dq, rq = queue.Queue(), queue.Queue()
    t = threading.Thread(target=worker.worker_main, args=(dq, rq, FakeParser), kwargs=dict(poll_s=0.01))
    t.start()
    import time
    time.sleep(0.1)           # idle polling
    dq.put(np.full(4, 7, np.uint8))
    time.sleep(0.1)
    assert rq.get(timeout=2) == ("msg", 7)  # delivered without waiting for a further block
    dq.put(None)
    t.join(10)
    assert not t.is_alive()
