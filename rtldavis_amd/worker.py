"""DSP worker loops on the GPU demodulator.

``worker_main(data_queue, result_queue, station_id, symbol_length, log_level)`` has the reference's signature
(src/rtldavis/worker.py:10-16) and is a drop-in ``multiprocessing.Process`` target for runners/rtlsdr.py:61-65: it
builds the reference's own ``protocol.Parser`` on ``rtldavis_amd.dsp`` (module swap) and
runs ``worker_loop``.

``worker_loop`` is the reference's loop (worker.py:34-58): blocks arrive on a ``multiprocessing.Queue`` (``None`` stops
the loop, a 1 s poll keeps it interruptible), each block is demodulated and parsed, messages go to ``result_queue``;
an exception in the DSP step is logged and the block dropped (worker.py:56-58).  Two things differ, both invisible to
the caller:

* the demodulator call is split into ``submit`` / ``fetch`` (rd_demod_submit / rd_demod_fetch): a block is handed to
  the GPU as soon as it arrives and the loop goes back to the queue - waiting for, and unpickling, block i+1
  (runners/rtlsdr.py:100-103 puts one pickled ndarray per block) runs beside block i's copy and kernels.  Block i is
  fetched AND parsed before block i+1 is submitted: ``Parser.parse`` reads ``demodulator.discriminated`` of the block
  it was given (protocol.py:304-311), which needs a handle with nothing in flight - so there is never more than one
  block on the GPU here, and ``parse()`` itself does not overlap GPU work;
* ``multi_worker_main`` drains several ``data_queue``s (one per dongle / hop channel) into ONE
  ``MultiDemodulator`` launch per round.

The protocol layer is not rebuilt here: the parser is the reference's ``protocol.Parser`` (or, for ``worker_loop``,
any object with ``.cfg``, ``.demodulator`` and ``.parse(packets)`` that ``parser_factory`` supplies).
"""
from __future__ import annotations

import importlib
import logging
import queue
import sys
from typing import Callable, List, Optional, Sequence

import numpy as np

from . import dsp


def _get(q, timeout: float):
    """data_queue.get with the reference's semantics: (block, False), or (None, True) on the stop
    sentinel, or (None, False) when nothing arrived in time (worker.py:35-46)."""
    try:
        samples = q.get(timeout=timeout)
    except queue.Empty:
        return None, False
    if samples is None:
        return None, True
    return samples, False


# The reference package whose ``protocol.Parser`` worker_main builds (a test points this at a stand-in).
REFERENCE_PACKAGE = "rtldavis"


def reference_parser_factory(station_id: Optional[int], symbol_length: int) -> Callable[[], object]:
    """``lambda: protocol.Parser(symbol_length=..., station_id=...)`` (worker.py:29) with ``<package>.dsp`` swapped
    for ``rtldavis_amd.dsp`` first - what INTEGRATION.md section 2 adds to the reference's ``__init__``; done here
    as well so that the entry point works on an unmodified checkout, whether or not ``<package>.protocol`` has been
    imported already (the forked child of runners/rtlsdr.py inherits it: its ``dsp`` name is rebound)."""
    def make():
        pkg = REFERENCE_PACKAGE
        importlib.import_module(pkg)
        sys.modules[pkg + ".dsp"] = dsp
        protocol = importlib.import_module(pkg + ".protocol")
        # runners/rtlsdr.py:6 imports `protocol` before it starts this worker, and under the fork start method the child
        # inherits that module bound to the reference's own dsp.  protocol.py uses `dsp` through module-global lookups at
        # call time only (dsp.PacketConfig in new_packet_config, dsp.Demodulator in Parser.__post_init__), so rebinding
        # the name is all the swap needs
        if getattr(protocol, "dsp", dsp) is not dsp:
            protocol.dsp = dsp
        return protocol.Parser(symbol_length=symbol_length, station_id=station_id)
    return make


def worker_main(data_queue, result_queue, station_id: Optional[int], symbol_length: int, log_level: int) -> None:
    """Drop-in for the reference's ``worker.worker_main`` (worker.py:10-16, same positional arguments): the target of
    ``multiprocessing.Process(target=worker_main, args=(data_queue, result_queue, args.station_id, 14, log_level))``
    at runners/rtlsdr.py:61-65 needs no edit beyond the import."""
    worker_loop(data_queue, result_queue, reference_parser_factory(station_id, symbol_length), log_level)


def worker_loop(data_queue, result_queue, parser_factory: Callable[[], object],
                log_level: int = logging.INFO, poll_s: float = 1.0) -> None:
    """The loop of worker.worker_main (worker.py:18-58) on the submit / fetch halves of the demodulator.

    ``parser_factory()`` must return the parser (reference: ``protocol.Parser(symbol_length=...,
    station_id=...)``, worker.py:29); its ``demodulator`` has to be an ``rtldavis_amd.dsp.Demodulator``.
    """
    logging.basicConfig(level=log_level, format="%(asctime)s - %(name)s - %(levelname)s - %(message)s")
    logger = logging.getLogger("rtldavis.worker")
    logger.info("DSP worker process started")
    try:
        p = parser_factory()
    except Exception as e:  # worker.py:30-32
        logger.exception(f"Failed to initialize worker: {e}")
        return
    dem = p.demodulator
    pending = False  # a block is on the GPU, its packets not yet fetched

    def finish() -> None:
        nonlocal pending
        try:
            packets = dem.fetch()
            # parse() reads demodulator.discriminated for the block just fetched (protocol.py:304-311):
            # the handle is quiet here, the next block is submitted only afterwards
            for msg in p.parse(packets):
                result_queue.put(msg)
        except Exception as e:  # worker.py:56-58: log, drop the block, go on
            logger.error(f"Error in DSP loop: {e}")
            # a fetch that failed on the device side leaves its block in flight: take it out, or every later
            # fetch would hand back the block before the one asked for
            try:
                while getattr(dem, "inflight", 0):
                    dem.fetch()
            except Exception:
                try:
                    dem.reset()
                except Exception as e2:
                    logger.error(f"Error in DSP loop: demodulator reset failed: {e2}")
        pending = False

    while True:
        try:
            samples, stop = _get(data_queue, 0.0 if pending else poll_s)
        except KeyboardInterrupt:
            break
        if samples is None and not stop:
            if pending:
                finish()  # nothing new to overlap with: deliver what is in flight
            continue
        if pending:
            finish()
        if stop:
            logger.info("Worker received stop signal")
            break
        try:
            dem.submit(samples)
            pending = True
        except Exception as e:
            logger.error(f"Error in DSP loop: {e}")
    if pending:
        finish()


def ring_worker_main(ring_name: str, result_queue, station_id: Optional[int], symbol_length: int, log_level: int) -> None:
    """``worker_main`` with the data queue replaced by a shared-memory ring (``rtldavis_amd.ring.BlockRing``): the
    ``Process`` target for a runner whose producer calls ``ring.put(samples)`` where runners/rtlsdr.py:100-103 calls
    ``data_queue.put(samples)``."""
    from .ring import BlockRing
    ring = BlockRing.attach(ring_name)
    try:
        ring_worker_loop(ring, result_queue, reference_parser_factory(station_id, symbol_length), log_level)
    finally:
        ring.close()


def ring_worker_loop(ring, result_queue, parser_factory: Callable[[], object], log_level: int = logging.INFO,
                     poll_s: float = 1.0) -> None:
    """The loop of worker.worker_main (worker.py:18-58) on blocks that lie in a shared-memory ring: no pickle, no
    pipe, and no copy on this side either - the ring's segment is registered with the device
    (``Demodulator.register_input``) and every block is launched on where the producer wrote it (``submit_from``).
    As in ``worker_loop`` a block is fetched and parsed before the next one is submitted (``Parser.parse`` reads the
    demodulator's state of the block it was given, protocol.py:304-311); its slot goes back to the producer then.
    ``ring.stop()`` on the producer's side plays the ``None`` sentinel (worker.py:40-42)."""
    from .ring import KIND_C128, STOP
    logging.basicConfig(level=log_level, format="%(asctime)s - %(name)s - %(levelname)s - %(message)s")
    logger = logging.getLogger("rtldavis.worker")
    logger.info("DSP worker process started (shared-memory ring)")
    try:
        p = parser_factory()
        dem = p.demodulator
        dem.register_input(ring.data)
    except Exception as e:  # worker.py:30-32
        logger.exception(f"Failed to initialize worker: {e}")
        return
    pending = False  # a block is on the GPU, its packets not yet fetched, its slot not yet released

    def finish() -> None:
        nonlocal pending
        try:
            for msg in p.parse(dem.fetch()):
                result_queue.put(msg)
        except Exception as e:  # worker.py:56-58: log, drop the block, go on
            logger.error(f"Error in DSP loop: {e}")
            try:
                while getattr(dem, "inflight", 0):
                    dem.fetch()
            except Exception:
                try:
                    dem.reset()
                except Exception as e2:
                    logger.error(f"Error in DSP loop: demodulator reset failed: {e2}")
        ring.release()
        pending = False

    try:
        while True:
            try:
                item = ring.get(taken=1 if pending else 0, timeout=0.0 if pending else poll_s)
            except KeyboardInterrupt:
                break
            if pending:
                finish()       # (a block that arrived meanwhile waits in its slot: nothing is lost)
            if item is STOP:
                # (blocks committed before stop() are still handed out first: get() says STOP only when drained)
                logger.info("Worker received stop signal")
                break
            if item is None:
                continue
            _slot, offset, kind, count = item
            try:
                dem.submit_from(offset, count, kind == KIND_C128)
                pending = True
            except Exception as e:
                logger.error(f"Error in DSP loop: {e}")
                ring.release()
        if pending:
            finish()
    finally:
        try:
            dem.register_input(None)
        except Exception:
            pass


class _StreamView:
    """What ``Parser.parse`` reads from its demodulator (protocol.py:304-311), for one stream of a
    ``MultiDemodulator``."""

    def __init__(self, multi: dsp.MultiDemodulator, stream: int) -> None:
        self._multi, self._stream = multi, stream
        self.cfg = multi.cfg

    @property
    def discriminated(self) -> np.ndarray:
        return self._multi.discriminated(self._stream)


def multi_worker_main(data_queues: Sequence, result_queue, parser_factory: Callable[[int], object],
                      log_level: int = logging.INFO, poll_s: float = 1.0) -> None:
    """Several receivers, one GPU launch per round: every ``data_queue`` delivers one block per
    round (lock step, like dongles clocked alike); the blocks are demodulated together by one
    ``MultiDemodulator`` and parsed per stream; messages go to ``result_queue`` as ``(stream, msg)``.
    A ``None`` on any queue stops the worker.  ``parser_factory(k)`` supplies stream k's parser; its
    ``demodulator`` attribute is replaced by a view onto stream k of the shared demodulator.
    """
    logging.basicConfig(level=log_level, format="%(asctime)s - %(name)s - %(levelname)s - %(message)s")
    logger = logging.getLogger("rtldavis.worker")
    n = len(data_queues)
    try:
        parsers = [parser_factory(k) for k in range(n)]
        multi = dsp.MultiDemodulator(parsers[0].cfg, n)
        for k, p in enumerate(parsers):
            p.demodulator = _StreamView(multi, k)
    except Exception as e:
        logger.exception(f"Failed to initialize worker: {e}")
        return
    logger.info(f"DSP worker started for {n} receivers")
    bs2 = 2 * multi.cfg.block_size
    while True:
        blocks: List[Optional[np.ndarray]] = []
        stop = False
        for q in data_queues:
            while True:
                samples = None
                try:
                    samples, stop = _get(q, poll_s)
                except KeyboardInterrupt:
                    stop = True
                if samples is not None or stop:
                    break
            if stop:
                break
            blocks.append(samples)
        if stop:
            logger.info("Worker received stop signal")
            break
        try:
            arr = np.stack([np.ascontiguousarray(b, dtype=np.uint8).reshape(bs2) for b in blocks])
            per_stream = multi.demodulate(arr)
            for k, p in enumerate(parsers):
                for msg in p.parse(per_stream[k]):
                    result_queue.put((k, msg))
        except Exception as e:
            logger.error(f"Error in DSP loop: {e}")
            continue
