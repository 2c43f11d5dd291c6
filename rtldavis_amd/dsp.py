"""Drop-in mirror of rtldavis's ``dsp`` module backed by hand-written HIP kernels (gfx950).

Same names, argument meaning and error behaviour as /root/reference/src/rtldavis/dsp.py so
``protocol.Parser`` and ``worker.worker_main`` run unchanged on top of it:

    Packet, ByteToCmplxLUT, rotate_fs4, fir9, discriminate, quantize, PacketConfig, Demodulator

Every function executes on the GPU through the C ABI of librtldavis_hip.so; nothing here
computes the path on the CPU.  The HIP context is created lazily by the first
``demodulate()`` (not by ``__init__``) because the reference builds one Demodulator in
the parent and one in a ``fork``ed worker (runners/rtlsdr.py:30, worker.py:29).
"""
from __future__ import annotations

import ctypes as C
import logging
import math
import os
from dataclasses import dataclass
from typing import List, Optional

import numpy as np

from . import _lib

logger = logging.getLogger(__name__)


@dataclass
class Packet:
    """dsp.Packet (dsp.py:12-17)."""

    index: int
    data: np.ndarray
    rssi: float
    snr: float


def _c128(a: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.complex128)


class ByteToCmplxLUT:
    """dsp.ByteToCmplxLUT (dsp.py:20-39): (k - 127.4) / 127.6 per byte, on the device."""

    def __init__(self) -> None:
        # kept for attribute compatibility (dsp.py:26); execute() does not read it
        self.lut: np.ndarray = (np.arange(256, dtype=np.float64) - 127.4) / 127.6

    def execute(self, in_bytes: np.ndarray, out_cmplx: np.ndarray) -> None:
        if in_bytes.size != out_cmplx.size * 2:
            logger.error(
                f"Incompatible array sizes: in_bytes.size={in_bytes.size}, out_cmplx.size={out_cmplx.size}"
            )
            raise ValueError("Incompatible array sizes")
        src = np.ascontiguousarray(in_bytes, dtype=np.uint8)
        tmp = np.empty(out_cmplx.size, dtype=np.complex128)
        _lib.check(_lib.lib().rd_lut_execute(src.ctypes.data, src.size, tmp.ctypes.data, tmp.size))
        out_cmplx[...] = tmp


def rotate_fs4(in_cmplx: np.ndarray, out_cmplx: np.ndarray) -> None:
    """dsp.rotate_fs4 (dsp.py:42-49).  in and out may be the same array."""
    src = _c128(in_cmplx)
    tmp = np.empty(src.size, dtype=np.complex128)
    _lib.check(_lib.lib().rd_rotate_fs4(src.ctypes.data, tmp.ctypes.data, src.size))
    out_cmplx[...] = tmp


def fir9(in_cmplx: np.ndarray, out_cmplx: np.ndarray) -> None:
    """dsp.fir9 (dsp.py:52-73): the first out.size 'valid' outputs of the 9-tap filter."""
    src = _c128(in_cmplx)
    n = out_cmplx.size
    tmp = np.empty(n, dtype=np.complex128)
    _lib.check(_lib.lib().rd_fir9(src.ctypes.data, src.size, tmp.ctypes.data, n))
    out_cmplx[:] = tmp


def discriminate(in_cmplx: np.ndarray, out_float: np.ndarray) -> None:
    """dsp.discriminate (dsp.py:76-90)."""
    src = _c128(in_cmplx)
    n = max(src.size - 1, 0)
    tmp = np.empty(n, dtype=np.float64)
    _lib.check(_lib.lib().rd_discriminate(src.ctypes.data, src.size, tmp.ctypes.data, n))
    out_float[:n] = tmp


def quantize(in_float: np.ndarray, out_byte: np.ndarray) -> None:
    """dsp.quantize (dsp.py:93-98): IEEE-754 sign bit of each value."""
    src = np.ascontiguousarray(in_float, dtype=np.float64)
    tmp = np.empty(src.size, dtype=np.uint8)
    _lib.check(_lib.lib().rd_quantize(src.ctypes.data, tmp.ctypes.data, src.size))
    out_byte[: src.size] = tmp


class PacketConfig:
    """dsp.PacketConfig (dsp.py:101-125), attribute for attribute."""

    def __init__(self, bit_rate: int, symbol_length: int, preamble_symbols: int, packet_symbols: int,
                 preamble: str, block_size: int = 512) -> None:
        self.bit_rate = bit_rate
        self.symbol_length = symbol_length
        self.preamble_symbols = preamble_symbols
        self.packet_symbols = packet_symbols
        self.preamble = preamble
        self.preamble_bytes = np.array([int(b) for b in preamble], dtype=np.uint8)
        self.preamble_str = self.preamble_bytes.tobytes()
        self.sample_rate = self.bit_rate * self.symbol_length
        self.block_size = block_size
        self.block_size2 = self.block_size * 2
        self.preamble_length = self.preamble_symbols * self.symbol_length
        self.packet_length = self.packet_symbols * self.symbol_length
        self.buffer_length = (self.packet_length // self.block_size + 2) * self.block_size

    def _c(self) -> _lib.RdConfig:
        return _lib.make_config(self.bit_rate, self.symbol_length, self.preamble_symbols, self.packet_symbols,
                                self.preamble, self.block_size)


def _cfg_struct(cfg) -> _lib.RdConfig:
    """rd_config from any PacketConfig-like object (ours or the reference's dsp.PacketConfig)."""
    return _lib.make_config(cfg.bit_rate, cfg.symbol_length, cfg.preamble_symbols, cfg.packet_symbols,
                            cfg.preamble, cfg.block_size)


def _packets_from(recs, n: int) -> List[Packet]:
    out = []
    for i in range(n):
        r = recs[i]
        data = np.frombuffer(bytes(r.data[: r.nbytes]), dtype=np.uint8)  # read-only, like dsp.py:241
        snr = float(r.snr)
        if snr == -math.inf:
            # signal_power == 0 over a non-silent noise estimate: the reference's math.log10(0) raises out of
            # demodulate() at this packet (dsp.py:231-236); the device reports the same condition as -inf
            raise ValueError("math domain error")
        out.append(Packet(index=int(r.index), data=data, rssi=float(r.rssi), snr=snr))
    return out


class Demodulator:
    """dsp.Demodulator (dsp.py:128-253) on the GPU.

    ``demodulate(block)`` takes exactly one block (uint8[2*block_size] interleaved I,Q or
    complex[block_size]) and returns the reference's ``List[Packet]``.  ``discriminated``,
    ``filtered`` and ``quantized`` are materialised from device state on access.
    """

    def __init__(self, cfg: PacketConfig) -> None:
        self.cfg = cfg
        self._h = C.c_void_p()
        self._pid = os.getpid()
        _lib.check(_lib.lib().rd_create(C.byref(_cfg_struct(cfg)), C.byref(self._h)))  # host state only
        self._cap = 64
        self._recs = (_lib.RdPacket * self._cap)()
        self.byte_to_cmplx = ByteToCmplxLUT()

    def __del__(self):
        try:
            if self._h and self._pid == os.getpid():
                _lib.lib().rd_destroy(self._h)
        except Exception:
            pass

    def _handle(self):
        if self._pid != os.getpid():
            # forked copy of a handle created in the parent: it holds no device state yet
            # (lazy init), so it is safe to keep using it in the child.
            self._pid = os.getpid()
        return self._h

    def _check_block(self, input_data: np.ndarray):
        """Size checks of dsp.py:32-36,145-149 and a contiguous buffer of the right type."""
        bs = self.cfg.block_size
        if np.iscomplexobj(input_data):
            if input_data.size != bs:
                logger.error(f"Incompatible array sizes: input_data.size={input_data.size}, dest.size={bs}")
                raise ValueError("Incompatible array sizes")
            buf = _c128(input_data)
            count, is_c = buf.size, 1
        else:
            if input_data.size != 2 * bs:
                logger.error(f"Incompatible array sizes: in_bytes.size={input_data.size}, out_cmplx.size={bs}")
                raise ValueError("Incompatible array sizes")
            buf = np.ascontiguousarray(input_data, dtype=np.uint8)
            count, is_c = buf.size, 0
        return buf, count, is_c

    def _take(self, rc: int, n: "C.c_int") -> List[Packet]:
        """Packets of a finished call.  The reference's list is unbounded (dsp.py:190-246): when the
        record array was too small the block has still been consumed and its packets are kept by the
        handle - grow the array and fetch them again."""
        if rc == _lib.RD_ERR_CAPACITY:
            self._cap = max(2 * self._cap, n.value)
            self._recs = (_lib.RdPacket * self._cap)()
            rc = _lib.lib().rd_demod_refetch(self._handle(), self._recs, self._cap, C.byref(n))
        _lib.check(rc)
        return _packets_from(self._recs, n.value)

    def demodulate(self, input_data: np.ndarray) -> List[Packet]:
        buf, count, is_c = self._check_block(input_data)
        n = C.c_int(0)
        rc = _lib.lib().rd_demod_block(self._handle(), buf.ctypes.data, count, is_c, self._recs, self._cap, C.byref(n))
        return self._take(rc, n)

    # --- the same call in two halves (worker.py:34-58 without the wait) ---
    def submit(self, input_data: np.ndarray) -> None:
        """Queue one block: host-to-device copy and kernels run asynchronously; at most two blocks
        may be in flight.  ``fetch()`` returns their packets in submission order."""
        buf, count, is_c = self._check_block(input_data)
        _lib.check(_lib.lib().rd_demod_submit(self._handle(), buf.ctypes.data, count, is_c))

    def fetch(self) -> List[Packet]:
        """Packets of the oldest block in flight (waits for it)."""
        n = C.c_int(0)
        rc = _lib.lib().rd_demod_fetch(self._handle(), self._recs, self._cap, C.byref(n))
        return self._take(rc, n)

    # --- zero-copy input: blocks that already lie in a producer's buffer ---
    def register_input(self, buf: Optional[np.ndarray]) -> None:
        """Pin a producer-owned uint8 buffer (e.g. ``rtldavis_amd.ring.BlockRing.data``) and map it into the device;
        ``submit_from`` then launches on blocks inside it without copying them.  None unregisters.  The buffer must
        outlive the registration."""
        if buf is None:
            _lib.check(_lib.lib().rd_demod_register_input(self._handle(), None, 0))
            self._ext = None
            return
        a = np.asarray(buf)
        if a.dtype != np.uint8 or not a.flags["C_CONTIGUOUS"]:
            raise ValueError("register_input needs a contiguous uint8 buffer")
        _lib.check(_lib.lib().rd_demod_register_input(self._handle(), a.ctypes.data, a.size))
        self._ext = a   # (keeps the mapping alive)

    def submit_from(self, offset: int, count: int, is_complex: bool = False) -> None:
        """``submit`` for the block at ``offset`` bytes into the registered buffer: ``count`` = 2 * block_size bytes, or
        block_size complex128 samples.  The block must stay untouched until its ``fetch()`` has returned."""
        if count != (self.cfg.block_size if is_complex else 2 * self.cfg.block_size):
            logger.error(f"Incompatible array sizes: count={count}, block_size={self.cfg.block_size}")
            raise ValueError("Incompatible array sizes")
        _lib.check(_lib.lib().rd_demod_submit_from(self._handle(), int(offset), int(count), 1 if is_complex else 0))

    @property
    def inflight(self) -> int:
        return int(_lib.lib().rd_demod_inflight(self._handle()))

    @property
    def input_pushed(self) -> bool:
        """True when this handle's copied blocks are written by the host straight into device memory (PCIe large BAR),
        False when the kernel reads them from a pinned host slot (``set_input_push``)."""
        rc = int(_lib.lib().rd_demod_input_mode(self._handle()))
        if rc < 0:
            _lib.check(rc)
        return rc == 1

    def reset(self) -> None:
        _lib.check(_lib.lib().rd_reset(self._handle()))

    @property
    def discriminated(self) -> np.ndarray:
        out = np.empty(2 * self.cfg.block_size, dtype=np.float64)
        _lib.check(_lib.lib().rd_copy_discriminated(self._handle(), out.ctypes.data, out.size))
        return out

    @property
    def filtered(self) -> np.ndarray:
        out = np.empty(self.cfg.block_size + 1, dtype=np.complex128)
        _lib.check(_lib.lib().rd_copy_filtered(self._handle(), out.ctypes.data, out.size))
        return out

    @property
    def quantized(self) -> np.ndarray:
        out = np.empty(self.cfg.buffer_length, dtype=np.uint8)
        _lib.check(_lib.lib().rd_copy_quantized(self._handle(), out.ctypes.data, out.size))
        return out

    def _search(self) -> List[int]:
        """dsp.Demodulator._search (dsp.py:171-188) over the current quantized buffer."""
        return search(self.quantized, self.cfg)


class MultiDemodulator:
    """Several Demodulators in lock step on one GPU: one block per stream per call, all streams
    in one set of kernel launches (one per SDR / hop channel).  Each stream
    behaves exactly like its own ``Demodulator`` (dsp.py:128-253)."""

    def __init__(self, cfg: PacketConfig, n_streams: int) -> None:
        self.cfg = cfg
        self.n_streams = int(n_streams)
        self._h = C.c_void_p()
        _lib.check(_lib.lib().rd_create_multi(C.byref(_cfg_struct(cfg)), self.n_streams, C.byref(self._h)))
        self._cap = 64 * self.n_streams
        self._recs = (_lib.RdPacket * self._cap)()

    def __del__(self):
        try:
            if self._h:
                _lib.lib().rd_destroy(self._h)
        except Exception:
            pass

    def _check_blocks(self, blocks: np.ndarray) -> np.ndarray:
        a = np.ascontiguousarray(blocks, dtype=np.uint8)
        if a.size != self.n_streams * 2 * self.cfg.block_size:
            logger.error(f"Incompatible array sizes: blocks.size={a.size}")
            raise ValueError("Incompatible array sizes")
        return a

    def _take(self, rc: int, n: "C.c_int") -> List[List[Packet]]:
        if rc == _lib.RD_ERR_CAPACITY:  # nothing is lost: the handle keeps the block's packets (dsp.py:190-246)
            self._cap = max(2 * self._cap, n.value)
            self._recs = (_lib.RdPacket * self._cap)()
            rc = _lib.lib().rd_demod_refetch(self._h, self._recs, self._cap, C.byref(n))
        _lib.check(rc)
        out: List[List[Packet]] = [[] for _ in range(self.n_streams)]
        for i in range(n.value):
            r = self._recs[i]
            data = np.frombuffer(bytes(r.data[: r.nbytes]), dtype=np.uint8)
            snr = float(r.snr)
            if snr == -math.inf:
                # as Demodulator._packets_from: the reference's math.log10(0) raises out of that stream's
                # demodulate() (dsp.py:231-236).  One exception for the whole round: the streams run in lock step.
                raise ValueError("math domain error")
            out[r.stream].append(Packet(int(r.index), data, float(r.rssi), snr))
        return out

    def demodulate(self, blocks: np.ndarray) -> List[List[Packet]]:
        """blocks: uint8 [n_streams, 2*block_size].  Returns one ``List[Packet]`` per stream."""
        a = self._check_blocks(blocks)
        n = C.c_int(0)
        rc = _lib.lib().rd_demod_blocks(self._h, a.ctypes.data, a.size, self._recs, self._cap, C.byref(n))
        return self._take(rc, n)

    def submit(self, blocks: np.ndarray) -> None:
        """Queue one block per stream (asynchronous copy + kernels; at most two in flight)."""
        a = self._check_blocks(blocks)
        _lib.check(_lib.lib().rd_demod_submit(self._h, a.ctypes.data, a.size, 0))

    def fetch(self) -> List[List[Packet]]:
        """Packets of the oldest submitted set of blocks, per stream."""
        n = C.c_int(0)
        rc = _lib.lib().rd_demod_fetch(self._h, self._recs, self._cap, C.byref(n))
        return self._take(rc, n)

    @property
    def inflight(self) -> int:
        return int(_lib.lib().rd_demod_inflight(self._h))

    def reset(self) -> None:
        _lib.check(_lib.lib().rd_reset(self._h))

    def discriminated(self, stream: int) -> np.ndarray:
        """``Demodulator.discriminated`` of one stream (read by protocol.py:307-309)."""
        out = np.empty(2 * self.cfg.block_size, dtype=np.float64)
        _lib.check(_lib.lib().rd_copy_discriminated_stream(self._h, int(stream), out.ctypes.data, out.size))
        return out


def set_input_push(on: Optional[bool] = None) -> bool:
    """How the streaming handles created AFTERWARDS get their copied blocks onto the device: None / True = the host
    writes them into device memory where the device allows it (the default), False = a pinned host slot the kernel reads
    across the bus.  Returns whether the push is available on the current device.  Same packets either way."""
    rc = int(_lib.lib().rd_set_input_push(-1 if on is None else (1 if on else 0)))
    if rc < 0:
        _lib.check(rc)
    return rc == 1


def search(quantized: np.ndarray, cfg: PacketConfig) -> List[int]:
    """Preamble search over a 0/1-per-byte buffer in the reference's order (dsp.py:171-188,
    dsp/dsp.go:105-131)."""
    q = np.ascontiguousarray(quantized, dtype=np.uint8)
    idx = np.empty(max(q.size, 1), dtype=np.int32)
    n = C.c_int(0)
    _lib.check(_lib.lib().rd_search(C.byref(_cfg_struct(cfg)), q.ctypes.data, q.size, idx.ctypes.data, idx.size, C.byref(n)))
    return [int(v) for v in idx[: n.value]]
