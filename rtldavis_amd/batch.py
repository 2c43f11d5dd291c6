"""Batched demodulation: many independent IQ streams (one per hop channel / station /
capture) in one pass on one GPU.  Each stream gets exactly the per-call packet lists the
reference's ``Demodulator.demodulate`` returns when fed its blocks one by one from reset
(src/rtldavis/dsp.py:139-169)."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import numpy as np

from . import _lib
from .dsp import Packet, PacketConfig, _cfg_struct


RD_PACKET_DTYPE = np.dtype([("stream", "<i4"), ("call", "<i4"), ("index", "<i4"), ("nbytes", "<i4"),
                            ("data", "u1", (_lib.RD_MAX_PKT_BYTES,)), ("rssi", "<f8"), ("snr", "<f8")])
assert RD_PACKET_DTYPE.itemsize == C.sizeof(_lib.RdPacket)


RD_PARSED_DTYPE = np.dtype([("stream", "<i4"), ("call", "<i4"), ("index", "<i4"), ("freq_err", "<i4"),
                            ("id", "<i4"), ("nbytes", "<i4"), ("data", "u1", (_lib.RD_MAX_PKT_BYTES,)),
                            ("rssi", "<f8"), ("snr", "<f8")])
assert RD_PARSED_DTYPE.itemsize == C.sizeof(_lib.RdParsed)


class BatchDemodulator:
    def __init__(self, cfg: PacketConfig, n_streams: int, n_blocks: int, device: Optional[int] = None) -> None:
        self.cfg = cfg
        self.n_streams = int(n_streams)
        self.n_blocks = int(n_blocks)
        self.n_samples = self.n_blocks * cfg.block_size
        self._b = C.c_void_p()
        if device is not None:
            _lib.check(_lib.lib().rd_set_device(int(device)))
        _lib.check(_lib.lib().rd_batch_create(C.byref(_cfg_struct(cfg)), self.n_streams, self.n_blocks, C.byref(self._b)))
        self._cap = 0
        self._recs = None

    def close(self) -> None:
        """Free the device buffers now (waits for a run still in flight); the object cannot be used afterwards."""
        if self._b:
            b, self._b = self._b, None
            _lib.lib().rd_batch_destroy(b)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- input ----------------------------------------------------------------------------
    def input_ptr(self):
        """(device pointer, nbytes) of the resident uint8 [n_streams][n_samples][2] input."""
        p, n = C.c_void_p(), C.c_size_t()
        _lib.check(_lib.lib().rd_batch_input_ptr(self._b, C.byref(p), C.byref(n)))
        return p.value, n.value

    def upload(self, iq: np.ndarray) -> None:
        """Copy host IQ (uint8, shape [n_streams, 2*n_samples]) to the device."""
        a = np.ascontiguousarray(iq, dtype=np.uint8)
        if a.size != self.n_streams * 2 * self.n_samples:
            raise ValueError("Incompatible array sizes")
        _lib.check(_lib.lib().rd_batch_upload(self._b, a.ctypes.data, a.size))

    def upload_async(self, iq: np.ndarray, hip_stream: int) -> None:
        """``upload`` issued on a copy stream without waiting (rd_batch_upload_async): the next ``run()`` waits for it
        on the device.  ``iq`` must be contiguous uint8 in PINNED memory and stay untouched until that run's results
        have been fetched."""
        if iq.dtype != np.uint8 or not iq.flags["C_CONTIGUOUS"] or iq.size != self.n_streams * 2 * self.n_samples:
            raise ValueError("Incompatible array sizes")
        _lib.check(_lib.lib().rd_batch_upload_async(self._b, iq.ctypes.data, iq.size, C.c_void_p(hip_stream or None)))

    # ---- run ------------------------------------------------------------------------------
    def run(self, hip_stream: int = 0) -> None:
        """Launch the whole path (asynchronous) on a hipStream_t given as an integer handle."""
        _lib.check(_lib.lib().rd_batch_run(self._b, C.c_void_p(hip_stream or None)))

    def set_parse(self, enabled: bool) -> None:
        """Also run the front half of protocol.Parser.parse (protocol.py:282-318) on the device
        for every packet: bit swap, CRC gate, frequency error.  Call before run()."""
        _lib.check(_lib.lib().rd_batch_set_parse(self._b, int(bool(enabled))))

    def parsed(self) -> np.ndarray:
        """CRC-valid messages of the last run (structured array, fields of rd_parsed), sorted by
        (stream, call, reference order): what Parser.parse keeps before sensor decoding."""
        cap = max(1024, 4 * self.n_streams)
        while True:
            buf = (_lib.RdParsed * cap)()
            n = C.c_int(0)
            rc = _lib.lib().rd_batch_parsed(self._b, buf, cap, C.byref(n))
            if rc == _lib.RD_ERR_CAPACITY:
                cap = n.value + 1024
                continue
            _lib.check(rc)
            return np.frombuffer(buf, dtype=RD_PARSED_DTYPE, count=n.value).copy()

    def set_timing(self, level) -> None:
        """0/False: off; 1/True: demod kernel + whole run (3 events); 2: every stage (5 events)."""
        _lib.check(_lib.lib().rd_batch_set_timing(self._b, int(level)))

    def set_pipelined(self, enabled=True) -> None:
        """Pipelined completion (include/rtldavis_hip.h: rd_batch_set_pipelined): for several runs kept queued on one
        stream; a run's results are ready one demod kernel later and the stream never idles between runs."""
        _lib.check(_lib.lib().rd_batch_set_pipelined(self._b, 1 if enabled else 0))

    def timing(self) -> dict:
        t = _lib.RdTiming()
        _lib.check(_lib.lib().rd_batch_get_timing(self._b, C.byref(t)))
        return {k: float(getattr(t, k)) for k, _ in t._fields_}

    def last_run_forms(self) -> dict:
        """Which forms the last run took (rd_batch_last_run_forms): for tests of the opt-in forms."""
        f = C.c_uint32()
        _lib.check(_lib.lib().rd_batch_last_run_forms(self._b, C.byref(f)))
        return {"ordered_tail": bool(f.value & 1), "second_pass": bool(f.value & 8), "one_launch_tail": bool(f.value & 16)}

    def counters(self) -> dict:
        f, m = C.c_uint64(), C.c_uint64()
        _lib.check(_lib.lib().rd_batch_get_counters(self._b, C.byref(f), C.byref(m)))
        return {"fixup_runs": f.value, "matches": m.value}

    # ---- results --------------------------------------------------------------------------
    def results(self) -> np.ndarray:
        """All packets of the last run as one structured array (fields of rd_packet: stream,
        call, index, nbytes, data[32], rssi, snr), sorted by (stream, call, reference order).
        The array is a view of a buffer that the next results() call overwrites."""
        L = _lib.lib()
        n = C.c_int(0)
        if self._recs is None:
            self._cap = max(1024, 16 * self.n_streams)
            self._recs = (_lib.RdPacket * self._cap)()
        rc = L.rd_batch_results(self._b, self._recs, self._cap, C.byref(n))
        if rc == _lib.RD_ERR_CAPACITY:
            self._cap = n.value + 1024
            self._recs = (_lib.RdPacket * self._cap)()
            rc = L.rd_batch_results(self._b, self._recs, self._cap, C.byref(n))
        _lib.check(rc)
        return np.frombuffer(self._recs, dtype=RD_PACKET_DTYPE, count=n.value)

    def records(self):
        """Flat list of (stream, call, Packet) in (stream, call, reference order)."""
        out = []
        for r in self.results():
            data = np.frombuffer(r["data"][: int(r["nbytes"])].tobytes(), dtype=np.uint8)  # read-only copy
            out.append((int(r["stream"]), int(r["call"]),
                        Packet(int(r["index"]), data, float(r["rssi"]), float(r["snr"]))))
        return out

    def packets(self) -> List[List[List[Packet]]]:
        """packets()[stream][call] == what demodulate() returns for that block."""
        res = [[[] for _ in range(self.n_blocks)] for _ in range(self.n_streams)]
        for s, c, p in self.records():
            res[s][c].append(p)
        return res

    def bits(self, stream: int) -> np.ndarray:
        """Packed sign bits of one stream, LSB first (sample t -> byte t//8, bit t%8)."""
        out = np.empty((self.n_samples + 7) // 8, dtype=np.uint8)
        _lib.check(_lib.lib().rd_batch_copy_bits(self._b, int(stream), out.ctypes.data, out.size))
        return out

    def discriminated(self, stream: int, t0: int, n: int) -> np.ndarray:
        """float64 discriminator output d[t0:t0+n] of one stream (dsp.py:76-90)."""
        out = np.empty(int(n), dtype=np.float64)
        _lib.check(_lib.lib().rd_batch_copy_discriminated(self._b, int(stream), int(t0), out.ctypes.data, out.size))
        return out

    def demodulate(self, iq: np.ndarray) -> List[List[List[Packet]]]:
        """upload + run + packets in one call."""
        self.upload(iq)
        self.run()
        return self.packets()
