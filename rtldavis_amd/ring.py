"""A shared-memory ring of IQ blocks between the process that owns the SDR and the DSP worker.

SURVEY section 8f-4: the reference moves every block through a ``multiprocessing.Queue`` - a pickle of the ndarray, a
pipe write, a pipe read and an unpickle per block (/root/reference/src/rtldavis/runners/rtlsdr.py:100-103
``data_queue.put(samples)`` -> /root/reference/src/rtldavis/worker.py:37 ``data_queue.get()``; 16 KB per uint8 block,
128 KB per complex128 block).  Here the producer writes the block once into a slot of a ``multiprocessing.shared_memory``
segment; the worker registers that segment with the device (``Demodulator.register_input``) and launches on the slot
where it lies (``submit_from``): no pickle, no pipe, no copy on the consumer side.

One producer, one consumer.  Layout: a 128-byte header (magic, geometry, the two sequence numbers, a stop flag), one
16-byte descriptor per slot (kind, element count), then the slots, 4096-byte aligned, each a multiple of 64 bytes.
``wr`` counts committed blocks, ``rd`` released ones; the producer owns slot ``wr % n`` while ``wr - rd < n``, the
consumer reads slots ``rd % n .. (wr - 1) % n``.  A slot is released only after the block's packets have been fetched:
the GPU reads it in place.  No lock: each counter has one writer, and on the x86-64 hosts this runs on stores become
visible in program order (the slot's bytes before the counter that publishes them).
"""
from __future__ import annotations

import time
from multiprocessing import shared_memory
from typing import Optional, Tuple

import numpy as np

MAGIC = 0x52444252  # "RDBR"
HEADER = 128
KIND_U8, KIND_C128 = 0, 1
STOP = object()     # what get() returns once the producer has called stop() and the ring is drained


class BlockRing:
    def __init__(self, shm: shared_memory.SharedMemory, owner: bool) -> None:
        self._shm, self._owner = shm, owner
        self._hdr = np.ndarray(8, dtype=np.uint32, buffer=shm.buf, offset=0)         # magic, version, n_slots, slot_bytes, block_size, stop, -, -
        self._seq = np.ndarray(2, dtype=np.uint64, buffer=shm.buf, offset=64)        # wr, rd
        if int(self._hdr[0]) != MAGIC:
            raise ValueError("not a BlockRing segment")
        self.n_slots, self.slot_bytes, self.block_size = int(self._hdr[2]), int(self._hdr[3]), int(self._hdr[4])
        self._desc = np.ndarray((self.n_slots, 4), dtype=np.uint32, buffer=shm.buf, offset=HEADER)
        self.data_offset = -(-(HEADER + 16 * self.n_slots) // 4096) * 4096
        self.data = np.ndarray(self.n_slots * self.slot_bytes, dtype=np.uint8, buffer=shm.buf, offset=self.data_offset)

    # ---- construction ----------------------------------------------------------------------------------------------
    @classmethod
    def create(cls, n_slots: int = 8, block_size: int = 8192, name: Optional[str] = None) -> "BlockRing":
        """A ring of ``n_slots`` blocks of ``block_size`` samples; a slot holds the larger form (complex128: 16 bytes per
        sample).  The creating process unlinks the segment in ``close()``."""
        if n_slots < 2 or block_size < 1:
            raise ValueError("n_slots >= 2 and block_size >= 1")
        slot_bytes = -(-16 * block_size // 64) * 64
        data_offset = -(-(HEADER + 16 * n_slots) // 4096) * 4096
        shm = shared_memory.SharedMemory(create=True, size=data_offset + n_slots * slot_bytes, name=name)
        hdr = np.ndarray(8, dtype=np.uint32, buffer=shm.buf, offset=0)
        hdr[:] = (MAGIC, 1, n_slots, slot_bytes, block_size, 0, 0, 0)
        np.ndarray(2, dtype=np.uint64, buffer=shm.buf, offset=64)[:] = 0
        return cls(shm, owner=True)

    @classmethod
    def attach(cls, name: str) -> "BlockRing":
        return cls(shared_memory.SharedMemory(name=name), owner=False)

    @property
    def name(self) -> str:
        return self._shm.name

    def close(self) -> None:
        # (the numpy views hold the segment's buffer: drop them before the mapping goes away)
        self._hdr = self._seq = self._desc = self.data = None
        try:
            self._shm.close()
        except BufferError:
            pass  # a caller still holds a view of a slot: the mapping lives until that view dies
        if self._owner:
            try:
                self._shm.unlink()
            except FileNotFoundError:
                pass

    # ---- producer --------------------------------------------------------------------------------------------------
    def claim(self, timeout: Optional[float] = None) -> Optional[np.ndarray]:
        """The next free slot as a uint8 view (``slot_bytes`` long), or None when the ring stayed full for ``timeout``
        seconds (None: wait for ever).  Fill it, then ``commit``."""
        t0 = time.monotonic()
        while int(self._seq[0]) - int(self._seq[1]) >= self.n_slots:
            if timeout is not None and time.monotonic() - t0 >= timeout:
                return None
            time.sleep(0.0002)
        i = int(self._seq[0]) % self.n_slots
        return self.data[i * self.slot_bytes: (i + 1) * self.slot_bytes]

    def commit(self, kind: int, count: int) -> None:
        """Publish the claimed slot: ``count`` elements of ``kind`` (KIND_U8: bytes, KIND_C128: complex128 samples)."""
        i = int(self._seq[0]) % self.n_slots
        self._desc[i, 0], self._desc[i, 1] = kind, count
        self._seq[0] = self._seq[0] + np.uint64(1)

    def put(self, samples: np.ndarray, timeout: Optional[float] = None) -> bool:
        """``data_queue.put(samples)`` of runners/rtlsdr.py:100-103: one copy into the ring, no pickle.  Complex input
        of any precision is stored as complex128 (what dsp.py:144-150 copies into ``raw_samples``)."""
        slot = self.claim(timeout)
        if slot is None:
            return False
        a = np.asarray(samples)
        if np.iscomplexobj(a):
            n = a.size
            if 16 * n > self.slot_bytes:
                raise ValueError("Incompatible array sizes")
            slot[: 16 * n].view(np.complex128)[:] = a.reshape(-1)
            self.commit(KIND_C128, n)
        else:
            n = a.size
            if n > self.slot_bytes:
                raise ValueError("Incompatible array sizes")
            slot[:n] = a.reshape(-1).astype(np.uint8, copy=False)
            self.commit(KIND_U8, n)
        return True

    def stop(self) -> None:
        """The ``data_queue.put(None)`` of the reference's shutdown (worker.py:40-42): the consumer's get() returns STOP
        once every committed block has been handed out."""
        self._hdr[5] = 1

    # ---- consumer --------------------------------------------------------------------------------------------------
    def get(self, taken: int = 0, timeout: Optional[float] = None) -> object:
        """The oldest committed block the consumer has not taken yet: ``(slot, byte offset into .data, kind, count)``.
        ``taken``: blocks already handed out and not yet released (the consumer keeps up to two on the GPU).  None on
        timeout, STOP after stop() once the ring holds nothing new."""
        t0 = time.monotonic()
        while True:
            nxt = int(self._seq[1]) + taken
            if nxt < int(self._seq[0]):
                i = nxt % self.n_slots
                return i, i * self.slot_bytes, int(self._desc[i, 0]), int(self._desc[i, 1])
            if int(self._hdr[5]):
                return STOP
            if timeout is not None and time.monotonic() - t0 >= timeout:
                return None
            time.sleep(0.0002)

    def release(self) -> None:
        """Give the oldest taken slot back to the producer (after its block's packets have been fetched)."""
        self._seq[1] = self._seq[1] + np.uint64(1)

    def slot_view(self, slot: int, kind: int, count: int) -> np.ndarray:
        """The block in ``slot`` as the array the producer stored."""
        raw = self.data[slot * self.slot_bytes: (slot + 1) * self.slot_bytes]
        return raw[: 16 * count].view(np.complex128) if kind == KIND_C128 else raw[:count]

    @property
    def backlog(self) -> Tuple[int, int]:
        """(committed, released) block counts."""
        return int(self._seq[0]), int(self._seq[1])
