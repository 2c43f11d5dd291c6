"""CPU oracle for the rtldavis IQ -> bits -> packets path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product (rtldavis_amd/) never does.

This is a NumPy restatement of the reference's algorithm, each function citing
the reference lines it follows (paths relative to /root/reference).  It is pinned
against the real reference by tools/gen_golden.py (which imports
src/rtldavis/dsp.py in the build container and writes tests/golden/) and by
tests/test_oracle_golden.py (which replays those fixtures through this file).

Two formulations are provided and tested against each other:

* ``OracleDemodulator`` - call-for-call mirror of ``dsp.Demodulator``
  (src/rtldavis/dsp.py:128-253): rolling buffers, per-block stages, whole-buffer
  search, slice with per-call dedupe and the reference's RSSI/SNR windows.
* ``demod_stream_oneshot`` + ``calls_from_oneshot`` - the whole-stream
  formulation the GPU path uses (SURVEY.md section 8a "streaming equivalence").
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Sequence, Tuple

import numpy as np

# src/rtldavis/dsp.py:56-69 (Go twin: dsp/dsp.go:65-83)
FIR9_COEFFS = np.array(
    [
        0.017682261285,
        0.048171339939,
        0.122424706672,
        0.197408519126,
        0.228626345955,
        0.197408519126,
        0.122424706672,
        0.048171339939,
        0.017682261285,
    ],
    dtype=np.float64,
)
DISC_EPSILON = 1e-10  # src/rtldavis/dsp.py:88


@dataclass
class OraclePacket:
    """Mirror of dsp.Packet (src/rtldavis/dsp.py:12-17)."""

    index: int
    data: np.ndarray
    rssi: float
    snr: float


class OracleConfig:
    """Derived constants of dsp.PacketConfig (src/rtldavis/dsp.py:101-125)."""

    def __init__(self, bit_rate=19200, symbol_length=14, preamble_symbols=16,
                 packet_symbols=80, preamble="1100101110001001", block_size=512):
        self.bit_rate = bit_rate
        self.symbol_length = symbol_length
        self.preamble_symbols = preamble_symbols
        self.packet_symbols = packet_symbols
        self.preamble = preamble
        self.preamble_bytes = np.array([int(b) for b in preamble], dtype=np.uint8)
        self.preamble_str = self.preamble_bytes.tobytes()
        self.sample_rate = bit_rate * symbol_length
        self.block_size = block_size
        self.block_size2 = block_size * 2
        self.preamble_length = preamble_symbols * symbol_length
        self.packet_length = packet_symbols * symbol_length
        self.buffer_length = (self.packet_length // block_size + 2) * block_size


def production_config(symbol_length: int = 14) -> OracleConfig:
    """protocol.new_packet_config (src/rtldavis/protocol.py:68-76)."""
    return OracleConfig(19200, symbol_length, 16, 80, "1100101110001001", 8192)


# ----------------------------------------------------------------------------
# stage functions
# ----------------------------------------------------------------------------

def lut_table() -> np.ndarray:
    """ByteToCmplxLUT.__init__ (src/rtldavis/dsp.py:25-26; dsp/dsp.go:28-33)."""
    return (np.arange(256, dtype=np.float64) - 127.4) / 127.6


def byte_to_cmplx(in_bytes: np.ndarray) -> np.ndarray:
    """ByteToCmplxLUT.execute (src/rtldavis/dsp.py:28-39)."""
    if in_bytes.size % 2:
        raise ValueError("Incompatible array sizes")
    lut = lut_table()
    out = np.empty(in_bytes.size // 2, dtype=np.complex128)
    out.real = lut[in_bytes[0::2]]
    out.imag = lut[in_bytes[1::2]]
    return out


def rotate_fs4(x: np.ndarray) -> np.ndarray:
    """rotate_fs4 (src/rtldavis/dsp.py:42-49): y[n] = x[n] * j**(n mod 4).

    The phase restarts at the beginning of ``x``; block sizes are multiples of
    four so per-block rotation is continuous across a stream.
    """
    y = np.array(x, dtype=np.complex128, copy=True)
    y[1::4] = x[1::4] * 1j
    y[2::4] = x[2::4] * -1
    y[3::4] = x[3::4] * -1j
    return y


def fir9(iq: np.ndarray, n_out: int) -> np.ndarray:
    """fir9 (src/rtldavis/dsp.py:52-73): first n_out 'valid' outputs."""
    return np.convolve(iq, FIR9_COEFFS, mode="valid")[:n_out]


def discriminate(filt: np.ndarray) -> np.ndarray:
    """discriminate (src/rtldavis/dsp.py:76-90); len(out) = len(filt) - 1."""
    n = filt[:-1]
    np_ = filt[1:]
    return (n.imag * np_.real - n.real * np_.imag) / (n.real ** 2 + n.imag ** 2 + DISC_EPSILON)


def quantize(d: np.ndarray) -> np.ndarray:
    """quantize (src/rtldavis/dsp.py:93-98): the IEEE-754 sign bit (-0.0 -> 1)."""
    return np.signbit(np.asarray(d, dtype=np.float64)).astype(np.uint8)


def quantize_per_sample(d: np.ndarray) -> np.ndarray:
    """quantize as the reference executes it (src/rtldavis/dsp.py:93-98): one Python-level step per
    sample that takes the top bit of the IEEE-754 pattern.  Same result as ``quantize``; it exists so
    that the CPU baseline's per-block leg has the reference's cost structure (this loop is 86 % of the
    reference's run time)."""
    import struct
    out = np.empty(len(d), dtype=np.uint8)
    for i, v in enumerate(d):
        out[i] = struct.unpack("<Q", struct.pack("<d", v))[0] >> 63
    return out


def search(quantized: np.ndarray, cfg: OracleConfig) -> List[int]:
    """Demodulator._search (src/rtldavis/dsp.py:171-188): phase-major order."""
    out: List[int] = []
    s = cfg.symbol_length
    for offset in range(s):
        view = quantized[offset::s].tobytes()
        start = 0
        while True:
            idx = view.find(cfg.preamble_str, start)
            if idx == -1:
                break
            out.append(idx * s + offset)
            start = idx + 1
    return out


def slice_bytes(quantized: np.ndarray, q_idx: int, cfg: OracleConfig) -> bytes:
    """Bit packing of Demodulator._slice (src/rtldavis/dsp.py:197-202)."""
    sym = quantized[q_idx: q_idx + cfg.packet_symbols * cfg.symbol_length: cfg.symbol_length]
    sym = sym[: cfg.packet_symbols]
    nbytes = (cfg.packet_symbols + 7) // 8
    pkt = bytearray(nbytes)
    for i, b in enumerate(sym):
        pkt[i >> 3] = ((pkt[i >> 3] << 1) | int(b)) & 0xFF
    return bytes(pkt)


def rssi_snr(filtered: np.ndarray, q_idx: int, cfg: OracleConfig) -> Tuple[float, float]:
    """RSSI/SNR of Demodulator._slice (src/rtldavis/dsp.py:207-236).

    ``filtered`` is the reference's newest-block-only buffer (B+1 samples) while
    ``q_idx`` indexes the whole search window; the resulting mis-aligned window
    is observable behaviour and is reproduced, not fixed.  An empty preamble
    window gives NaN from np.mean, exactly as the reference does.
    """
    noise_start = max(0, q_idx - cfg.preamble_length)
    if q_idx > noise_start:
        noise_power = float(np.mean(np.abs(filtered[noise_start:q_idx]) ** 2))
    else:
        noise_power = 1e-9
    pre = filtered[q_idx: q_idx + cfg.preamble_length]
    with np.errstate(all="ignore"):
        signal_power = float(np.mean(np.abs(pre) ** 2)) if pre.size else float("nan")
    rssi = 10 * math.log10(signal_power) if signal_power > 0 else -120
    snr = 10 * math.log10(signal_power / noise_power) if noise_power > 0 else 50
    return rssi, snr


# ----------------------------------------------------------------------------
# call-for-call mirror
# ----------------------------------------------------------------------------

class OracleDemodulator:
    """Mirror of dsp.Demodulator (src/rtldavis/dsp.py:128-253)."""

    def __init__(self, cfg: OracleConfig, per_sample_quantize: bool = False):
        self.cfg = cfg
        self._quantize = quantize_per_sample if per_sample_quantize else quantize
        self.reset()

    def reset(self) -> None:
        """src/rtldavis/dsp.py:129-137,248-253."""
        c = self.cfg
        self.iq = np.zeros(c.block_size + 9, dtype=np.complex128)
        self.filtered = np.zeros(c.block_size + 1, dtype=np.complex128)
        self.discriminated = np.zeros(c.block_size * 2, dtype=np.float64)
        self.quantized = np.zeros(c.buffer_length, dtype=np.uint8)

    def demodulate(self, input_data: np.ndarray) -> List[OraclePacket]:
        """src/rtldavis/dsp.py:139-169."""
        c = self.cfg
        B = c.block_size
        if np.iscomplexobj(input_data):
            if input_data.size != B:
                raise ValueError("Incompatible array sizes")
            block = np.asarray(input_data, dtype=np.complex128)
        else:
            if input_data.size != 2 * B:
                raise ValueError("Incompatible array sizes")
            block = byte_to_cmplx(np.asarray(input_data))
        self.iq = np.roll(self.iq, -B)
        self.filtered = np.roll(self.filtered, -B)
        self.discriminated = np.roll(self.discriminated, -B)
        self.quantized = np.roll(self.quantized, -B)
        self.iq[9:] = rotate_fs4(block)
        self.filtered[1:] = fir9(self.iq, B)
        self.discriminated[B:] = discriminate(self.filtered)
        self.quantized[c.buffer_length - B:] = self._quantize(self.discriminated[B:])
        return self._slice(search(self.quantized, c))

    def _slice(self, indices: Sequence[int]) -> List[OraclePacket]:
        """src/rtldavis/dsp.py:190-246."""
        c = self.cfg
        seen = set()
        out: List[OraclePacket] = []
        for q in indices:
            if q > c.block_size:
                continue
            data = slice_bytes(self.quantized, q, c)
            if data in seen:
                continue
            seen.add(data)
            rssi, snr = rssi_snr(self.filtered, q, c)
            out.append(OraclePacket(q, np.frombuffer(data, dtype=np.uint8), rssi, snr))
        return out


# ----------------------------------------------------------------------------
# whole-stream formulation
# ----------------------------------------------------------------------------

def demod_stream_oneshot(raw: np.ndarray):
    """Whole-stream stages.  Returns (f, d, bits) for N = raw.size // 2 samples.

    f[t] = sum_m c_m * ypad[t+m] with ypad = 9 zeros ++ y (so f[t] uses
    y[t-9..t-1]); d[t] = disc(f[t-1], f[t]) with f[-1] = 0; bits = signbit(d).
    Equivalent to running dsp.Demodulator block by block from reset
    (src/rtldavis/dsp.py:154-166).
    """
    if np.iscomplexobj(raw):
        x = np.asarray(raw, dtype=np.complex128)
    else:
        x = byte_to_cmplx(np.asarray(raw))
    n = x.size
    y = rotate_fs4(x)
    ypad = np.concatenate([np.zeros(9, dtype=np.complex128), y])
    f = fir9(ypad, n)
    fp = np.concatenate([np.zeros(1, dtype=np.complex128), f])
    d = discriminate(fp)
    return f, d, quantize(d)


def calls_from_oneshot(f: np.ndarray, bits: np.ndarray, cfg: OracleConfig) -> List[List[OraclePacket]]:
    """Per-call packet lists rebuilt from whole-stream arrays.

    Call b sees quantized = bits[(b+1)B-L : (b+1)B] (zeros before the stream)
    and filtered[j] = f[bB + j - 1] (src/rtldavis/dsp.py:154-166).
    """
    B, L = cfg.block_size, cfg.buffer_length
    n = bits.size
    assert n % B == 0
    padded = np.concatenate([np.zeros(L, dtype=np.uint8), bits])
    fpad = np.concatenate([np.zeros(1, dtype=np.complex128), f])
    calls = []
    for b in range(n // B):
        end = (b + 1) * B
        window = padded[end: end + L]  # == bits[end-L:end] with zero history
        filt = fpad[b * B: b * B + B + 1]
        seen = set()
        pk: List[OraclePacket] = []
        for q in search(window, cfg):
            if q > B:
                continue
            data = slice_bytes(window, q, cfg)
            if data in seen:
                continue
            seen.add(data)
            rssi, snr = rssi_snr(filt, q, cfg)
            pk.append(OraclePacket(q, np.frombuffer(data, dtype=np.uint8), rssi, snr))
        calls.append(pk)
    return calls


def pack_bits_le(bits: np.ndarray) -> np.ndarray:
    """Packed bitstream format of the build: sample t -> byte t//8, bit t%8 (LSB first)."""
    return np.packbits(bits.astype(np.uint8), bitorder="little")


# ----------------------------------------------------------------------------
# protocol.Parser.parse front half
# ----------------------------------------------------------------------------

def swap_bit_order(b: int) -> int:
    """protocol.swap_bit_order (src/rtldavis/protocol.py:79-83)."""
    b = ((b & 0xF0) >> 4) | ((b & 0x0F) << 4)
    b = ((b & 0xCC) >> 2) | ((b & 0x33) << 2)
    b = ((b & 0xAA) >> 1) | ((b & 0x55) << 1)
    return b


def crc16_ccitt(data: bytes) -> int:
    """crc.CRC("CCITT-16", 0, 0x1021, 0).checksum (src/rtldavis/crc.py:19-42)."""
    crc = 0
    for byte in data:
        crc ^= byte << 8
        for _ in range(8):
            crc = ((crc << 1) ^ 0x1021) & 0xFFFF if crc & 0x8000 else (crc << 1) & 0xFFFF
    return crc


def freq_error(discriminated: np.ndarray, index: int, cfg: OracleConfig) -> int:
    """Frequency error of Parser.parse (src/rtldavis/protocol.py:304-311)."""
    mean = np.mean(discriminated[index: index + cfg.preamble_length])
    return -int((mean * float(cfg.sample_rate)) / (2 * math.pi))
