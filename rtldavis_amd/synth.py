"""Canonical synthetic Davis IQ streams.

Synthetic data only: the reference tree holds no recorded IQ.  The five
payloads are the CRC-valid on-air packets found in the reference's own tests
(tests/test_protocol.py:29, debug_tools/test_tx.py:37,
decoders/temperature_test.py:15, decoders/humidity_test.py:15,20), in the
on-air bit order ``Demodulator._slice`` emits (src/rtldavis/dsp.py:197-200).
"""
from __future__ import annotations

import numpy as np

BLOCK_SIZE = 8192
BLOCKS_PER_STREAM = 33
STREAM_SAMPLES = BLOCK_SIZE * BLOCKS_PER_STREAM  # 270336 complex samples = 1.0057 s
SAMPLE_RATE = 19200 * 14
SYMBOL_LENGTH = 14

OTA_PACKETS = (
    "cb8907c02b0b80408eff",
    "cb8981a0b1ccd3f08fbb",
    "cb8901a034349fd02679",
    "cb8905604ac11c005a13",
    "cb890520ac8bd4000e5c",
)


def packet_bits(ota_hex: str) -> np.ndarray:
    """MSB-first bits of a 10-byte on-air packet."""
    return np.unpackbits(np.frombuffer(bytes.fromhex(ota_hex), dtype=np.uint8))


def synth_stream(seed: int, n_samples: int = STREAM_SAMPLES,
                 amplitude: float = 0.5, noise: float = 0.05,
                 start: int | None = None, symbol_length: int = SYMBOL_LENGTH,
                 margin: int = BLOCK_SIZE) -> np.ndarray:
    """One stream of uint8 interleaved IQ (length 2*n_samples) holding one burst.

    The burst sits at -Fs/4 (+cfo) in the raw IQ because the reference's
    Fs/4 rotation (dsp.py:42-49) moves that frequency to DC.  Bit 1 is the
    +4800 Hz deviation (negative discriminator output, dsp.py:88-98).
    Draw order from the generator: payload, start, cfo, noise-real, noise-imag.
    ``start`` overrides the drawn burst position (the draw still happens, so
    the noise is unchanged); ``symbol_length`` changes samples per symbol (the
    sample rate follows, 19200 * symbol_length); ``margin`` is the burst-free
    guard at both ends.
    """
    sample_rate = 19200 * symbol_length
    rng = np.random.default_rng(seed)
    payload = OTA_PACKETS[int(rng.integers(0, len(OTA_PACKETS)))]
    sym = np.concatenate([
        np.tile(np.array([1, 0], dtype=np.uint8), 16),
        packet_bits(payload),
        np.zeros(8, dtype=np.uint8),
    ])
    chips = np.repeat(sym, symbol_length)
    drawn = int(rng.integers(margin, n_samples - chips.size - margin))
    start = drawn if start is None else int(start)
    cfo = float(rng.uniform(-2000.0, 2000.0))
    freq = np.full(n_samples, -sample_rate / 4.0 + cfo)
    on = np.zeros(n_samples)
    on[start:start + chips.size] = 1.0
    freq[start:start + chips.size] += np.where(chips == 1, 4800.0, -4800.0)
    phase = np.cumsum(freq) * (2.0 * np.pi / sample_rate)
    x = amplitude * on * np.exp(1j * phase)
    x = x + noise * (rng.standard_normal(n_samples) + 1j * rng.standard_normal(n_samples))
    out = np.empty(2 * n_samples, dtype=np.uint8)
    out[0::2] = np.clip(np.rint(x.real * 127.6 + 127.4), 0, 255).astype(np.uint8)
    out[1::2] = np.clip(np.rint(x.imag * 127.6 + 127.4), 0, 255).astype(np.uint8)
    return out


def synth_two_bursts(seed: int, gap: int, n_samples: int = 4 * BLOCK_SIZE, amplitude: float = 0.5,
                      noise: float = 0.05) -> np.ndarray:
    """Two identical bursts `gap` samples apart (start to start) in one stream: both land in the
    same demodulate() window, so the per-call dedupe (dsp.py:203-205) has real work to do."""
    rng = np.random.default_rng(seed)
    payload = OTA_PACKETS[int(rng.integers(0, len(OTA_PACKETS)))]
    sym = np.concatenate([np.tile(np.array([1, 0], dtype=np.uint8), 16), packet_bits(payload),
                          np.zeros(8, dtype=np.uint8)])
    chips = np.repeat(sym, SYMBOL_LENGTH)
    start = BLOCK_SIZE + int(rng.integers(0, 2000))
    cfo = float(rng.uniform(-2000.0, 2000.0))
    freq = np.full(n_samples, -SAMPLE_RATE / 4.0 + cfo)
    on = np.zeros(n_samples)
    for s0 in (start, start + gap):
        on[s0:s0 + chips.size] = 1.0
        freq[s0:s0 + chips.size] += np.where(chips == 1, 4800.0, -4800.0)
    phase = np.cumsum(freq) * (2.0 * np.pi / SAMPLE_RATE)
    x = amplitude * on * np.exp(1j * phase)
    x = x + noise * (rng.standard_normal(n_samples) + 1j * rng.standard_normal(n_samples))
    out = np.empty(2 * n_samples, dtype=np.uint8)
    out[0::2] = np.clip(np.rint(x.real * 127.6 + 127.4), 0, 255).astype(np.uint8)
    out[1::2] = np.clip(np.rint(x.imag * 127.6 + 127.4), 0, 255).astype(np.uint8)
    return out


def synth_streams(seeds, n_samples: int = STREAM_SAMPLES) -> np.ndarray:
    """Stack of streams, shape [len(seeds), 2*n_samples] uint8."""
    seeds = list(seeds)
    out = np.empty((len(seeds), 2 * n_samples), dtype=np.uint8)
    for i, s in enumerate(seeds):
        out[i] = synth_stream(s, n_samples)
    return out


def synth_wideband(seeds, shifts_hz, n_out: int, decim: int = 100, out_rate: int = SYMBOL_LENGTH * 19200,
                   amplitude: float = 0.12, noise: float = 0.02, noise_seed: int = 1234):
    """One wideband capture (uint8 I,Q interleaved, decim*out_rate samples/s, n_out*decim samples)
    holding one burst per entry: burst i is the packet synth_stream(seeds[i]) carries, at
    shifts_hz[i] Hz from the capture's centre (+ a random cfo of +-2 kHz), starting somewhere
    between the first and the last 8192 output samples.  Returns (raw, [(payload_hex, start_out)])."""
    fw = decim * out_rate
    n = n_out * decim
    rng = np.random.default_rng(noise_seed)
    x = noise * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    info = []
    for seed, shift in zip(seeds, shifts_hz):
        r = np.random.default_rng(seed)
        payload = OTA_PACKETS[int(r.integers(0, len(OTA_PACKETS)))]
        sym = np.concatenate([np.tile(np.array([1, 0], dtype=np.uint8), 16), packet_bits(payload),
                              np.zeros(8, dtype=np.uint8)])
        chips = np.repeat(sym, SYMBOL_LENGTH * decim)
        start_out = int(r.integers(BLOCK_SIZE, n_out - chips.size // decim - BLOCK_SIZE))
        cfo = float(r.uniform(-2000.0, 2000.0))
        lo, hi = start_out * decim, start_out * decim + chips.size
        freq = float(shift) + cfo + np.where(chips == 1, 4800.0, -4800.0)
        # phase of the tone from sample 0 (continuous through the burst)
        phase = (float(shift) + cfo) * lo * (2.0 * np.pi / fw) + np.cumsum(freq) * (2.0 * np.pi / fw)
        x[lo:hi] += amplitude * np.exp(1j * phase)
        info.append((payload, start_out))
    out = np.empty(2 * n, dtype=np.uint8)
    out[0::2] = np.clip(np.rint(x.real * 127.6 + 127.4), 0, 255).astype(np.uint8)
    out[1::2] = np.clip(np.rint(x.imag * 127.6 + 127.4), 0, 255).astype(np.uint8)
    return out, info


def payload_of(seed: int) -> str:
    """The on-air packet hex synth_stream(seed) carries."""
    rng = np.random.default_rng(seed)
    return OTA_PACKETS[int(rng.integers(0, len(OTA_PACKETS)))]
