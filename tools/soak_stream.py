"""Randomised soak of the streaming handle's one-launch forms against the C oracle (test infrastructure): the Davis
configuration at random block sizes (one to four workgroups per complex block, ragged pieces), inputs from noise through
tones and constants to synthetic Davis streams, blocks fed as uint8, as complex128 (the bytes through the reference's
LUT, py:26) or uint8 first and complex128 behind; synchronously or with two blocks in flight; pushed into device memory
by the host or read from the pinned slot.  Every call's packets (index, bytes, order; RSSI / SNR to 1e-3 dB) and the
final quantized window against the oracle's.
usage: soak_stream.py [n_cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import c_oracle as CO
from rtldavis_amd import dsp, synth
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from soak import make_input

PRE = "1100101110001001"
LUT = (np.arange(256, dtype=np.float64) - 127.4) / 127.6


def bursty(rng, n):
    """n samples of noise with one to six Davis bursts (on-air packets of synth.OTA_PACKETS, 1010... lead-in) at random
    positions, amplitudes and frequency offsets - some a few samples apart from a block boundary, some overlapping."""
    fs = synth.SAMPLE_RATE
    freq = np.full(n, -fs / 4.0)
    amp = np.zeros(n)
    for _ in range(int(rng.integers(1, 7))):
        payload = synth.OTA_PACKETS[int(rng.integers(0, len(synth.OTA_PACKETS)))]
        sym = np.concatenate([np.tile(np.array([1, 0], np.uint8), 16), synth.packet_bits(payload), np.zeros(8, np.uint8)])
        chips = np.repeat(sym, synth.SYMBOL_LENGTH)
        if chips.size + 64 >= n:
            break
        start = int(rng.integers(32, n - chips.size - 32))
        freq[start:start + chips.size] = -fs / 4.0 + float(rng.uniform(-2000.0, 2000.0)) + np.where(chips == 1, 4800.0, -4800.0)
        amp[start:start + chips.size] = float(rng.uniform(0.05, 0.9))
    x = amp * np.exp(1j * np.cumsum(freq) * (2.0 * np.pi / fs))
    noise = float(rng.choice([0.0, 0.01, 0.05, 0.2]))
    x = x + noise * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    out = np.empty((1, 2 * n), np.uint8)
    out[0, 0::2] = np.clip(np.rint(x.real * 127.6 + 127.4), 0, 255)
    out[0, 1::2] = np.clip(np.rint(x.imag * 127.6 + 127.4), 0, 255)
    return out


def soak(n_cases, seed, verbose=True):
    rng = np.random.default_rng(seed)
    t0 = time.time()
    nrec = nblocks = 0
    kinds = {}
    for case in range(n_cases):
        B = int(rng.choice([2048, 2080, 3072, 4096, 5000 // 32 * 32, 6176, 8192]))
        nb = int(rng.integers(3, 12))
        raw = bursty(rng, B * nb) if rng.integers(0, 3) else make_input(rng, 1, B * nb)
        mode = ("uint8", "complex", "mixed")[int(rng.integers(0, 3))]
        piped = bool(rng.integers(0, 2))
        push = bool(rng.integers(0, 2))
        first_c = 0 if mode == "complex" else nb if mode == "uint8" else int(rng.integers(1, nb))
        want, wbits = CO.demod_batch(raw, CO.make_cfg(19200, 14, 16, 80, PRE, B), threads=2, want_bits=True, cap_per_stream=400000)
        exp = [(p.call, p.index, bytes(p.data).hex(), p.rssi, p.snr) for p in want[0]]
        cplx = LUT[raw[0, 0::2]] + 1j * LUT[raw[0, 1::2]]
        dsp.set_input_push(push)
        dem = dsp.Demodulator(dsp.PacketConfig(19200, 14, 16, 80, PRE, B))
        blocks = [cplx[B * b: B * (b + 1)] if b >= first_c else raw[0, 2 * B * b: 2 * B * (b + 1)] for b in range(nb)]
        calls = []
        if piped:
            dem.submit(blocks[0])
            for x in blocks[1:]:
                dem.submit(x)
                calls.append(dem.fetch())
            calls.append(dem.fetch())
        else:
            calls = [dem.demodulate(x) for x in blocks]
        got = [(b, p.index, bytes(p.data).hex(), p.rssi, p.snr) for b, c in enumerate(calls) for p in c]
        tag = (case, B, nb, mode, first_c, piped, push)
        assert [g[:3] for g in got] == [e[:3] for e in exp], (tag, len(got), len(exp), got[:4], exp[:4])
        for g, e in zip(got, exp):
            for a, w in ((g[3], e[3]), (g[4], e[4])):
                assert (a != a and w != w) or abs(a - w) < 1e-3, (tag, g, e)
        allbits = np.unpackbits(wbits[0], bitorder="little")[: B * nb]
        q = np.asarray(dem.quantized).astype(np.uint8)
        tail = allbits[-2 * B:] if nb >= 2 else np.concatenate([np.zeros(B, np.uint8), allbits])
        assert np.array_equal(q, tail), (tag, "quantized window")
        nrec += len(exp); nblocks += nb
        kinds[(mode, piped, push)] = kinds.get((mode, piped, push), 0) + 1
    dsp.set_input_push(None)
    if not verbose:
        return nrec
    print(f"soak_stream: {n_cases} cases, {nblocks} blocks, {nrec} packets equal to the C oracle's in {time.time() - t0:.1f} s (seed {seed})")
    print("  cases by (input, two in flight, host push): " + ", ".join(f"{k[0]}/{int(k[1])}/{int(k[2])}: {v}" for k, v in sorted(kinds.items())))
    return nrec


if __name__ == "__main__":
    soak(int(sys.argv[1]) if len(sys.argv) > 1 else 300, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
