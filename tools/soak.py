"""Randomised soak of the batch and streaming paths against the C oracle (test infrastructure):
small random PacketConfigs and repetitive inputs that produce dense matches, long runs of
identical packets at adjacent positions and many block-boundary positions.
usage: soak.py [n_cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import c_oracle as CO
from rtldavis_amd import batch, dsp


def make_input(rng, ns, n):
    kind = rng.integers(0, 6)
    if kind == 0:   # noise around mid-scale
        return rng.integers(96, 160, size=(ns, 2 * n), dtype=np.uint8)
    if kind == 1:   # two-level
        return rng.choice(np.array([100, 156], np.uint8), size=(ns, 2 * n))
    if kind == 2:   # periodic with a short period (repeating bits -> identical packets everywhere)
        per = int(rng.integers(2, 40))
        base = rng.integers(64, 192, size=(ns, 2 * per), dtype=np.uint8)
        return np.tile(base, (1, n // per + 1))[:, : 2 * n].copy()
    if kind == 3:   # a tone plus a little noise
        t = np.arange(n)
        f = rng.uniform(-0.3, 0.3, size=(ns, 1))
        x = 60 * np.exp(2j * np.pi * (f * t + rng.uniform(0, 1, (ns, 1)))) + rng.normal(0, 2, (ns, n)) + 1j * rng.normal(0, 2, (ns, n))
        out = np.empty((ns, 2 * n), np.uint8)
        out[:, 0::2] = np.clip(np.rint(x.real + 127.4), 0, 255)
        out[:, 1::2] = np.clip(np.rint(x.imag + 127.4), 0, 255)
        return out
    if kind == 4:   # constant
        return np.full((ns, 2 * n), int(rng.integers(0, 256)), np.uint8)
    return rng.integers(0, 256, size=(ns, 2 * n), dtype=np.uint8)


def soak(n_cases, seed, verbose=True):
    rng = np.random.default_rng(seed)
    t0 = time.time()
    nrec = 0
    for case in range(n_cases):
        S = int(rng.integers(1, 9))
        P = int(rng.integers(1, 7))
        K = int(rng.integers(P, 41))
        B = int(rng.choice([32, 36, 40, 64, 96, 128, 256]))
        nb = int(rng.integers(1, 12))
        ns = int(rng.integers(1, 6))
        pre = "".join(str(int(b)) for b in rng.integers(0, 2, size=P))
        raw = make_input(rng, ns, B * nb)
        cfg = dsp.PacketConfig(19200, S, P, K, pre, B)
        ocfg = CO.make_cfg(19200, S, P, K, pre, B)
        want, wbits = CO.demod_batch(raw, ocfg, threads=2, want_bits=True, cap_per_stream=400000)
        bd = batch.BatchDemodulator(cfg, ns, nb)
        res = bd.demodulate(raw)
        tag = (case, S, P, K, B, nb, ns, pre)
        for i in range(ns):
            assert np.array_equal(bd.bits(i), wbits[i]), (tag, i, "bits")
            got = [(c, p.index, bytes(p.data).hex()) for c, ps in enumerate(res[i]) for p in ps]
            exp = [(p.call, p.index, bytes(p.data).hex()) for p in want[i]]
            assert got == exp, (tag, i, len(got), len(exp), got[:5], exp[:5])
            flat = [p for ps in res[i] for p in ps]
            for a, b in zip(flat, want[i]):
                ok = abs(a.rssi - b.rssi) < 1e-3 and (abs(a.snr - b.snr) < 1e-3 or (a.snr != a.snr and b.snr != b.snr)
                                                      or (np.isinf(a.snr) and np.isinf(b.snr)))
                assert ok, (tag, i, a, b)
            nrec += len(got)
        # streaming handle on stream 0 (skip when a call exceeds its 64-packet limit)
        # A call in which some packet has signal_power == 0 (a preamble of zeros matching inside the zero start-up
        # state) ends in the reference's math.log10(0): ValueError out of demodulate(), the call's packets are
        # lost, the state has advanced (dsp.py:231-236).  The oracle reports that packet with snr = -inf.
        dem = dsp.Demodulator(cfg)
        raising = {p.call for p in want[0] if np.isinf(p.snr) and p.snr < 0}
        for b in range(nb):
            exp = [(p.index, bytes(p.data).hex()) for p in want[0] if p.call == b]
            try:
                got = [(p.index, bytes(p.data).hex()) for p in dem.demodulate(raw[0][2 * B * b: 2 * B * (b + 1)])]
                assert b not in raising, (tag, "streaming: the reference raises here", b)
                assert got == exp, (tag, "streaming", b)
            except ValueError as e:
                assert "math domain error" in str(e) and b in raising, (tag, "streaming", b, str(e))
        if verbose and case % 20 == 0:
            print(f"case {case}: ok ({nrec} packets so far, {time.time() - t0:.0f} s)", flush=True)
    if verbose:
        print(f"soak ok: {n_cases} cases, {nrec} packets, {time.time() - t0:.0f} s")
    return nrec


def soak_bursts(n_streams, seed, verbose=True):
    """Production config: one burst per stream with its preamble placed on and around the block
    boundaries (q = B in one call and q = 0 in the next, py:194), several amplitudes and noise
    levels, 4-block streams; batch path and (for a few streams) the streaming handle."""
    from rtldavis_amd import synth
    rng = np.random.default_rng(seed)
    B, nb = 8192, 4
    cfg = dsp.PacketConfig(19200, 14, 16, 80, "1100101110001001", B)
    ocfg = CO.make_cfg()
    raws = []
    for i in range(n_streams):
        # the packet's preamble starts 32 symbols after the burst start
        edge = int(rng.integers(1, nb)) * B
        start = edge - 32 * 14 + int(rng.integers(-20, 21)) - int(rng.integers(0, 2)) * B // 2
        raws.append(synth.synth_stream(int(rng.integers(0, 1 << 30)), n_samples=B * nb, amplitude=float(rng.choice([0.1, 0.3, 0.5, 0.9])),
                                       noise=float(rng.choice([0.01, 0.05, 0.1])), start=max(64, start), margin=64))
    raw = np.stack(raws)
    want, wbits = CO.demod_batch(raw, ocfg, threads=8, want_bits=True)
    bd = batch.BatchDemodulator(cfg, n_streams, nb)
    res = bd.demodulate(raw)
    n = 0
    twice = 0
    for i in range(n_streams):
        assert np.array_equal(bd.bits(i), wbits[i]), (i, "bits")
        got = [(c, p.index, bytes(p.data).hex()) for c, ps in enumerate(res[i]) for p in ps]
        exp = [(p.call, p.index, bytes(p.data).hex()) for p in want[i]]
        assert got == exp, (i, got, exp)
        # RSSI / SNR of every packet against the oracle's float64 windows (1e-3 dB): the batch path evaluates them on
        # the matrix pipe, windows that reach the end of the stream on its fp32 path
        flat = [p for ps in res[i] for p in ps]
        for a, b in zip(flat, want[i]):
            assert abs(a.rssi - b.rssi) < 1e-3 and abs(a.snr - b.snr) < 1e-3, (i, a, b)
        twice += sum(1 for g in got if g[1] == B)
        n += len(got)
    for i in range(min(n_streams, 24)):
        dem = dsp.Demodulator(cfg)
        calls = [dem.demodulate(raw[i][2 * B * b: 2 * B * (b + 1)]) for b in range(nb)]
        got = [(c, p.index, bytes(p.data).hex()) for c, ps in enumerate(calls) for p in ps]
        assert got == [(p.call, p.index, bytes(p.data).hex()) for p in want[i]], (i, "streaming")
    if verbose:
        print(f"burst soak ok: {n_streams} streams, {n} packets, {twice} reported at q = B")
    return n, twice


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "bursts":
        soak_bursts(int(sys.argv[2]) if len(sys.argv) > 2 else 512, int(sys.argv[3]) if len(sys.argv) > 3 else 3)
        sys.exit(0)
    soak(int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 7)
