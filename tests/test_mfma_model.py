"""CPU checks of the matrix-pipe FIR formulation (rd_mfma.h): the exported tap matrix pushed through
the documented MFMA lane maps reproduces g[t] = sum_m T_m j^m U[t-9+m] exactly, the arithmetic stays
inside f32's exact-integer range, the error constant is what exact rational arithmetic gives, and the
guard band built on it covers every sign the oracle disagrees with."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import mfma_model as M  # noqa: E402
from oracle import dsp_oracle as O  # noqa: E402
from rtldavis_amd.synth import synth_stream  # noqa: E402


def test_tap_scale_and_error_constant():
    # T_m = round(S c_m); the summed quantisation error is what rd_mfma.h states
    for t, c in zip(M.T, M.C12):
        assert t == round(M.SCALE * c / 1e12)
    eps = M.tap_error_sum()
    assert abs(float(eps) - 0.0980034) < 1e-6
    # the -127.4 offset: D = 127.4 (2 T0 - 2 T2 + T4) misses 2048 DHI by 3.6
    d = 127.4 * (2 * M.T[0] - 2 * M.T[2] + M.T[4])
    assert abs(abs(d - 2048 * M.DHI) - 3.6) < 1e-6
    e0 = (float(eps) * 255 + 3.6) / 2 ** 24
    assert e0 < M.E0 < 1.02 * e0  # RD_MF_E0 bounds it with < 2 % slack
    # in byte units of f: 20 times below the fp32 VALU path's 3.3e-5
    assert M.E0 / (M.SCALE / 2 ** 24) < 1.7e-6


def test_digits_are_f16_exact_and_sums_stay_below_2_24():
    taps = M.taps_from_lib()
    assert np.all(taps == np.rint(taps)) and np.abs(taps).max() <= 2048
    # per output row: sum |digit| * max|U| bounds every partial sum of the f32 accumulation
    for dig in range(2):
        rows = np.zeros(32)
        for d in range(3):
            for lane in range(64):
                rows[lane & 31] += np.abs(taps[dig, d, lane]).sum()
        assert rows.max() * 255 + M.DHI < 2 ** 24, (dig, rows.max() * 255)
    # hi * 2048 + lo reproduces the signed integer taps
    full = 2048 * taps[0] + taps[1]
    assert set(np.unique(np.abs(full)).astype(np.int64)) == {0, *M.T}


def test_lane_maps_reproduce_the_direct_filter():
    rng = np.random.default_rng(5)
    taps = M.taps_from_lib()
    for trial in range(3):
        raw = rng.integers(0, 256, size=16 + 2 * M.TILE, dtype=np.uint8)
        if trial == 1:
            raw[:] = rng.choice(np.array([0, 255], dtype=np.uint8), size=raw.size)  # extreme magnitudes
        got = M.model_tile(raw, taps)
        # history for the direct form: the 16-byte halo (8 samples) behind one filler sample; g[1], the
        # first output the kernel computes, starts exactly at the halo (samples -8 .. 0)
        want = M.g_direct(raw[16:], np.concatenate([np.full(2, 127, np.uint8), raw[:16]]))
        assert np.array_equal(got, want[1: M.TILE + 1]), trial


def test_rotation_free_numerator_matches_the_oracle():
    raw = synth_stream(3, n_samples=4 * 8192)
    _, _, bits = O.demod_stream_oneshot(raw)
    g = M.g_true(raw)  # g[t], t = 0..n with zero history
    gm1 = np.concatenate([[0], g[:-1]])  # g[t-1] for t = 0..n
    p = (gm1 * np.conj(g)).real[: bits.size]
    mine = (p > 0).astype(np.uint8)
    # exact integers: the only disagreements possible are exact zeros (zero history start)
    bad = np.nonzero(mine != bits)[0]
    assert bad.size == 0 or bad.max() < 10, bad[:10]


def test_guard_band_covers_every_mismatch():
    """Signs decided from float32(G) with the kernel's threshold: every sample whose fast sign differs
    from the oracle's lies in a flagged group - on noise, on a weak signal and on +-1 LSB data."""
    rng = np.random.default_rng(11)
    cases = [synth_stream(7, n_samples=4 * 8192), synth_stream(8, n_samples=4 * 8192, amplitude=0.02, noise=0.01)]
    tiny = (127 + rng.integers(0, 2, size=2 * 4 * 8192)).astype(np.uint8)
    cases.append(tiny)
    total_flagged = 0
    for raw in cases:
        _, _, bits = O.demod_stream_oneshot(raw)
        g = M.g_true(raw) * M.UNIT  # kernel units (the 3.6 of the offset is part of E0)
        gf_re = g.real.astype(np.float32)
        gf_im = g.imag.astype(np.float32)
        n = bits.size
        a, b = gf_re[:n], gf_im[:n]          # g[t-1] for t = 1..n  (index t-1)
        c, d = gf_re[1: n + 1], gf_im[1: n + 1]
        t1 = (b * d).astype(np.float32)
        # the kernel's fma: -a*c - t1 with one rounding (float64 holds the exact product and sum)
        num32 = (-(a.astype(np.float64) * c.astype(np.float64)) - t1.astype(np.float64)).astype(np.float32)
        fast = np.signbit(num32).astype(np.uint8)
        r = (np.abs(num32).astype(np.float64) - 4.76837158e-7 * np.abs(t1).astype(np.float64)).astype(np.float32)
        # groups of 8 samples: min r over the group against c0(F), F = largest component the group used
        t = np.arange(1, n)
        F = np.maximum(np.maximum(np.abs(a), np.abs(b)), np.maximum(np.abs(c), np.abs(d)))[: n - 1]
        grp = (t // 8)
        Fg = np.zeros(grp.max() + 1, dtype=np.float32)
        np.maximum.at(Fg, grp, F)
        flagged_sample = ~(r[: n - 1] > M.c0(Fg)[grp])
        flagged_group = np.zeros(grp.max() + 1, dtype=bool)
        np.logical_or.at(flagged_group, grp, flagged_sample)
        mism = fast[: n - 1] != bits[1:n]
        assert not np.any(mism & ~flagged_group[grp]), "a wrong sign outside the guard band"
        total_flagged += int(flagged_group.sum())
    assert total_flagged > 0  # the +-1 LSB case does hit exact zeros


def test_b8_rows_are_f16_exact_and_every_row_sum_stays_below_2_24():
    """The 8-output formulation (RD_OPT_B8): a row of the 32-row tile is ONE digit of one output component, so the
    exactness argument is the 16-output formulation's, row by row."""
    taps8 = M.taps8_from_lib()
    assert np.all(taps8 == np.rint(taps8)) and np.abs(taps8).max() <= 2048
    rows = np.zeros(32)
    for d in range(2):
        for lane in range(64):
            rows[lane & 31] += np.abs(taps8[d, lane]).sum()
    assert rows.max() * 255 + M.DHI < 2 ** 24, rows.max() * 255
    # register rho of a lane half: rho < 8 hi digit, rho >= 8 lo digit of the same (output, component)
    for half in range(2):
        for rho in range(8):
            r_hi = (rho & 3) + 8 * (rho >> 2) + 4 * half
            r_lo = ((rho + 8) & 3) + 8 * ((rho + 8) >> 2) + 4 * half
            row_hi = np.concatenate([taps8[d, r_hi + 32 * hk] for d in range(2) for hk in range(2)])
            row_lo = np.concatenate([taps8[d, r_lo + 32 * hk] for d in range(2) for hk in range(2)])
            full = 2048 * row_hi + row_lo
            assert set(np.unique(np.abs(full)).astype(np.int64)) <= {0, *M.T}
            assert np.count_nonzero(full) == 9  # the nine taps of one output component


def test_b8_lane_maps_reproduce_the_direct_filter():
    rng = np.random.default_rng(6)
    taps8 = M.taps8_from_lib()
    for trial in range(3):
        raw = rng.integers(0, 256, size=16 + 2 * M.TILE, dtype=np.uint8)
        if trial == 1:
            raw[:] = rng.choice(np.array([0, 255], dtype=np.uint8), size=raw.size)
        got = M.model_tile8(raw, taps8)
        want = M.g_direct(raw[16:], np.concatenate([np.full(2, 127, np.uint8), raw[:16]]))
        assert np.array_equal(got, want[1: M.TILE + 1]), trial


def test_sparse_tap_matrix_is_the_dense_one():
    """The kernel issues ONE 2:4-sparse matrix instruction per block where the dense formulation has two: the compressed
    taps + positions the library hands it (rd_debug_mfma_taps8s), expanded by the instruction's operand layout, are the
    dense 32 x 32 tap matrix of rd_debug_mfma_taps8 element for element - so everything test_b8_* says about that
    matrix (digits exact in f16, row sums below 2^24, the nine taps of every output) holds for what the kernel runs."""
    taps8 = M.taps8_from_lib()
    dense = np.zeros((32, 32))
    for d in range(2):
        for lane in range(64):
            dense[lane & 31, 16 * d + 8 * (lane >> 5): 16 * d + 8 * (lane >> 5) + 8] = taps8[d, lane]
    sparse = M.taps8_sparse_from_lib()
    assert np.array_equal(sparse, dense)
    # 2:4: no group of four consecutive K of any row holds more than two taps
    assert (np.count_nonzero(dense.reshape(32, 8, 4), axis=2) <= 2).all()

