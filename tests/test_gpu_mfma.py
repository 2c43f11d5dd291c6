"""GPU checks of k_demod_mfma in isolation (through the C ABI test hook rd_debug_demod_mfma):
the matrix pipe returns the integer FIR exactly, the signs outside the fix-up list equal the oracle's,
and the list holds what it must (first run of a stream, first group of a chunk, guard-band groups)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import mfma_model as M  # noqa: E402
from oracle import dsp_oracle as O  # noqa: E402
from rtldavis_amd import _lib  # noqa: E402
from rtldavis_amd.synth import synth_stream  # noqa: E402

pytestmark = pytest.mark.gpu


def run_kernel(streams: np.ndarray, hist: np.ndarray | None = None):
    ns, nbytes = streams.shape
    n = nbytes // 2
    if hist is None:
        buf, hb, hm = np.ascontiguousarray(streams), 0, 0
    else:
        hb, hm = hist.shape[1], 1
        buf = np.ascontiguousarray(np.concatenate([hist, streams], axis=1))
    tiles = (n + M.TILE - 1) // M.TILE
    words = (n + 31) // 32
    g = np.zeros((ns, tiles * M.TILE, 2), dtype=np.float32)
    bits = np.zeros((ns, words), dtype=np.uint32)
    cap = ns * words
    fix = np.zeros(cap, dtype=np.uint32)
    nfix = C.c_uint32(0)
    rc = _lib.lib().rd_debug_demod_mfma(buf.ctypes.data, ns, n, hm, hb, g.ctypes.data, bits.ctypes.data,
                                        fix.ctypes.data, cap, C.byref(nfix))
    _lib.check(rc)
    assert nfix.value <= cap
    return g, bits, fix[: nfix.value]


def flagged_groups(fix: np.ndarray, ns: int, words: int) -> np.ndarray:
    out = np.zeros((ns, words * 4), dtype=bool)
    for e in fix:
        w, m = int(e) >> 4, int(e) & 15
        s, wi = divmod(w, words)
        for gidx in range(4):
            if m >> gidx & 1:
                out[s, 4 * wi + gidx] = True
    return out


def check(streams: np.ndarray, hist=None, chunk=6):
    g, bits, fix = run_kernel(streams, hist)
    ns, n = streams.shape[0], streams.shape[1] // 2
    words = (n + 31) // 32
    fl = flagged_groups(fix, ns, words)
    for s in range(ns):
        h = None if hist is None else hist[s]
        # exact values in kernel units, index t = 0..n; without history the first outputs see made-up bytes
        gd = M.g_direct(streams[s], h if h is not None else np.full(18, 127, np.uint8)) * M.UNIT
        want = np.stack([gd.real, gd.imag], axis=1).astype(np.float32)  # one rounding, like the kernel's fma
        got = g[s]
        tl = np.arange(got.shape[0]) % M.TILE
        t = np.arange(got.shape[0])
        ok = (tl >= 1) & (t < n)   # the kernel computes outputs 1..2047 of a tile (2048 is the next tile's 0)
        if hist is None:
            ok &= t >= 9               # earlier outputs see the zero state, which the kernel leaves to the fix-up
        else:
            ok &= (t >= 1)
        assert np.array_equal(got[ok], want[t[ok]]), f"stream {s}: g differs"
        # signs
        if hist is None:
            _, _, ob = O.demod_stream_oneshot(streams[s])
        else:
            both = np.concatenate([h, streams[s]])
            _, _, ob_all = O.demod_stream_oneshot(both)
            ob = ob_all[h.size // 2:]
        mine = np.unpackbits(bits[s].view(np.uint8), bitorder="little")[:n]
        grp = np.arange(n) // 8
        bad = (mine != ob) & ~fl[s][grp]
        assert not bad.any(), f"stream {s}: wrong sign outside the fix-up list at {np.nonzero(bad)[0][:8]}"
        # bits past the end of a ragged stream are zero
        allbits = np.unpackbits(bits[s].view(np.uint8), bitorder="little")
        assert not allbits[n:].any()
        # what must be on the list
        if hist is None:
            assert fl[s][:4].all(), "first run of a stream must be re-evaluated exactly"
        else:
            assert fl[s][0]
    return fix, fl


def test_matrix_pipe_is_exact_and_signs_match_random():
    rng = np.random.default_rng(21)
    streams = rng.integers(0, 256, size=(3, 2 * 8 * M.TILE), dtype=np.uint8)
    streams[1] = rng.choice(np.array([0, 255], dtype=np.uint8), size=streams.shape[1])  # largest magnitudes
    check(streams)


def test_synthetic_bursts_and_few_flags():
    streams = np.stack([synth_stream(s, n_samples=4 * 8192) for s in range(4)])
    fix, fl = check(streams)
    # the guard band itself is tiny now: almost everything on the list is a forced chunk/stream start
    tiles = streams.shape[1] // 2 // M.TILE
    assert fl.sum() <= 4 * (4 + tiles + 8)


def test_ragged_tail_and_history():
    rng = np.random.default_rng(22)
    n = 8192 + 512 + 32  # not a multiple of the tile, nor of 64 samples
    streams = rng.integers(100, 156, size=(2, 2 * n), dtype=np.uint8)
    check(streams)
    hist = rng.integers(100, 156, size=(2, 64), dtype=np.uint8)
    check(streams, hist=hist)


def test_degenerate_input_is_flagged_not_wrong():
    rng = np.random.default_rng(23)
    streams = (127 + rng.integers(0, 2, size=(2, 2 * 2 * M.TILE))).astype(np.uint8)
    streams[1, :] = 127  # constant input: g is the same small DC value everywhere
    check(streams)


def test_dense_matrix_instruction_build_is_bit_exact_too():
    """The product issues ONE 2:4-sparse v_smfmac_f32_32x32x32_f16 per block (rd_mfma.h: the tap rows are 2:4 sparse by
    construction); `make -C rtldavis_amd/csrc dense` builds round 3's pair of dense v_mfma_f32_32x32x16_f16 instead
    (1.5 % slower at the power cap: profiles/r04_sparse_mfma.txt).  Same arithmetic: this file's raw-pipe-output checks
    and the batch parity tests run on that library in a child process (the library is chosen at load time).  Skipped when
    the library has not been built."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "rtldavis_amd", "librtldavis_hip_dense.so")
    if not os.path.exists(lib):
        pytest.skip("librtldavis_hip_dense.so not built (make -C rtldavis_amd/csrc dense)")
    env = dict(os.environ, RTLDAVIS_HIP_LIB=lib)
    sel = ("not dense_matrix_instruction and (matrix_pipe or synthetic_bursts or ragged_tail or degenerate or test_batch_ or "
           "random_configs or soak_small or full_size or startup or one_launch_tail or long_streams)")
    out = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-k", sel, "-p", "no:cacheprovider",
                          os.path.join(root, "tests", "test_gpu_mfma.py"), os.path.join(root, "tests", "test_gpu_parity.py")],
                         env=env, cwd=root, capture_output=True, text=True, timeout=900)
    tail = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-400:]
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-1000:]
    assert " passed" in tail and "failed" not in tail, tail
