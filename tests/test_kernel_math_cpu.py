"""The arithmetic the HIP kernels execute (rtldavis_amd/csrc/rd_math.h), compiled for the
host by tests/host_harness.cpp, against the oracle.  Covers what a GPU test cannot show
cheaply: that every fp32 mismatch lies inside the guard band, on many input classes."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import dsp_oracle as O
from rtldavis_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tests", "_build", "libhostharness.so")


@pytest.fixture(scope="module")
def hh():
    src = os.path.join(ROOT, "tests", "host_harness.cpp")
    hdr = os.path.join(ROOT, "rtldavis_amd", "csrc", "rd_math.h")
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        os.makedirs(os.path.dirname(SO), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-march=x86-64-v3", "-shared", "-fPIC", "-o", SO, src])
    L = C.CDLL(SO)
    L.hh_fast_stream.restype = C.c_long
    L.hh_fast_stream.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_void_p]
    L.hh_exact_stream.argtypes = [C.c_void_p, C.c_long, C.c_void_p]
    L.hh_exact_group.restype = C.c_uint8
    L.hh_exact_group.argtypes = [C.c_void_p, C.c_long, C.c_long]
    L.hh_exact_group_dw.restype = C.c_uint8
    L.hh_exact_group_dw.argtypes = [C.c_void_p, C.c_long, C.c_long]
    L.hh_f64_stream.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_void_p]
    L.hh_threshold.restype = C.c_float
    L.hh_threshold.argtypes = [C.c_float]
    return L


def run(hh, raw):
    raw = np.ascontiguousarray(raw, dtype=np.uint8)
    n = raw.size // 2
    words = np.zeros(n // 32, np.uint32)
    flg = np.zeros(n // 32, np.uint8)
    nf = hh.hh_fast_stream(raw.ctypes.data, n, words.ctypes.data, flg.ctypes.data)
    ex = np.zeros((n + 31) // 32, np.uint32)
    hh.hh_exact_stream(raw.ctypes.data, n, ex.ctypes.data)
    return words, flg, nf, ex


def inputs():
    rng = np.random.default_rng(99)
    n = 2 * 65536
    yield "synthetic0", synth.synth_stream(0)[: 2 * 8192 * 8]
    yield "synthetic_burst", synth.synth_stream(0)[2 * 8192 * 20: 2 * 8192 * 23]
    yield "uniform", rng.integers(0, 256, size=n, dtype=np.uint8)
    yield "lsb_noise", rng.integers(127, 129, size=n, dtype=np.uint8)
    yield "pm3", rng.integers(124, 132, size=n, dtype=np.uint8)
    yield "saturated", rng.choice(np.array([0, 255], np.uint8), size=n)
    yield "constant", np.full(n, 127, np.uint8)
    yield "ramp", (np.arange(n) % 256).astype(np.uint8)
    yield "strong_tone", np.clip(np.rint(127.4 + 120 * np.cos(np.arange(n) * 0.7)), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("name,raw", list(inputs()), ids=[n for n, _ in inputs()])
def test_exact_path_equals_oracle_and_guard_covers_fast_path(hh, name, raw):
    f, d, bits = O.demod_stream_oneshot(raw)
    ref = O.pack_bits_le(bits).view(np.uint32)
    words, flg, nf, ex = run(hh, raw)
    assert np.array_equal(ex, ref), "exact integer path differs from the float64 oracle"
    # per 8-sample group (one output byte): every fp32 mismatch must be flagged
    gb, rb = words.view(np.uint8), ref.view(np.uint8)
    gflag = ((flg[:, None] >> np.arange(4)[None, :]) & 1).astype(bool).reshape(-1)
    mism = gb != rb
    assert not np.any(mism & ~gflag), "an fp32 sign error escaped the guard band"
    assert flg[0] == 0xF  # zero-history run is always re-evaluated
    # what the fix-up kernel stores for flagged groups equals the reference byte
    n = raw.size // 2
    for gi in np.flatnonzero(gflag)[:200]:
        assert hh.hh_exact_group(np.ascontiguousarray(raw).ctypes.data, n, int(gi) * 8) == rb[gi]
    if name in ("synthetic0", "uniform"):
        assert nf / gflag.size < 0.02


def test_fixup_dword_window_form_equals_run_form(hh):
    rng = np.random.default_rng(17)
    raw = rng.integers(0, 256, size=2 * 4096, dtype=np.uint8)
    f, d, bits = O.demod_stream_oneshot(raw)
    ref = O.pack_bits_le(bits)
    for g in list(range(0, 8)) + list(rng.integers(0, 512, size=100)) + [511]:
        assert hh.hh_exact_group_dw(raw.ctypes.data, 4096, int(g) * 8) == ref[int(g)], g
    # ragged end: 1000 samples, last group holds 8 valid samples, a 996-sample stream holds 4
    for n in (1000, 996):
        f, d, bits = O.demod_stream_oneshot(raw[: 2 * n])
        want = np.zeros(1024, np.uint8); want[:n] = bits
        ref = O.pack_bits_le(want)
        g = (n - 1) // 8
        assert hh.hh_exact_group_dw(raw.ctypes.data, n, g * 8) == ref[g]


def test_ragged_tail_exact(hh):
    raw = np.random.default_rng(3).integers(0, 256, size=2 * 1000, dtype=np.uint8)
    f, d, bits = O.demod_stream_oneshot(raw)
    ex = np.zeros(32, np.uint32)
    hh.hh_exact_stream(raw.ctypes.data, 1000, ex.ctypes.data)
    want = np.zeros(32 * 32, np.uint8)
    want[:1000] = bits
    assert np.array_equal(ex, O.pack_bits_le(want).view(np.uint32))


def test_float64_values_match_oracle(hh):
    raw = synth.synth_stream(1)[: 2 * 8192 * 2]
    n = raw.size // 2
    filt = np.zeros(2 * (n + 1))
    disc = np.zeros(n)
    hh.hh_f64_stream(raw.ctypes.data, n, filt.ctypes.data, disc.ctypes.data)
    f, d, bits = O.demod_stream_oneshot(raw)
    np.testing.assert_allclose(filt.view(np.complex128)[1:], f, rtol=0, atol=1e-14)
    assert np.all(np.abs(disc - d) <= 1e-9 * np.maximum(1, np.abs(d)))
    assert np.array_equal(np.signbit(disc), np.signbit(d))


def test_threshold_is_monotone_and_positive(hh):
    prev = 0.0
    for F in [0.0, 1e-3, 0.1, 1.0, 10.0, 100.0, 258.0]:
        t = hh.hh_threshold(F)
        assert t > 0 and t >= prev
        prev = t
    # worst case stays far below typical |num| of a full-scale signal (F^2 * sin)
    assert hh.hh_threshold(258.0) < 0.1
