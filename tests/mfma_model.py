"""NumPy model of k_demod_mfma's data flow (rtldavis_amd/csrc/rd_demod_mfma.hip, rd_mfma.h).
TEST INFRASTRUCTURE ONLY.

It follows the kernel lane by lane: which window bytes a lane holds, the element order of the B
fragment, the documented lane maps of v_mfma_f32_32x32x16_f16 (A[row = lane & 31][k = 8 (lane >> 5) + j],
B[k][col = lane & 31], D[row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)][col = lane & 31]) and the tap
matrix the library exports (rd_debug_mfma_taps).  Everything is an integer, so the model is exact;
tests compare it with a direct evaluation of g[t] = sum_m T_m j^m U[t-9+m] and with the oracle.
The raw byte k enters the matrix pipe as the f16 subnormal k * 2^-24; the -127.4 offset is the constant DHI.
"""
from __future__ import annotations

from fractions import Fraction

import numpy as np

TILE = 2048
SCALE = 18255980.028508045
T = (322807, 879415, 2234983, 3603886, 4173798)
DHI = 21738  # 127.4 * (2 T0 - 2 T2 + T4) = 2048 * DHI - 3.6
C12 = (17682261285, 48171339939, 122424706672, 197408519126, 228626345955)  # fir9 taps * 1e12 (dsp.py:56-69)
ELEM = (0, 2, 1, 3, 4, 6, 5, 7)  # element j of a B fragment = byte ELEM[j] of the lane's 8 window bytes
E0 = 1.73e-6  # RD_MF_E0 (kernel units: 2^-24 S per byte unit)
UNIT = 2.0 ** -24  # a raw byte read as an f16 bit pattern


def taps_from_lib() -> np.ndarray:
    """[digit][k-step][lane][element] as float64 (decoded f16)."""
    from rtldavis_amd import _lib
    raw = np.zeros(2 * 3 * 64 * 8, dtype=np.uint16)
    _lib.lib().rd_debug_mfma_taps(raw.ctypes.data)
    return raw.view(np.float16).astype(np.float64).reshape(2, 3, 64, 8)


def tap_error_sum() -> Fraction:
    """sum over the nine taps of |T_m - S c_m| in exact rational arithmetic."""
    s = Fraction(SCALE)
    eps = [abs(Fraction(t) - s * Fraction(c, 10 ** 12)) for t, c in zip(T, C12)]
    return 2 * sum(eps[:4]) + eps[4]


def _fir(zp: np.ndarray, n: int) -> np.ndarray:
    taps = [T[m if m <= 4 else 8 - m] * (1j ** m) for m in range(9)]
    g = np.zeros(n + 1, dtype=np.complex128)
    for m in range(9):
        g += taps[m] * zp[m: m + n + 1]
    return g


def g_true(raw: np.ndarray, hist: np.ndarray | None = None) -> np.ndarray:
    """sum_m T_m j^m (k - 127.4)[t-9+m] for t = 0..n: the quantised-tap filter on the CENTRED samples
    (x = (k - 127.4) / 127.6, dsp.py:26).  Samples before the stream are `hist` bytes (interleaved, oldest
    first) or - with no history - the zero state y = 0 of dsp.py:131."""
    z = (raw[0::2].astype(np.float64) - 127.4) + 1j * (raw[1::2].astype(np.float64) - 127.4)
    if hist is None:
        pre = np.zeros(9, dtype=np.complex128)
    else:
        pre = ((hist[0::2].astype(np.float64) - 127.4) + 1j * (hist[1::2].astype(np.float64) - 127.4))[-9:]
    return _fir(np.concatenate([pre, z]), z.size)


def g_direct(raw: np.ndarray, hist: np.ndarray) -> np.ndarray:
    """What the kernel computes, as exact integers: sum_m T_m j^m k[t-9+m] - 2048 DHI (1 + j) for t = 0..n
    (multiply by UNIT for kernel units).  Needs real bytes as history (>= 9 samples)."""
    z = raw[0::2].astype(np.float64) + 1j * raw[1::2].astype(np.float64)
    pre = (hist[0::2].astype(np.float64) + 1j * hist[1::2].astype(np.float64))[-9:]
    return _fir(np.concatenate([pre, z]), z.size) - 2048.0 * DHI * (1 + 1j)


def model_tile(win: np.ndarray, taps: np.ndarray) -> np.ndarray:
    """One tile through the kernel's lane maps.  `win` = the 16 bytes before the tile followed by the
    tile's 4096 bytes.  Returns g (complex, integer parts) for outputs t = 1 .. 2048 of the tile
    (index t - 1), i.e. lane (n, h), block b, register pair r -> t = 64 n + 16 b + 8 h + 1 + r."""
    assert win.size == 16 + 2 * TILE
    u = win.astype(np.float64)
    out = np.zeros(TILE, dtype=np.complex128)
    lanes = np.arange(64)
    n, h = lanes & 31, lanes >> 5
    for b in range(4):
        acc = np.zeros((2, 32, 32))  # [digit][row][col]
        for d in range(3):
            # B[k = 8 h + j][col n] = window byte 128 n + 16 (2b + d) + 8 h + ELEM[j] of the column window,
            # whose origin is 16 bytes before the column: win index = that (win already starts 16 bytes early)
            bmat = np.zeros((16, 32))
            for lane in lanes:
                for j in range(8):
                    bmat[8 * h[lane] + j, n[lane]] = u[128 * n[lane] + 16 * (2 * b + d) + 8 * h[lane] + ELEM[j]]
            for dig in range(2):
                amat = np.zeros((32, 16))
                for lane in lanes:
                    amat[lane & 31, 8 * (lane >> 5): 8 * (lane >> 5) + 8] = taps[dig, d, lane]
                acc[dig] += amat @ bmat
        full = 2048.0 * (acc[0] - DHI) + acc[1]
        for lane in lanes:
            for reg in range(16):
                row = (reg & 3) + 8 * (reg >> 2) + 4 * h[lane]
                r, comp = reg >> 1, reg & 1
                t = 64 * n[lane] + 16 * b + 8 * h[lane] + 1 + r
                v = full[row, n[lane]]
                out[t - 1] += v * (1j if comp else 1)
    return out


def bits_from_g(g: np.ndarray) -> np.ndarray:
    """bit[t] = signbit(-(Re g[t-1] conj g[t])) for t = 0..n-1 given g[-1..n-1] (g[0] = g at t = -1)."""
    p = (g[:-1] * np.conj(g[1:])).real
    return (p > 0).astype(np.uint8)


def c0(F: np.ndarray) -> np.ndarray:
    """rd_mf_c0 in float32 arithmetic: what r = |num| - 2^-21 |b d| must exceed for a certain sign."""
    F = F.astype(np.float32)
    return (np.float32(4.0) * np.float32(E0) * F + np.float32(3.0e-10)) * np.float32(1.000002)


def threshold(F: np.ndarray) -> np.ndarray:
    """rd_mf_threshold in float32 arithmetic."""
    F = F.astype(np.float32)
    return ((F * (np.float32(4.0) * np.float32(E0) + np.float32(4.76837158e-7) * F) + np.float32(3.0e-10))
            * np.float32(1.000001))


# ---- the 8-output formulation (RD_OPT_B8, rd_mfma.h): both digits in the rows, two k-steps per block ----
def taps8_from_lib() -> np.ndarray:
    """[k-step][lane][element] as float64 (decoded f16)."""
    from rtldavis_amd import _lib
    raw = np.zeros(2 * 64 * 8, dtype=np.uint16)
    _lib.lib().rd_debug_mfma_taps8(raw.ctypes.data)
    return raw.view(np.float16).astype(np.float64).reshape(2, 64, 8)


def taps8_sparse_from_lib() -> np.ndarray:
    """The tap matrix of the 8-output formulation as the kernel hands it to the 2:4-sparse matrix instruction
    (v_smfmac_f32_32x32x32_f16; rd_mfma.h: rd_mf_taps8s), expanded to the dense [32 rows][32 K] matrix by the instruction's
    operand layout (tools/ubench/smfmac_layout.hip): lane (R, s), compressed element i -> K = 16 s + 4 (i // 2) + position,
    position = bits [2 i + 1 : 2 i] of the lane's index word; K = 16 s + 8 h + e is element e of lane half h of k-step s."""
    from rtldavis_amd import _lib
    vals = np.zeros(64 * 8, dtype=np.uint16)
    idx = np.zeros(64, dtype=np.uint32)
    _lib.lib().rd_debug_mfma_taps8s(vals.ctypes.data, idx.ctypes.data)
    v = vals.view(np.float16).astype(np.float64).reshape(64, 8)
    dense = np.zeros((32, 32))
    for lane in range(64):
        R, s = lane & 31, lane >> 5
        assert idx[lane] >> 16 == 0
        for j in range(4):
            p0, p1 = (int(idx[lane]) >> (4 * j)) & 3, (int(idx[lane]) >> (4 * j + 2)) & 3
            assert p0 < p1, "positions of a group ascending and distinct"
            dense[R, 16 * s + 4 * j + p0] += v[lane, 2 * j]
            dense[R, 16 * s + 4 * j + p1] += v[lane, 2 * j + 1]
    return dense


def model_tile8(win: np.ndarray, taps8: np.ndarray) -> np.ndarray:
    """One tile through the B8 lane maps: lane (n, h), block b = 0..7 holds - in registers 2 r' + comp (hi digit) and
    8 + 2 r' + comp (lo digit) - output t = 64 n + 8 b + 4 h + 1 + r', r' = 0..3.  Block b's window = chunks b, b + 1
    (16 bytes each) of the column's 144-byte window; the hi rows start at -DHI."""
    assert win.size == 16 + 2 * TILE
    u = win.astype(np.float64)
    out = np.zeros(TILE, dtype=np.complex128)
    lanes = np.arange(64)
    n, h = lanes & 31, lanes >> 5
    amats = []
    for d in range(2):
        amat = np.zeros((32, 16))
        for lane in lanes:
            amat[lane & 31, 8 * (lane >> 5): 8 * (lane >> 5) + 8] = taps8[d, lane]
        amats.append(amat)
    for b in range(8):
        acc = np.zeros((32, 32))
        for d in range(2):
            bmat = np.zeros((16, 32))
            for lane in lanes:
                for j in range(8):
                    bmat[8 * h[lane] + j, n[lane]] = u[128 * n[lane] + 16 * (b + d) + 8 * h[lane] + ELEM[j]]
            acc += amats[d] @ bmat
        for lane in lanes:
            for reg in range(8):
                row_hi = (reg & 3) + 8 * (reg >> 2) + 4 * h[lane]
                row_lo = ((reg + 8) & 3) + 8 * ((reg + 8) >> 2) + 4 * h[lane]
                r, comp = reg >> 1, reg & 1
                t = 64 * n[lane] + 8 * b + 4 * h[lane] + 1 + r
                v = 2048.0 * (acc[row_hi, n[lane]] - DHI) + acc[row_lo, n[lane]]
                out[t - 1] += v * (1j if comp else 1)
    return out
