// rd_mfma.h - the 9-tap rotated FIR of the demod path as a Toeplitz product on the matrix pipe.
//
// Reference stages covered (py = /root/reference/src/rtldavis/dsp.py): LUT py:26,38-39, rotate_fs4
// py:46-49, fir9 py:56-73; the numerator of discriminate py:89 and quantize py:98 follow in the kernel.
//
// 1. The Fs/4 rotation drops out.  y[n] = x[n] j^n and f[t] = sum_m c_m y[t-9+m] give
//    f[t] = j^(t-9) g[t] with the SHIFT-INVARIANT complex-tap filter g[t] = sum_m (c_m j^m) x[t-9+m],
//    and the numerator of py:89, Im(f[t-1] conj f[t]), equals -Re(g[t-1] conj g[t]).  So
//        bit[t] = signbit( -(g_re[t-1] g_re[t] + g_im[t-1] g_im[t]) ).
// 2. Everything is an integer.  x = (k - 127.4)/127.6 = U/638 with U = 5k - 637, and the taps are
//    replaced by T_m = round(S c_m) for a scale S chosen so that all five distinct S c_m lie within
//    0.02 of an integer (found by search; the sign of the numerator is scale invariant).  With
//    T = 2048 hi + lo (hi <= 1974, lo in [-1024, 1023]) both digits and U are exact in f16, every
//    product and every partial sum of  sum_m dig_m U_m  is an integer below 2^24 (bounds below), so
//    v_mfma_f32_32x32x16_f16 accumulates it EXACTLY in f32 whatever its internal order, and
//        G[t] = 2048 * (sum hi U) + (sum lo U)     (one fma, one rounding)
//    is the FIR output in units of 2^-12 * 5 S / 1 byte-unit.  The only error against the true
//    (real arithmetic) g is the tap quantisation, |T_m - S c_m| summed over the nine taps:
//    RD_MF_E0 below, 4.9e-7 byte units - 70 times below the fp32 VALU path's rigorous bound - plus the
//    2^-24 relative rounding of that one fma.
// 3. Shape.  B operand = raw samples: lane (column n = lane & 31, half h = lane >> 5) holds 8 bytes
//    of its column's window per 16-byte k-step, converted to f16 (5k-637) * 2^-12 by one mask and one
//    v_pk_fma_f16 per two bytes (a byte b read as an f16 bit pattern is the subnormal b * 2^-24).
//    A operand = constant tap matrix (32 rows = 16 outputs x {re, im}, 16 k), six fragments
//    (3 k-steps x 2 digits) kept in registers.  One 16-output block of 32 columns costs 6 MFMAs;
//    a 2048-sample tile 24.  The rows are permuted so that a lane's 16 accumulator registers are
//    re/im of EIGHT CONSECUTIVE outputs: lane (n, h), block b holds g[a0 + 16 b + 8 h + 1 + r],
//    r = 0..7, a0 = first sample of column n.  (The +1: with outputs a0+16b+1.. the 48-byte window
//    of a block starts 16-byte aligned.)  The two predecessors g[base-1], g[base] of a lane's 8-sample
//    group come from the neighbouring lane through LDS.
#pragma once
#include <stdint.h>

#define RD_MF_SCALE 17683709.98098367
#define RD_MF_T0 312688
#define RD_MF_T1 851848
#define RD_MF_T2 2164923
#define RD_MF_T3 3490915
#define RD_MF_T4 4042962
// sum_m |T_m - S c_m| over the nine taps (exact rational arithmetic, tests/test_mfma_model.py): 0.0677192
// E0 = 2^-12 * 638 * that = 0.010548 in units of G; 1.5 % margin
#define RD_MF_E0 0.0107f
// G units per byte unit of f: 2^-12 * 5 * S
#define RD_MF_G_PER_BYTE (5.0 * RD_MF_SCALE / 4096.0)
// Exactness of the f32 accumulation: sum_m |hi_m| * 638 = 5 509 768 and sum_m |lo_m| * 638 <= 9 * 1024 * 638
// = 5 879 808, both < 2^24 = 16 777 216 (times the common 2^-12, which only moves the exponent).

// Error of a computed numerator.  Components x_hat = x + e, |e| <= E0 + 2^-24 |x_hat| (tap quantisation,
// one rounding of the digit combine).  t1 = fl(b_hat d_hat), N_hat = fl(-a_hat c_hat - t1) (one fma):
//   |a_hat c_hat - a c| <= E0 (|a_hat| + |c_hat|) + 2^-23 |a_hat c_hat| + (E0 + 2^-24 F)^2, same for b d;
//   the two roundings add 2^-24 |b_hat d_hat| + 2^-24 |N_hat|;  |a_hat c_hat| <= |N_hat| (1 + 2^-23) + |t1|.
// With F >= every |component| involved and (E0 + 2^-24 F_MAX)^2 < 0.031:
//   |N_hat - N| <= 4 E0 F + 2^-22 |N_hat| + 2^-21 |t1| + 0.07,
// so the sign of N_hat is the sign of the exact N whenever
//   r := |N_hat| - 2^-21 |t1|  >  rd_mf_c0(F) := (4 E0 F + 0.07) * (1 + 2^-19)
// (the factor covers the 2^-22 |N_hat| term and the fp32 rounding of r and of this expression).
// rd_mf_threshold(F) = F (4 E0 + 2^-21 F) + 3e-4 is the cruder form |N_hat| > ... with |t1| <= F^2; kept for
// the host-side model.
#if defined(__HIPCC__)
__host__ __device__ __forceinline__
#else
static inline
#endif
float rd_mf_c0(float F) { return (4.0f * RD_MF_E0 * F + 0.07f) * 1.000002f; }
#if defined(__HIPCC__)
__host__ __device__ __forceinline__
#else
static inline
#endif
float rd_mf_threshold(float F) {
    return (F * (4.0f * RD_MF_E0 + 4.76837158e-7f * F) + 3.0e-4f) * 1.000001f;
}

// The largest |component| of g any input can produce: 2^-12 * 638 * sum_m T_m = 2 754 468 (kernel units,
// 127.6 byte units), and the threshold that goes with it (3.74e6 = 0.008 byte units squared): a numerator
// above it has a certain sign whatever the signal level.
#define RD_MF_F_MAX 2754500.0f
#define RD_MF_THR_MAX 3.74e6f
// rd_mf_c0(RD_MF_F_MAX) = 117 893: 2.5e-4 byte units squared
#define RD_MF_C0_MAX 1.179e5f

// element j of a lane's 8-element B fragment is byte RD_MF_ELEM(j) of the 8 window bytes the lane
// reads: registers (b0,b2) (b1,b3) (b4,b6) (b5,b7) - the even bytes of a dword come out of one AND.
#define RD_MF_ELEM(j) ((((j) & 4)) | (((j) & 1) << 1) | (((j) >> 1) & 1))

struct alignas(16) rd_mf_taps {
    // [digit: 0 = hi, 1 = lo][k-step d of a block][lane][element j] as f16 bit patterns
    uint16_t v[2][3][64][8];
};

// f16 bit pattern of a small integer (|v| <= 2048: exact)
constexpr uint16_t rd_mf_f16_of_int(int v) {
    if (v == 0) return 0;
    const uint16_t sign = v < 0 ? 0x8000 : 0;
    unsigned a = v < 0 ? (unsigned)(-v) : (unsigned)v;
    int e = 0;
    while ((a >> (e + 1)) != 0) e++;  // a in [2^e, 2^(e+1))
    const unsigned mant = e <= 10 ? (a << (10 - e)) & 0x3FF : (a >> (e - 10)) & 0x3FF;
    return (uint16_t)(sign | ((unsigned)(e + 15) << 10) | mant);
}

// Coefficient of window byte w (block-relative, 0..47) in output row R of the tap matrix, as the
// signed integer tap T (0 when the byte is outside the output's nine samples or belongs to the
// other component).  Row R of the 32x32 accumulator tile lands in lane half (R >> 2) & 1, register
// rho = (R & 3) + 4 (R >> 3); rho = 2 r + comp, output p = 8 * half + r of the block.
constexpr int rd_mf_coef(int R, int w) {
    const int half = (R >> 2) & 1, rho = (R & 3) + 4 * (R >> 3);
    const int r = rho >> 1, comp = rho & 1, p = 8 * half + r;
    const int m = (w >> 1) - p, isq = w & 1;
    if (m < 0 || m > 8) return 0;
    const int T[5] = {RD_MF_T0, RD_MF_T1, RD_MF_T2, RD_MF_T3, RD_MF_T4};
    const int t = T[m <= 4 ? m : 8 - m];
    // c_m j^m (I + jQ): re takes I * {+,0,-,0}[m&3] and Q * {0,-,0,+}[m&3];
    //                   im takes I * {0,+,0,-}[m&3] and Q * {+,0,-,0}[m&3]
    const int ph = m & 3;
    int sgn = 0;
    if (comp == 0) sgn = isq ? (ph == 1 ? -1 : ph == 3 ? 1 : 0) : (ph == 0 ? 1 : ph == 2 ? -1 : 0);
    else sgn = isq ? (ph == 0 ? 1 : ph == 2 ? -1 : 0) : (ph == 1 ? 1 : ph == 3 ? -1 : 0);
    return sgn * t;
}

constexpr rd_mf_taps rd_mf_make_taps() {
    rd_mf_taps t = {};
    for (int d = 0; d < 3; d++)
        for (int lane = 0; lane < 64; lane++)
            for (int j = 0; j < 8; j++) {
                const int R = lane & 31, hk = lane >> 5;
                const int w = 16 * d + 8 * hk + RD_MF_ELEM(j);
                const int c = rd_mf_coef(R, w);
                const int a = c < 0 ? -c : c;
                const int hi = (a + 1024) >> 11, lo = a - hi * 2048;
                t.v[0][d][lane][j] = rd_mf_f16_of_int(c < 0 ? -hi : hi);
                t.v[1][d][lane][j] = rd_mf_f16_of_int(c < 0 ? -lo : lo);
            }
    return t;
}
