// rd_mfma.h - the 9-tap rotated FIR of the demod path as a Toeplitz product on the matrix pipe.
//
// Reference stages covered (py = /root/reference/src/rtldavis/dsp.py): LUT py:26,38-39, rotate_fs4
// py:46-49, fir9 py:56-73; the numerator of discriminate py:89 and quantize py:98 follow in the kernel.
//
// 1. The Fs/4 rotation drops out.  y[n] = x[n] j^n and f[t] = sum_m c_m y[t-9+m] give
//    f[t] = j^(t-9) g[t] with the SHIFT-INVARIANT complex-tap filter g[t] = sum_m (c_m j^m) x[t-9+m],
//    and the numerator of py:89, Im(f[t-1] conj f[t]), equals -Re(g[t-1] conj g[t]).  So
//        bit[t] = signbit( -(g_re[t-1] g_re[t] + g_im[t-1] g_im[t]) ).
// 2. Everything is an integer.  x = (k - 127.4)/127.6 for a byte k, and the taps are replaced by
//    T_m = round(S c_m) for a scale S chosen (by search) so that all five distinct S c_m lie within 0.03 of an
//    integer AND D = 127.4 * (2 T_0 - 2 T_2 + T_4), the response of both components to the -127.4 offset, lies
//    within 3.6 of 2048 * D_hi with D_hi an integer (the sign of the numerator is scale invariant).  With
//    T = 2048 hi + lo (hi <= 2038, lo in [-1024, 1023]) both digits are exact in f16, and so is the raw byte:
//    read as an f16 BIT PATTERN a byte k is the subnormal k * 2^-24, which v_mfma_f32_32x32x16_f16 takes at
//    face value and at full rate (tools/ubench/mfma_subnormal.hip).  Every product and every partial sum of
//    sum_m dig_m k_m  (and of the hi sum less D_hi) is an integer below 2^24 times 2^-24, so the matrix pipe
//    accumulates it EXACTLY in f32 whatever its internal order, and
//        G[t] = 2048 * (sum hi k - D_hi) + (sum lo k)     (one fma, one rounding)
//    is the FIR output of the centred samples in units of 2^-24 S per byte unit.  The only error against
//    the true (real arithmetic) g is the tap quantisation |T_m - S c_m| summed over the nine taps times
//    |k| <= 255, plus the 3.6 of the offset: RD_MF_E0 below, 1.6e-6 byte units - 20 times below the fp32
//    VALU path's rigorous bound - plus the 2^-24 relative rounding of that one fma.
// 3. Shape.  B operand = raw samples: lane (column n = lane & 31, half h = lane >> 5) holds 8 bytes
//    of its column's window per 16-byte k-step; a mask isolates the even bytes of a dword as two f16
//    patterns, one v_perm_b32 the odd ones - no conversion arithmetic at all.  The offset enters as the C
//    operand of the first hi-digit MFMA of a block (a constant 16-register tuple).
//    A operand = constant tap matrix (32 rows = 16 outputs x {re, im}, 16 k), six fragments
//    (3 k-steps x 2 digits) kept in registers.  One 16-output block of 32 columns costs 6 MFMAs;
//    a 2048-sample tile 24.  The rows are permuted so that a lane's 16 accumulator registers are
//    re/im of EIGHT CONSECUTIVE outputs: lane (n, h), block b holds g[a0 + 16 b + 8 h + 1 + r],
//    r = 0..7, a0 = first sample of column n.  (The +1: with outputs a0+16b+1.. the 48-byte window
//    of a block starts 16-byte aligned.)  The two predecessors g[base-1], g[base] of a lane's 8-sample
//    group come from the neighbouring lane through LDS.
#pragma once
#include <stdint.h>

#define RD_MF_SCALE 18255980.028508045
#define RD_MF_T0 322807
#define RD_MF_T1 879415
#define RD_MF_T2 2234983
#define RD_MF_T3 3603886
#define RD_MF_T4 4173798
// 127.4 * (2 T0 - 2 T2 + T4) = 44 519 420.4 = 2048 * 21738 - 3.6
#define RD_MF_DHI 21738
// sum_m |T_m - S c_m| over the nine taps (exact rational arithmetic, tests/test_mfma_model.py): 0.0980034
// E0 = 2^-24 * (255 * that + 3.6) = 1.7041e-6 in units of G; 1.5 % margin
#define RD_MF_E0 1.73e-6f
// G units per byte unit of f: 2^-24 * S
#define RD_MF_G_PER_BYTE (RD_MF_SCALE / 16777216.0)
// Exactness of the f32 accumulation: sum_m |hi_m| * 255 + D_hi = 2 294 808 and sum_m |lo_m| * 255 <= 9 * 1024 * 255
// = 2 350 080, both < 2^24 = 16 777 216 (times the common 2^-24, which only moves the exponent).

// Error of a computed numerator.  Components x_hat = x + e, |e| <= E0 + 2^-24 |x_hat| (tap quantisation,
// one rounding of the digit combine).  t1 = fl(b_hat d_hat), N_hat = fl(-a_hat c_hat - t1) (one fma):
//   |a_hat c_hat - a c| <= E0 (|a_hat| + |c_hat|) + 2^-23 |a_hat c_hat| + (E0 + 2^-24 F)^2, same for b d;
//   the two roundings add 2^-24 |b_hat d_hat| + 2^-24 |N_hat|;  |a_hat c_hat| <= |N_hat| (1 + 2^-23) + |t1|.
// With F >= every |component| involved and (E0 + 2^-24 F_MAX)^2 < 1.1e-10:
//   |N_hat - N| <= 4 E0 F + 2^-22 |N_hat| + 2^-21 |t1| + 3e-10,
// so the sign of N_hat is the sign of the exact N whenever
//   r := |N_hat| - 2^-21 |t1|  >  rd_mf_c0(F) := (4 E0 F + 3e-10) * (1 + 2^-19)
// (the factor covers the 2^-22 |N_hat| term and the fp32 rounding of r and of this expression).
// The kernel tests GROUPS of eight samples with r_group = min_i |N_hat_i| - 2^-21 max_i |t1_i| <= every r_i
// (two three-operand min / max per pair of samples, one fma per group), against rd_mf_c0 at F = the largest
// |component| any input can produce and, in the rare wave that fails there, at the F of the values at hand.
// rd_mf_threshold(F) = F (4 E0 + 2^-21 F) + 3e-10 is the cruder form |N_hat| > ... with |t1| <= F^2; kept for
// the host-side model.
#if defined(__HIPCC__)
__host__ __device__ __forceinline__
#else
static inline
#endif
float rd_mf_c0(float F) { return (4.0f * RD_MF_E0 * F + 3.0e-10f) * 1.000002f; }
#if defined(__HIPCC__)
__host__ __device__ __forceinline__
#else
static inline
#endif
float rd_mf_threshold(float F) {
    return (F * (4.0f * RD_MF_E0 + 4.76837158e-7f * F) + 3.0e-10f) * 1.000001f;
}

// The largest |component| of g any input can produce: 2^-24 * 127.6 * sum_m T_m = 138.85 (kernel units,
// 127.6 byte units), and rd_mf_c0 of it, 9.61e-4 (8e-4 byte units squared): an r above that means a
// certain sign whatever the signal level.
#define RD_MF_F_MAX 138.9f
#define RD_MF_C0_MAX 9.62e-4f

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

// ------------------------------------------------------------------------------------------------------------------
// The 8-output formulation (RD_OPT_B8): both digits of a tap in ONE 32-row tile, two k-steps per block.
// A block of 8 outputs needs a window of 8 + 8 samples = 32 bytes = two 16-byte k-steps (the 16-output block above needs
// 48 bytes = three), and its 32 rows are 8 outputs x {re, im} x {hi digit, lo digit}: TWO MFMAs per 8 outputs and 32
// columns instead of six per 16 - a third fewer matrix instructions per sample, one accumulator tuple instead of two,
// two tap fragments instead of six.  Row R of the tile lands in lane half (R >> 2) & 1, register rho = (R & 3) + 4 (R >> 3);
// rho = 8 dig + 2 r' + comp and the output is p = 4 half + r': a lane holds re / im and both digits of FOUR consecutive
// outputs, g[a0 + 8 b + 4 h + 1 + r'], r' = 0..3 (a0 = first sample of column n, b = block 0..7, h = lane half), and
// G = 2048 * acc[2 r' + comp] + acc[8 + 2 r' + comp] as before.  The hi rows start at -D_hi, the lo rows at 0 (one constant
// C tuple).  Exactness: unchanged - every row is one digit's sum.
struct alignas(16) rd_mf_taps8 {
    uint16_t v[2][64][8];  // [k-step d of a block][lane][element j] as f16 bit patterns
};

// window byte w (block-relative, 0..31) in row R
constexpr int rd_mf8_coef(int R, int w, int *dig_out) {
    const int half = (R >> 2) & 1, rho = (R & 3) + 4 * (R >> 3);
    const int dig = rho >> 3, r = (rho & 7) >> 1, comp = rho & 1, p = 4 * half + r;
    *dig_out = dig;
    const int m = (w >> 1) - p, isq = w & 1;
    if (m < 0 || m > 8) return 0;
    const int T[5] = {RD_MF_T0, RD_MF_T1, RD_MF_T2, RD_MF_T3, RD_MF_T4};
    const int t = T[m <= 4 ? m : 8 - m];
    const int ph = m & 3;
    int sgn = 0;
    if (comp == 0) sgn = isq ? (ph == 1 ? -1 : ph == 3 ? 1 : 0) : (ph == 0 ? 1 : ph == 2 ? -1 : 0);
    else sgn = isq ? (ph == 0 ? 1 : ph == 2 ? -1 : 0) : (ph == 1 ? 1 : ph == 3 ? -1 : 0);
    return sgn * t;
}

constexpr rd_mf_taps8 rd_mf_make_taps8() {
    rd_mf_taps8 t = {};
    for (int d = 0; d < 2; d++)
        for (int lane = 0; lane < 64; lane++)
            for (int j = 0; j < 8; j++) {
                const int R = lane & 31, hk = lane >> 5;
                const int w = 16 * d + 8 * hk + RD_MF_ELEM(j);
                int dig = 0;
                const int c = rd_mf8_coef(R, w, &dig);
                const int a = c < 0 ? -c : c;
                const int hi = (a + 1024) >> 11, lo = a - hi * 2048;
                const int v = dig == 0 ? hi : lo;
                t.v[d][lane][j] = rd_mf_f16_of_int(c < 0 ? -v : v);
            }
    return t;
}

// ------------------------------------------------------------------------------------------------------------------
// The same tile on the 2:4-SPARSE matrix instruction (v_smfmac_f32_32x32x32_f16: K = 32 in the time of a dense K = 16).
// A real tap times j^n touches EITHER the I or the Q byte of a sample, never both, and a group of four consecutive
// elements of a k-step is (I_a, I_b, Q_a, Q_b) of two neighbouring samples (RD_MF_ELEM): at most two of the four are
// non-zero in any row - the A matrix is 2:4 sparse by construction, and a block's two k-steps become ONE instruction.
// Operand layout (tools/ubench/smfmac_layout.hip, found by trying hypotheses against the instruction):
//   B  lane (n, h), element i:      i < 8 -> K = 8 h + i (the first k-step's bytes of this lane half),
//                                   i >= 8 -> K = 16 + 8 h + (i - 8) (the second k-step's) = the two dense B fragments
//   A  lane (R, s), element i:      compressed slot of k-step s, group j = i / 2 (K = 16 s + 4 j .. + 3), its first (i even)
//                                   or second (i odd) kept element; position in the group = bits [2 i + 1 : 2 i] of idx
//   D  as the dense instruction.
// A kept element's value may be zero (a group with fewer than two taps): positions are then filled in ascending order.
struct alignas(16) rd_mf_taps8s {
    uint16_t v[64][8];   // [lane][compressed element] as f16 bit patterns
    uint32_t idx[64];    // [lane] sixteen bits: two per element
};

constexpr rd_mf_taps8s rd_mf_make_taps8s(bool *ok) {
    rd_mf_taps8s t = {};
    *ok = true;
    for (int lane = 0; lane < 64; lane++) {
        const int R = lane & 31, s = lane >> 5;
        uint32_t idx = 0;
        for (int j = 0; j < 4; j++) {              // group j of k-step s
            int pos[2] = {-1, -1}, val[2] = {0, 0}, n = 0;
            for (int q = 0; q < 4; q++) {          // dense element kk = 4 j + q of the k-step: lane half kk / 8, element kk % 8
                const int kk = 4 * j + q;
                const int w = 16 * s + 8 * (kk >> 3) + RD_MF_ELEM(kk & 7);
                int dig = 0;
                const int c = rd_mf8_coef(R, w, &dig);
                const int a = c < 0 ? -c : c;
                const int hi = (a + 1024) >> 11, lo = a - hi * 2048;
                const int v = c < 0 ? -(dig == 0 ? hi : lo) : (dig == 0 ? hi : lo);
                if (v != 0) {
                    if (n < 2) { pos[n] = q; val[n] = v; }
                    n++;
                }
            }
            if (n > 2) *ok = false;
            // fill up with zero-valued elements, positions ascending and distinct
            if (n == 0) { pos[0] = 0; pos[1] = 1; }
            else if (n == 1) {
                if (pos[0] < 3) { pos[1] = pos[0] + 1; val[1] = 0; }
                else { pos[1] = pos[0]; val[1] = val[0]; pos[0] = 2; val[0] = 0; }
            }
            t.v[lane][2 * j] = rd_mf_f16_of_int(val[0]);
            t.v[lane][2 * j + 1] = rd_mf_f16_of_int(val[1]);
            idx |= (uint32_t)(pos[0] | (pos[1] << 2)) << (4 * j);
        }
        t.idx[lane] = idx;
    }
    return t;
}
