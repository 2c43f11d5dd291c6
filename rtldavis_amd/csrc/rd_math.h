// rd_math.h - arithmetic of the IQ -> sign-bit path, shared by the HIP kernels.
//
// Every function is `RD_HD` (host + device) so tests can run the very same code on the CPU
// (tests/host_harness.cpp); the shipped library only ever calls them from device code.
//
// Reference stages covered (py = /root/reference/src/rtldavis/dsp.py):
//   LUT (k-127.4)/127.6 py:26,38-39 ; rotate_fs4 py:46-49 ; fir9 py:56-73 ;
//   discriminate numerator py:89 ; quantize (sign bit) py:98.
//
// Two evaluations of the same sign:
//   * fast: fp32, on raw byte values k (0..255), scale 127.6 dropped (sign is scale
//     invariant), centring folded into a per-phase constant, Fs/4 rotation folded into
//     register naming (swap) and signed tap constants.  Comes with a rigorous error bound
//     (RD_E_ABS / rd_run_threshold) so a run of samples whose smallest |numerator| is
//     inside the bound is re-evaluated exactly.
//   * exact: integers.  x = (5k-637)/638 and taps*1e12 are integers, so the FIR is exact
//     in int64 (|F| < 6.4e14) and the numerator in __int128.  The sign of the exact value
//     equals the reference's float64 sign unless |d_ref| is within float64 rounding of 0
//     (~1e-16 relative); exact zeros follow IEEE signed-zero rules of py:89 (see rd_exact_bit).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define RD_HD __host__ __device__ __forceinline__
#define RD_HDM __host__ __device__ __forceinline__
#else
#define RD_HD static inline
#define RD_HDM inline
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define RD_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define RD_SCHED_FENCE() do { } while (0)
#endif

#define RD_RUN 32          // samples per lane run (one packed output word)
#define RD_HALO 10         // f[t0-1] needs y[t0-10 .. t0-2]
#define RD_WIN (RD_RUN + 9) // samples a run reads: t0-10 .. t0+30

// fir9 taps (py:56-69).  Symmetric: c[m] == c[8-m].
#define RD_C0 0.017682261285
#define RD_C1 0.048171339939
#define RD_C2 0.122424706672
#define RD_C3 0.197408519126
#define RD_C4 0.228626345955
// sum_m c_m j^m = 2c0 - 2c2 + c4 (real): response of the filter to the rotated DC term.
#define RD_CDC (2.0 * RD_C0 - 2.0 * RD_C2 + RD_C4)
#define RD_DC ((float)(127.4 * RD_CDC))

// ---- rigorous fp32 error bound (units: raw byte values) -------------------------------
// f_hat = fma chain  acc0 = -+D; acc_{k+1} = fma(T_k, s_k, acc_k), k = 0..4, with
// T_k = +-fl32(c_k) and exact small integers s_k: s0 = w1+w9, s2 = w3+w7 in [0, 510] (raw bytes
// are not negative), s1 = w2-w8, s3 = w4-w6 in [-255, 255], s4 = w5 in [0, 255].  Two samples
// apart the Fs/4 rotation flips the sign, so T0, T2, T4 alternate (+,-,+ or -,+,-) and acc0
// (the rotated DC term, |D| = 2.44) opposes T0; T1, T3 take either sign.  Interval bounds of the
// partial sums for the pattern (+,.,-,.,+), the other one being its mirror image:
//   p0 = -D + c0 s0          in [ -2.44,   6.58]          |p0| <=   6.58
//   p1 = p0 +- c1 s1         in [-14.72,  18.86]          |p1| <=  18.86
//   p2 = p1 - c2 s2          in [-77.16,  18.86]          |p2| <=  77.16
//   p3 = p2 +- c3 s3         in [-127.5,  69.20]          |p3| <= 127.50
//   p4 = p3 + c4 s4          in [-127.5, 127.50]          |p4| <= 127.50
//   rounding of the 5 fmas : <= 2^-24 * sum |p_k| = 2^-24 * 357.6            = 2.13e-5
//   tap rounding           : <= 2^-24 * sum_k c_k max|s_k| = 2^-24 * 192.3   = 1.15e-5
//   rounding of D          : <= 2^-23 (ulp(2.44)/2)                          = 1.2e-7
// total < 3.30e-5; RD_E_ABS adds margin for the second-order terms of the numerator bound.
#define RD_E_ABS 3.5e-5f
// With a,b,c,d the true components (|.| <= F + E, F = max |component of f_hat| over the run):
// |num_hat - num| <= E(|a|+|b|+|c|+|d|) + 2E^2 + 3*2^-24 F^2 <= F*(4E + 2^-22 F) + 6E^2.
// 6E^2 < 2e-8; the additive 1e-7 and the factor (1 + 2^-20) cover it and the fp32
// rounding of this expression itself.
RD_HD float rd_run_threshold(float F) {
    return (F * (4.0f * RD_E_ABS + 2.3841858e-7f * F) + 1.0e-7f) * 1.000001f;
}

// (x, y) = (re, im) or (im, re) of one sample / FIR output.  On the device a 64-bit register
// pair so that v_pk_add_f32 / v_pk_fma_f32 can take it whole.
#if defined(__HIP_DEVICE_COMPILE__)
typedef float rd_f2 __attribute__((ext_vector_type(2)));
#else
struct rd_f2 {
    float x, y;
};
#endif

#define RD_GROUP 8                      // guard-band granularity: samples per re-evaluated group
#define RD_GROUPS (RD_RUN / RD_GROUP)   // groups per run (one output byte each)

struct rd_run_result {
    uint32_t word;           // bit r = sign bit of num[t0 + r]   (1 = negative, py:98)
    float fmax;              // max |component| of f_hat over f[t0-1 .. t0+31]
    float nmin[RD_GROUPS];   // min |num_hat| over each group of 8 samples
};

// bit g set = group g (samples t0+8g .. t0+8g+7) must be re-evaluated exactly
RD_HD uint32_t rd_guard_mask(const rd_run_result &r);

// Signs of rot = j^p applied to (I, Q):  p=0 ( I, Q)  p=1 (-Q, I)  p=2 (-I,-Q)  p=3 ( Q,-I).
// With w = (I,Q) for even p and (Q,I) for odd p, y = (sr*w.x, si*w.y) with sr = -1 for
// p in {1,2} and si = -1 for p in {2,3} (RD_NX / RD_NY in rd_fir_out).

// Fast fp32 evaluation of one run.  `win` holds the raw bytes of samples t0-10 .. t0+30
// (2*RD_WIN bytes, I then Q); t0 % 4 == 0 in absolute stream time.  Fully unrolled by the
// compiler: every phase / tap / sign below is a compile-time constant per unrolled step.
// `Src::f(i)` returns byte i of that window converted to float (on the device: one
// v_cvt_f32_ubyteN on a register dword, kept opaque to the optimiser - see rd_kernels.hip).
struct rd_ptr_src {
    const uint8_t *p;
    RD_HDM float f(int i) const { return (float)p[i]; }
};

// Issue model measured on MI355X (tools/ubench/valu_*.hip, profiles/): a SIMD issues one VALU
// instruction per 2 cycles whatever its class; the "slow" class (conversions, min/max,
// alignbit, packed fp32 ...) additionally needs 4 cycles between two of its own.  So the
// instruction COUNT is what matters as long as slow-class ops stay below half of the mix:
// RD_NPK of the nine FIR steps per output (4 pair sums, 5 multiply-adds) are issued as packed
// fp32 (one instruction for re and im), the rest as scalar pairs.  Results are identical
// either way (same IEEE operations per component).
#ifndef RD_FENCE_EVERY
#define RD_FENCE_EVERY 1
#endif
#ifndef RD_NPK_DEFAULT
#define RD_NPK_DEFAULT 0
#endif

// s = a + b or a - b (exact: small integers)
template <bool SUB, bool PACKED>
RD_HD rd_f2 rd_pair(rd_f2 a, rd_f2 b) {
    rd_f2 s;
#if defined(__HIP_DEVICE_COMPILE__)
    if (PACKED) {
        if (SUB) asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(s) : "v"(a), "v"(b));
        else asm("v_pk_add_f32 %0, %1, %2" : "=v"(s) : "v"(a), "v"(b));
        return s;
    }
#endif
    if (SUB) { s.x = a.x - b.x; s.y = a.y - b.y; } else { s.x = a.x + b.x; s.y = a.y + b.y; }
    return s;
}

// acc + (NX ? -c : c) * s.x, acc + (NY ? -c : c) * s.y   (one rounding per component)
// The packed form writes a fresh register pair (separate output operand) so that a constant
// `acc` (the DC term of the first tap) needs no copy.
template <bool NX, bool NY, bool PACKED>
RD_HD rd_f2 rd_tap(float c, rd_f2 s, rd_f2 acc) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (PACKED) {
        rd_f2 cc = {c, c};
        rd_f2 r;
        if (NX && NY) asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(r) : "s"(cc), "v"(s), "v"(acc));
        else if (NX) asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0]" : "=v"(r) : "s"(cc), "v"(s), "v"(acc));
        else if (NY) asm("v_pk_fma_f32 %0, %1, %2, %3 neg_hi:[1,0,0]" : "=v"(r) : "s"(cc), "v"(s), "v"(acc));
        else asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "s"(cc), "v"(s), "v"(acc));
        return r;
    }
#endif
    acc.x = __builtin_fmaf(NX ? -c : c, s.x, acc.x);
    acc.y = __builtin_fmaf(NY ? -c : c, s.y, acc.y);
    return acc;
}

// One FIR output f[t0+R] (R = -1..31) from the converted window w[]: compile-time R so that
// every phase, sign and tap is a constant.
template <int R, int RD_NPK>
RD_HD rd_f2 rd_fir_out(const rd_f2 *w) {
    // f[t0+R] = sum_m c_m y[t0+R-9+m]; window index of tap m is i = R+1+m
    constexpr int q = (R + 3 + 4) & 3;  // (t-9) mod 4 for the DC term
    // (1+j) j^q : q0 (1,1) q1 (-1,1) q2 (-1,-1) q3 (1,-1); acc starts at -D
    rd_f2 acc;
    acc.x = (q == 1 || q == 2) ? RD_DC : -RD_DC;
    acc.y = (q == 2 || q == 3) ? RD_DC : -RD_DC;
    // phase p of sample i: signs sr(p) = -1 for p in {1,2}, si(p) = -1 for p in {2,3}
#define RD_PH(i) (((i) + 2) & 3)
#define RD_NX(i) (RD_PH(i) == 1 || RD_PH(i) == 2)
#define RD_NY(i) (RD_PH(i) == 2 || RD_PH(i) == 3)
    // packed steps are spread over the nine: pair sums first (0..3), then taps 0..4
    const rd_f2 s0 = rd_pair<false, (RD_NPK > 0)>(w[R + 1], w[R + 9]);
    const rd_f2 s1 = rd_pair<true, (RD_NPK > 1)>(w[R + 2], w[R + 8]);
    const rd_f2 s2 = rd_pair<false, (RD_NPK > 2)>(w[R + 3], w[R + 7]);
    const rd_f2 s3 = rd_pair<true, (RD_NPK > 3)>(w[R + 4], w[R + 6]);
    acc = rd_tap<RD_NX(R + 1), RD_NY(R + 1), (RD_NPK > 4)>((float)RD_C0, s0, acc);
    acc = rd_tap<RD_NX(R + 2), RD_NY(R + 2), (RD_NPK > 5)>((float)RD_C1, s1, acc);
    acc = rd_tap<RD_NX(R + 3), RD_NY(R + 3), (RD_NPK > 6)>((float)RD_C2, s2, acc);
    acc = rd_tap<RD_NX(R + 4), RD_NY(R + 4), (RD_NPK > 7)>((float)RD_C3, s3, acc);
    acc = rd_tap<RD_NX(R + 5), RD_NY(R + 5), (RD_NPK > 8)>((float)RD_C4, w[R + 5], acc);
#undef RD_PH
#undef RD_NX
#undef RD_NY
    return acc;
}

RD_HD float rd_max3abs(float m, float x, float y);
RD_HD float rd_min3abs(float m, float a, float b);
RD_HD uint32_t rd_shift_in_sign(uint32_t word, float num);
RD_HD uint32_t rd_bitrev32(uint32_t v);

struct rd_run_state {
    rd_f2 prev;
    float fmaxv, num_even;
    float nminv[RD_GROUPS];
    uint32_t word;
};

// output R of the run: FIR, guard statistics, sign bit
template <int R, int NPK, class Src>
RD_HD void rd_fast_step(const Src &win, rd_f2 *w, rd_run_state &st) {
    {   // sample R+9 of the window is converted right before its first use
        constexpr int i = R + 9;
        const float kI = win.f(2 * i), kQ = win.f(2 * i + 1);
        if ((i + 2) & 1) { w[i].x = kQ; w[i].y = kI; } else { w[i].x = kI; w[i].y = kQ; }
    }
    const rd_f2 acc = rd_fir_out<R, NPK>(w);
    st.fmaxv = rd_max3abs(st.fmaxv, acc.x, acc.y);
    if (R >= 0) {
        // numerator of py:89: imag_n*real_np - real_n*imag_np, n = f[t-1], np = f[t]
        const float num = __builtin_fmaf(-st.prev.x, acc.y, st.prev.y * acc.x);
        if (R & 1) st.nminv[(R < 0 ? 0 : R) / RD_GROUP] = rd_min3abs(st.nminv[(R < 0 ? 0 : R) / RD_GROUP], st.num_even, num);
        else st.num_even = num;
        st.word = rd_shift_in_sign(st.word, num);  // one v_alignbit_b32; reversed after the loop
    }
    st.prev = acc;
    // keep conversions next to their first use (register pressure): a scheduling fence every
    // RD_FENCE_EVERY outputs
    if ((R & (RD_FENCE_EVERY - 1)) == RD_FENCE_EVERY - 1) RD_SCHED_FENCE();
}

template <int R, int NPK, class Src>
struct rd_fast_unroll {
    static RD_HDM void run(const Src &win, rd_f2 *w, rd_run_state &st) {
        rd_fast_step<R, NPK>(win, w, st);
        rd_fast_unroll<R + 1, NPK, Src>::run(win, w, st);
    }
};
template <int NPK, class Src>
struct rd_fast_unroll<RD_RUN, NPK, Src> {
    static RD_HDM void run(const Src &, rd_f2 *, rd_run_state &) {}
};

// Fast fp32 evaluation of one run.  `win` holds the raw bytes of samples t0-10 .. t0+30
// (2*RD_WIN bytes, I then Q); t0 % 4 == 0 in absolute stream time.
template <int NPK = RD_NPK_DEFAULT, class Src>
RD_HD rd_run_result rd_fast_run(const Src &win) {
    rd_f2 w[RD_WIN];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const float kI = win.f(2 * i), kQ = win.f(2 * i + 1);
        if ((i + 2) & 1) { w[i].x = kQ; w[i].y = kI; } else { w[i].x = kI; w[i].y = kQ; }
    }
    rd_run_state st;
    st.prev.x = 0.0f; st.prev.y = 0.0f;
    st.fmaxv = 0.0f; st.num_even = 0.0f; st.word = 0;
#pragma unroll
    for (int g = 0; g < RD_GROUPS; g++) st.nminv[g] = 3.0e38f;
    rd_fast_unroll<-1, NPK, Src>::run(win, w, st);
    rd_run_result out;
    out.word = rd_bitrev32(st.word);  // bit r = sign of num[t0+r]
    out.fmax = st.fmaxv;
#pragma unroll
    for (int g = 0; g < RD_GROUPS; g++) out.nmin[g] = st.nminv[g];
    return out;
}

RD_HD uint32_t rd_bitrev32(uint32_t v) {
#if defined(__clang__)
    return __builtin_bitreverse32(v);
#else
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
    v = ((v >> 8) & 0x00FF00FFu) | ((v & 0x00FF00FFu) << 8);
    return (v >> 16) | (v << 16);
#endif
}

// max(m, |x|, |y|) and min(m, |a|, |b|) as ONE instruction each.  Written with fmaxf/fminf the
// compiler adds a canonicalising v_max_f32 per operand (three slow-pipe ops per sample instead
// of one); the operands here are never NaN-sensitive (a NaN ends up flagged by rd_guard_mask).
RD_HD float rd_max3abs(float m, float x, float y) {
#if defined(__HIP_DEVICE_COMPILE__)
    float r;
    asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(r) : "v"(m), "v"(x), "v"(y));
    return r;
#else
    return __builtin_fmaxf(m, __builtin_fmaxf(__builtin_fabsf(x), __builtin_fabsf(y)));
#endif
}
RD_HD float rd_min3abs(float m, float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    float r;
    asm("v_min3_f32 %0, %1, |%2|, |%3|" : "=v"(r) : "v"(m), "v"(a), "v"(b));
    return r;
#else
    return __builtin_fminf(m, __builtin_fminf(__builtin_fabsf(a), __builtin_fabsf(b)));
#endif
}

// (word << 1) | signbit(num)
RD_HD uint32_t rd_shift_in_sign(uint32_t word, float num) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(word, __builtin_bit_cast(uint32_t, num), 31);
#else
    return (word << 1) | (__builtin_bit_cast(uint32_t, num) >> 31);
#endif
}

RD_HD uint32_t rd_guard_mask(const rd_run_result &r) {
    const float thr = rd_run_threshold(r.fmax);
    uint32_t m = 0;
#pragma unroll
    for (int g = 0; g < RD_GROUPS; g++) m |= (r.nmin[g] > thr) ? 0u : (1u << g);  // NaN -> flagged
    return m;
}

// ---- exact evaluation ---------------------------------------------------------------
// taps * 1e12 (exact integers)
#define RD_T0 17682261285LL
#define RD_T1 48171339939LL
#define RD_T2 122424706672LL
#define RD_T3 197408519126LL
#define RD_T4 228626345955LL

struct rd_i2 {
    int64_t x, y;
};

// Rotated integer sample U*j^p, U = 5k-637 (x = U/638), p = absolute index mod 4.
RD_HD rd_i2 rd_rot_int(int kI, int kQ, int p) {
    const int64_t a = 5 * kI - 637, b = 5 * kQ - 637;
    rd_i2 y;
    switch (p & 3) {
        case 0: y.x = a; y.y = b; break;
        case 1: y.x = -b; y.y = a; break;
        case 2: y.x = -a; y.y = -b; break;
        default: y.x = b; y.y = -a; break;
    }
    return y;
}

// Exact FIR output (scaled by 638e12) from nine rotated integer samples y[0..8] (oldest first).
RD_HD rd_i2 rd_fir_int(const rd_i2 *y) {
    rd_i2 f;
    f.x = RD_T0 * (y[0].x + y[8].x) + RD_T1 * (y[1].x + y[7].x) + RD_T2 * (y[2].x + y[6].x) +
          RD_T3 * (y[3].x + y[5].x) + RD_T4 * y[4].x;
    f.y = RD_T0 * (y[0].y + y[8].y) + RD_T1 * (y[1].y + y[7].y) + RD_T2 * (y[2].y + y[6].y) +
          RD_T3 * (y[3].y + y[5].y) + RD_T4 * y[4].y;
    return f;
}

// Sign bit of d = (ni*pr - nr*pi)/(|n|^2 + 1e-10) as the reference's float64 code produces it
// (py:89,98).  Non-zero exact numerator: its sign.  Exact zero: the float64 expression is
// fl(ni*pr) - fl(nr*pi); with n == 0 (zero history after reset) the products are signed
// zeros and the difference is -0.0 exactly when pr < 0 and pi > 0 (fixture
// tests/golden/startup_quadrants.json); otherwise equal products give +0.0.
RD_HD uint32_t rd_exact_bit(rd_i2 n, rd_i2 np) {
    const __int128 num = (__int128)n.y * np.x - (__int128)n.x * np.y;
    if (num < 0) return 1u;
    if (num > 0) return 0u;
    if (n.x == 0 && n.y == 0) return (np.x < 0 && np.y > 0) ? 1u : 0u;
    return 0u;
}

// Byte fetch used by the exact path: sample index n relative to the stream's first sample.
// Samples before `valid_from` (<= 0) are the zero history after reset and read as y = 0.
struct rd_stream_view {
    const uint8_t *base;  // &iq[0] of sample 0 (bytes before it hold history when valid_from < 0)
    long valid_from;      // first readable sample index (0: zero history; negative: history bytes)
    long n;               // number of samples
};

RD_HD rd_i2 rd_sample_int(const rd_stream_view &v, long n) {
    if (n < v.valid_from) {
        rd_i2 z = {0, 0};
        return z;
    }
    const uint8_t *p = v.base + 2 * n;
    return rd_rot_int(p[0], p[1], (int)(n & 3));
}

// The same decision through float64: U and taps*1e12 are integers below 2^53, so the FIR sums
// are EXACT in double (|F| < 6.4e14).  The two products of the numerator are rounded once each
// (relative 2^-53); when their difference clears 2^-51 (|p1| + |p2|) its sign is certain,
// otherwise (an event of probability ~1e-15 per sample) the __int128 path decides.  About four
// times fewer cycles than 64/128-bit integer arithmetic on the VALU.
struct rd_dd2 {
    double x, y;
};

RD_HD rd_dd2 rd_fir_f64_exact(const rd_dd2 *y) {
    rd_dd2 f;
    f.x = (double)RD_T0 * (y[0].x + y[8].x) + (double)RD_T1 * (y[1].x + y[7].x) + (double)RD_T2 * (y[2].x + y[6].x) +
          (double)RD_T3 * (y[3].x + y[5].x) + (double)RD_T4 * y[4].x;
    f.y = (double)RD_T0 * (y[0].y + y[8].y) + (double)RD_T1 * (y[1].y + y[7].y) + (double)RD_T2 * (y[2].y + y[6].y) +
          (double)RD_T3 * (y[3].y + y[5].y) + (double)RD_T4 * y[4].y;
    return f;  // every product and partial sum is an integer < 2^53: no rounding anywhere
}

RD_HD uint32_t rd_exact_bit_f64(rd_dd2 n, rd_dd2 np) {
    const double p1 = n.y * np.x, p2 = n.x * np.y;
    const double d = p1 - p2;
    const double guard = 4.440892098500626e-16 * (__builtin_fabs(p1) + __builtin_fabs(p2));  // 2^-51
    if (d > guard) return 0u;
    if (d < -guard) return 1u;
    rd_i2 ni = {(int64_t)n.x, (int64_t)n.y}, npi = {(int64_t)np.x, (int64_t)np.y};
    return rd_exact_bit(ni, npi);
}

// Exact bits of the 8-sample group starting at t0 (t0 % 8 == 0) from ten preloaded dwords that
// hold samples t0-10 .. t0+9 (dword d = samples t0-10+2d, t0-9+2d as I,Q,I,Q bytes).  Used by
// k_fixup so that all of a group's input is fetched with independent, aligned 4-byte loads.
// `first_valid`: samples with index < first_valid (zero history) read as y = 0.
RD_HD uint32_t rd_exact_group_dw(const uint32_t *dw, long t0, int count, long first_valid) {
    rd_dd2 y[17];  // rotated integer samples t0-10 .. t0+6 as doubles (t0 % 4 == 0: static phases)
    // window samples i < zh lie before the first readable sample (one 64-bit subtraction, then
    // 32-bit compares against constants)
    const long zh64 = first_valid - (t0 - 10);
    const int zh = zh64 <= 0 ? 0 : zh64 >= 17 ? 17 : (int)zh64;
#pragma unroll
    for (int i = 0; i < 17; i++) {
        const uint32_t d = dw[i >> 1] >> (16 * (i & 1));
        double a = (double)(5 * (int)(d & 0xFF) - 637), b = (double)(5 * (int)((d >> 8) & 0xFF) - 637);
        if (i < zh) { a = 0.0; b = 0.0; }  // zero history: y = 0 (5k-637 is never 0 otherwise)
        const int ph = (i + 2) & 3;  // (t0 - 10 + i) mod 4
        y[i].x = ph == 0 ? a : ph == 1 ? -b : ph == 2 ? -a : b;
        y[i].y = ph == 0 ? b : ph == 1 ? a : ph == 2 ? -b : -a;
    }
    rd_dd2 prev = rd_fir_f64_exact(&y[0]);  // f[t0-1] uses samples t0-10 .. t0-2
    uint32_t word = 0;
#pragma unroll
    for (int r = 0; r < RD_GROUP; r++) {
        const rd_dd2 cur = rd_fir_f64_exact(&y[r + 1]);  // f[t0+r] uses samples t0+r-9 .. t0+r-1
        if (r < count) word |= rd_exact_bit_f64(prev, cur) << r;
        prev = cur;
    }
    return word;
}

// Exact bits of samples [t0, t0+count), count <= 32, as a packed word (bit r = sample t0+r).
RD_HD uint32_t rd_exact_run(const rd_stream_view &v, long t0, int count) {
    rd_i2 y[9];
#pragma unroll
    for (int m = 0; m < 9; m++) y[m] = rd_sample_int(v, t0 - 10 + m);
    rd_i2 prev = rd_fir_int(y);  // f[t0-1]
    uint32_t word = 0;
    for (int r = 0; r < count; r++) {
#pragma unroll
        for (int m = 0; m < 8; m++) y[m] = y[m + 1];
        y[8] = rd_sample_int(v, t0 + r - 1);
        const rd_i2 cur = rd_fir_int(y);  // f[t0+r]
        word |= rd_exact_bit(prev, cur) << r;
        prev = cur;
    }
    return word;
}

// ---- float64 evaluation (values, not just signs) -------------------------------------
struct rd_d2 {
    double x, y;
};

// py:46-49: x * j^(n mod 4), exact
RD_HD rd_d2 rd_rot_f64(double a, double b, long n) {
    rd_d2 y;
    switch (n & 3) {
        case 0: y.x = a; y.y = b; break;
        case 1: y.x = -b; y.y = a; break;
        case 2: y.x = -a; y.y = -b; break;
        default: y.x = b; y.y = -a; break;
    }
    return y;
}

// py:26 lut[k] = (k - 127.4) / 127.6.  Evaluated here as (k - 127.4) * (1/127.6): at most 1 ulp
// from the reference's quotient, which only feeds values compared with a tolerance
// (filtered 1e-14, discriminated 1e-5 relative, RSSI/SNR 1e-3 dB) - never a sign decision,
// those use the exact integer path.  A float64 divide per sample is ~10x slower.
RD_HD double rd_lut(int k) { return ((double)k - 127.4) * (1.0 / 127.6); }

RD_HD rd_d2 rd_sample_f64(const rd_stream_view &v, long n) {
    rd_d2 y = {0.0, 0.0};
    if (n < v.valid_from) return y;
    const uint8_t *p = v.base + 2 * n;
    return rd_rot_f64(rd_lut(p[0]), rd_lut(p[1]), n);
}

// complex128 input (py:144-150): interleaved re,im doubles
struct rd_cplx_view {
    const double *base;  // sample 0
    long valid_from;
    long n;
};

RD_HD rd_d2 rd_sample_f64(const rd_cplx_view &v, long n) {
    rd_d2 y = {0.0, 0.0};
    if (n < v.valid_from) return y;
    return rd_rot_f64(v.base[2 * n], v.base[2 * n + 1], n);
}

// float64 fir9 output f[t] = sum_m c_m y[t-9+m], products summed in tap order m = 0..8
// (np.convolve's own order is unspecified; agreement is ~1 ulp).  Compiled with
// -ffp-contract=off so no fma contraction changes the rounding.  f[t] before the first
// readable sample is the zero state of py:133.
template <class View>
RD_HD rd_d2 rd_f_f64(const View &v, long t) {
    const double c[9] = {RD_C0, RD_C1, RD_C2, RD_C3, RD_C4, RD_C3, RD_C2, RD_C1, RD_C0};
    rd_d2 f = {0.0, 0.0};
    if (t < v.valid_from) return f;
#pragma unroll
    for (int m = 0; m < 9; m++) {
        const rd_d2 y = rd_sample_f64(v, t - 9 + m);
        f.x += c[m] * y.x;
        f.y += c[m] * y.y;
    }
    return f;
}

// py:80-90
RD_HD double rd_disc_f64(rd_d2 n, rd_d2 np) {
    return (n.y * np.x - n.x * np.y) / (n.x * n.x + n.y * n.y + 1e-10);
}

RD_HD uint32_t rd_signbit_f64(double d) { return (uint32_t)(__builtin_bit_cast(uint64_t, d) >> 63); }
