// rd_channelizer.hip - wideband front end: one uint8 IQ capture at
// decim x 268.8 kSPS -> one 268.8 kSPS uint8 IQ stream per hop channel, written straight into a
// batch demodulator's resident input buffer.
//
// rtldavis has no channelizer (it retunes one narrow-band dongle per hop, runners/rtlsdr.py:51,72),
// so there is no reference implementation and parity is UNPINNED: the definition below is this
// repo's own, restated in float64 by oracle/channelizer_oracle.py, and the tests tie it to the
// reference through the packets the reference demodulator recovers from its output.
//
//   x[n]   = lut(I[n]) + j lut(Q[n]),  lut(k) = (k - 127.4) / 127.6        (dsp.py:20-39)
//   z_c[t] = sum_{k<T} h[k] x[D t - k] e^{-j 2 pi shift_c (D t - k) / Fw}    (x[n<0] = 0)
//          = e^{-j 2 pi frac(shift_c t / Fo)} sum_k g_c[k] x[D t - k],   g_c[k] = h[k] e^{+j 2 pi shift_c k / Fw}
//   out_c[t] = clip(rint(gain z 127.6 + 127.4), 0, 255) per component       (the synth's quantiser)
// with Fw the wideband rate, D the decimation, Fo = Fw / D; shift_c an integer number of Hz, so the
// output phasor's phase is an exact integer remainder.
//
// Kernel: f16 MFMA (v_mfma_f32_32x32x16_f16) - the one dense contraction in this repo.  As a real
// GEMM, C[m][n] = sum_kappa A[m][kappa] B[kappa][n] with
//   m     = 2 c + part  (part 0 = re, 1 = im of channel c; rows 2c, 2c+1 land in one lane's registers)
//   kappa = 2 i + comp  (comp 0 = I, 1 = Q of window sample i, ascending in memory; inside a lane's four samples the
//           fragment's elements are ordered I0 I1 Q0 Q1 | I2 I3 Q2 Q3, which is what one v_and + one v_perm per dword give)
//   A[2c][2i] = g_r, A[2c][2i+1] = -g_i, A[2c+1][2i] = g_i, A[2c+1][2i+1] = g_r    (g = g_c[T-i], zero outside 0..T-1)
//   B[2i + comp][n] = b_comp[D (t0 + n) - T + i] 2^-24
// The samples are bytes: read as an f16 bit pattern, a byte in the low half of a 16-bit lane IS the subnormal
// b 2^-24 (round 3; the matrix pipe takes subnormal inputs at face value and at full rate,
// profiles/r02_ubench_mfma_subnormal.txt), and lut(b) = (b - 127.4) / 127.6, so
// z = (sum - 127.4 (1 + j) sum_k g_c[k]) / 127.6 with the second term a per-channel constant (a short
// table for the first outputs of a capture, whose history is zero).  The taps, scaled by a power of two
// so that the largest sits just below 2^15, are split into TWO f16 terms (hi + lo: 22 bits; round 1 used
// three bf16 terms for 24), i.e. two MFMAs per tile and K step into the same fp32 accumulator: products
// are exact, sums are fp32, the tap error 2^-22 relative - two orders below the fp32 accumulation's.
// 8 T flops per output: 113 GFLOP for one second of 51 channels; HBM traffic is 81 MB.
// A workgroup = 4 waves = 128 output times x a group of 4 row blocks (64 channels); wave w owns row
// block w and the four 32-time blocks (4 accumulator tiles).  The 127 D + T + 8 input samples are staged
// once in LDS as they are, two bytes each - 26 KiB (round 2: f16 pairs, 52 KiB and three workgroups per CU) - so a B
// fragment is one aligned ds_read_b64 (D = 100: lane stride 50 dwords, conflict-free) and four v_perm.  A wave only
// needs its own row block's A fragments (2 terms x 16 bytes per lane and K step): they come straight from L2
// into registers two K steps ahead; the main loop has no barrier.
// Measured (one second of capture, 27 M samples -> 51 x 270 k): 0.25 ms = 4000x real time.
// On the way: fp32 VALU kernel 0.88 ms (64 TFLOP/s, bound by the CU's LDS pipe: every fma needed
// 2.5 bytes from LDS), fp32 MFMA 0.84 ms, bf16 MFMA with the B bytes converted per fragment in
// each wave 0.55 ms (one wave per SIMD cannot hide ~25 VALU instructions per three MFMAs), A
// through a shared LDS chunk two steps ahead instead of one 0.55 -> 0.46, samples pre-converted in
// LDS 0.36, 128 instead of 256 output times per workgroup 0.27, A per wave from L2 without LDS or
// barriers 0.25 (three bf16 tap terms); two f16 tap terms: see DESIGN.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <unistd.h>

#include <hip/hip_runtime.h>

#include "../../include/rtldavis_hip.h"

extern int rd_fail_msg(int code, const char *fmt, ...);  // rd_api.hip: sets rd_last_error
extern int rd_ensure_device_public(void);

#define RD_CHAN_TB 4                        // 32-time blocks per wave
#define RD_CHAN_TT (32 * RD_CHAN_TB)        // output times per workgroup
#define RD_CHAN_RBG 4                       // row blocks (32 rows = 16 channels) per workgroup: one per wave
#define RD_CHAN_KC 8                        // window samples per K step (taps are padded to a multiple)
#define RD_CHAN_TERMS 2                     // f16 digits per tap
#define RD_CHAN_Q_BYTES (RD_CHAN_TERMS * RD_CHAN_RBG * 64 * 16)  // A bytes per K step: terms x 4 row blocks x 64 lanes x 16 B
#ifndef RD_CHAN_NPF
#define RD_CHAN_NPF 2                       // A chunks in flight in registers
#endif
#ifndef RD_CHAN_MINWAVES
#define RD_CHAN_MINWAVES 4                  // waves per SIMD the register allocation aims at (26 KiB of LDS admit 6)
#endif
#define RD_CHAN_EARLY 64                    // outputs per channel with a partial-history DC term kept in a table
#define RD_CHAN_DCN (RD_CHAN_EARLY + 2)     // table entries per channel: the early DC terms, the steady one, and the
                                            // phasor of 32 output times e^{-j 2 pi frac(32 shift / Fo)} (cos, sin)

typedef float rd_f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 rd_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 rd_f16x2 __attribute__((ext_vector_type(2)));

struct rd_chan {
    rd_chan_config cfg;
    int n_groups = 0;              // groups of 64 channels
    int t_pad = 0;                 // taps rounded up to RD_CHAN_KC (zero taps appended)
    int n_early = 0;               // outputs whose window reaches before the capture: ceil((t_pad - 1) / D)
    float tap_unscale = 1.0f;      // 2^-s: the taps in h_amat are scaled by 2^s
    std::vector<uint16_t> h_amat;  // f16 A operand in fragment order [group][K step][term][row block][lane][8]
    std::vector<float> h_dc;       // [channel][RD_CHAN_DCN][2]: -127.4 (1+j) sum of the taps a given output sees; the 32-step phasor
    std::vector<int64_t> shifts;   // Hz, reduced mod out_rate
    uint16_t *d_amat = nullptr;
    float *d_dc = nullptr;
    int64_t *d_shifts = nullptr;
    uint8_t *d_wide = nullptr;     // resident capture, 2 bytes per sample
    size_t wide_cap = 0, wide_n = 0;
    bool dev_ready = false;
    int device = -1;               // the device the buffers live on
    pid_t pid = 0;                 // the process that allocated them
};

// x mod m for integer-valued 0 <= x < 2^53, 1 <= m < 2^26 (exact: one fma, one correction step)
__device__ __forceinline__ double rd_chan_mod(double x, double m) {
    double r = fma(-floor(x / m), m, x);
    if (r < 0.0) r += m;
    if (r >= m) r -= m;
    return r;
}

// T is the padded tap count (multiple of RD_CHAN_KC, zero taps appended), D a multiple of 4.
// Round 3: the window is staged as the RAW bytes - a byte in the low half of a 16-bit lane IS the f16 subnormal
// k 2^-24, which the matrix pipe takes at face value and at full rate (the demod kernel's operands, rd_mfma.h) - 2
// bytes per sample in LDS instead of an f16 pair's 4 (26 KiB per workgroup instead of 52), no conversion arithmetic in
// the staging, and the 127.4 offset of the LUT as a per-channel constant (-127.4 sum of the taps an output sees).  The
// window starts one sample early, at n0 = D t0 - T, a multiple of 8 samples: 16-byte loads, 16-byte LDS writes and
// 8-byte fragment reads are all aligned; the price is one more K step (the leading tap of it is zero).
__global__ __launch_bounds__(256, RD_CHAN_MINWAVES) void k_channelize(const uint8_t *__restrict__ wide, long n_wide,
                                                    const uint4 *__restrict__ amat, const float2 *__restrict__ dc,
                                                    const int64_t *shifts, int T, int D, int n_ch, int n_early,
                                                    long out_rate, float gain, float tap_unscale, long n_out,
                                                    uint8_t *out, size_t out_stride, int xs_bytes) {
    extern __shared__ uint8_t lds[];
    uint8_t *xs = lds;                            // window samples 0 .. span-1, two bytes each (I, Q)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // = row block
    const int r = lane & 31, h = lane >> 5;
    const long t0 = (long)blockIdx.x * RD_CHAN_TT;
    const int grp = blockIdx.y;
    // stage samples n0 .. n0 + span - 1; a sample before the capture (or after it) is the byte 0 = no contribution
    const long n0 = (long)D * t0 - T;
    const int span = (RD_CHAN_TT - 1) * D + T + RD_CHAN_KC;
    const int n_vec = (span + 7) / 8;
    for (int q0 = threadIdx.x; q0 < n_vec; q0 += 4 * blockDim.x) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int q = q0 + u * blockDim.x;
            const long n = n0 + 8L * q;
            v[u] = uint4{0u, 0u, 0u, 0u};
            if (q < n_vec) {
                if (n >= 0 && n + 8 <= n_wide) {
                    v[u] = *(const uint4 *)(wide + 2 * n);
                } else {
                    uint16_t e[8];
#pragma unroll
                    for (int w = 0; w < 8; w++) e[w] = (n + w >= 0 && n + w < n_wide) ? *(const uint16_t *)(wide + 2 * (n + w)) : (uint16_t)0;
                    v[u] = uint4{(uint32_t)e[0] | ((uint32_t)e[1] << 16), (uint32_t)e[2] | ((uint32_t)e[3] << 16),
                                 (uint32_t)e[4] | ((uint32_t)e[5] << 16), (uint32_t)e[6] | ((uint32_t)e[7] << 16)};
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int q = q0 + u * blockDim.x;
            if (q < n_vec) *(uint4 *)(xs + 16 * q) = v[u];
        }
    }
    const int n_chunks = T / RD_CHAN_KC + 1;
    const uint4 *asrc = amat + (size_t)grp * n_chunks * (RD_CHAN_Q_BYTES / 16);
    // A wave only ever needs ITS row block's A fragments (RD_CHAN_TERMS x 16 bytes per lane and K step):
    // they come straight from L2 into registers, NPF steps ahead - no LDS, no barrier in the loop.
    constexpr int NPF = RD_CHAN_NPF;
    const uint4 *amine = asrc + wave * 64 + lane;   // + (q * RD_CHAN_TERMS + term) * RD_CHAN_RBG * 64
    uint4 pre[NPF][RD_CHAN_TERMS];
#pragma unroll
    for (int s = 0; s < NPF; s++)
#pragma unroll
        for (int term = 0; term < RD_CHAN_TERMS; term++)
            pre[s][term] = amine[(size_t)((s < n_chunks ? s : n_chunks - 1) * RD_CHAN_TERMS + term) * (RD_CHAN_RBG * 64)];
    __syncthreads();  // the staged samples

    rd_f32x16 acc[RD_CHAN_TB];
#pragma unroll
    for (int b = 0; b < RD_CHAN_TB; b++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[b][e] = 0.0f;
    // B fragment of (time block tb, K step q): window samples 8q + 4h .. +3 of column 32 tb + r = eight bytes
    // (I0 Q0 I1 Q1 | I2 Q2 I3 Q3) -> eight 16-bit lanes
    const uint8_t *xl = xs + 2 * (D * r + 4 * h);
    for (int c0 = 0; c0 < n_chunks; c0 += NPF) {
#pragma unroll
        for (int s = 0; s < NPF; s++) {
            const int q = c0 + s;
            if (q >= n_chunks) break;  // uniform
            rd_f16x8 a[RD_CHAN_TERMS];
#pragma unroll
            for (int term = 0; term < RD_CHAN_TERMS; term++) a[term] = __builtin_bit_cast(rd_f16x8, pre[s][term]);
            {   // refill this slot with step q + NPF (the last steps refetch the final one: a static
                // number of loads in flight)
                const int qn = (q + NPF < n_chunks) ? q + NPF : n_chunks - 1;
#pragma unroll
                for (int term = 0; term < RD_CHAN_TERMS; term++)
                    pre[s][term] = amine[(size_t)(qn * RD_CHAN_TERMS + term) * (RD_CHAN_RBG * 64)];
            }
            uint2 raw[RD_CHAN_TB];
#pragma unroll
            for (int tb = 0; tb < RD_CHAN_TB; tb++) raw[tb] = *(const uint2 *)(xl + 2 * (D * 32 * tb + 8 * q));
#pragma unroll
            for (int tb = 0; tb < RD_CHAN_TB; tb++) {
                uint4 f;
                // element order (I0 I1 Q0 Q1 | I2 I3 Q2 Q3) - the A fragments are laid out to match: the even bytes of a
                // dword by one v_and, the odd ones by one v_perm (selector 0x0c = a zero byte), as in rd_mf_frag
                f.x = raw[tb].x & 0x00FF00FFu;
                f.y = __builtin_amdgcn_perm(0u, raw[tb].x, 0x0c030c01u);
                f.z = raw[tb].y & 0x00FF00FFu;
                f.w = __builtin_amdgcn_perm(0u, raw[tb].y, 0x0c030c01u);
                const rd_f16x8 bfrag = __builtin_bit_cast(rd_f16x8, f);
#pragma unroll
                for (int term = 0; term < RD_CHAN_TERMS; term++)
                    acc[tb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[term], bfrag, acc[tb], 0, 0, 0);
            }
        }
    }
    // epilogue: register e of tile tb holds row (e & 3) + 8 (e >> 2) + 4 h, column r; rows 2i, 2i+1 =
    // (re, im) of channel 16 (4 grp + wave) + i
    const float scale = gain * (1.0f / 127.6f);
    // The output phasor e^{-j 2 pi frac(shift t / Fo)}: the exact remainder once per channel and lane, for the lane's
    // first time block (one float64 product, a quotient by multiplication, the hardware sine and cosine); the three
    // blocks behind it are 32 output times further on each - a rotation by the channel's constant from the table.
    // (Round 3, from the counters: the kernel's vector work - two float64 divisions and a sincospif per OUTPUT - took
    // as long as its MFMAs, and the two do not overlap.)
    const double fo = (double)out_rate, inv_fo = 1.0 / fo;
    // (t0 + r) mod Fo once per lane: a float estimate of the quotient is off by one at most
    long tm = t0 + r;
    {
        const long qe = (long)floorf((float)tm * (float)inv_fo);
        tm -= qe * out_rate;
        if (tm < 0) tm += out_rate;
        if (tm >= out_rate) tm -= out_rate;
    }
#pragma unroll
    for (int e = 0; e < 16; e += 2) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
        const int ch = 16 * (RD_CHAN_RBG * grp + wave) + (row >> 1);
        if (ch >= n_ch) continue;
        // frac(shift t / Fo) exactly: shift, tm < Fo < 2^26, the product is exact in float64; the quotient from a
        // multiplication by 1 / Fo is off by one at most
        const double x = (double)shifts[ch] * (double)tm;
        double rm = __builtin_fma(-floor(x * inv_fo), fo, x);
        if (rm < 0.0) rm += fo;
        if (rm >= fo) rm -= fo;
        // v_sin_f32 / v_cos_f32 take their argument in revolutions
        const float turns = -(float)(rm * inv_fo);
        float sn = __builtin_amdgcn_sinf(turns), cs = __builtin_amdgcn_cosf(turns);
        const float2 *dcc = dc + (size_t)ch * RD_CHAN_DCN;
        const float2 rot = dcc[RD_CHAN_EARLY + 1];
        const float ci = rot.x, si = rot.y;
#pragma unroll
        for (int tb = 0; tb < RD_CHAN_TB; tb++) {
            const long t = t0 + 32 * tb + r;
            if (t < n_out) {
                const float2 d0 = dcc[t < n_early ? (int)t : RD_CHAN_EARLY];
                const float re = __builtin_fmaf(acc[tb][e], tap_unscale, d0.x), im = __builtin_fmaf(acc[tb][e + 1], tap_unscale, d0.y);
                const float zr = (re * cs - im * sn) * scale, zi = (re * sn + im * cs) * scale;
                const float qr = fminf(fmaxf(rintf(zr * 127.6f + 127.4f), 0.0f), 255.0f);
                const float qi = fminf(fmaxf(rintf(zi * 127.6f + 127.4f), 0.0f), 255.0f);
                *(uint16_t *)(out + (size_t)ch * out_stride + 2 * t) = (uint16_t)((uint32_t)qr | ((uint32_t)qi << 8));
            }
            const float c2 = cs * ci - sn * si, s2 = sn * ci + cs * si;
            cs = c2; sn = s2;
        }
    }
}

// ------------------------------------------------------------------------------------------
// C ABI (include/rtldavis_hip.h)
// ------------------------------------------------------------------------------------------
#define CHK(x)                                                                                              \
    do {                                                                                                    \
        hipError_t e_ = (x);                                                                                \
        if (e_ != hipSuccess) return rd_fail_msg(RD_ERR_DEVICE, "%s: %s", #x, hipGetErrorString(e_));      \
    } while (0)

static uint16_t f16_rn(double v) {  // round to nearest even f16 (|v| < 65504)
    const _Float16 h = (_Float16)v;
    uint16_t u;
    memcpy(&u, &h, 2);
    return u;
}
static double f16_val(uint16_t b) {
    _Float16 h;
    memcpy(&h, &b, 2);
    return (double)h;
}

extern "C" int rd_chan_create(const rd_chan_config *cfg, const double *taps, const int64_t *shift_hz, rd_chan **out) {
    if (!cfg || !taps || !shift_hz || !out) return rd_fail_msg(RD_ERR_ARG, "null argument");
    if (cfg->decim < 4 || cfg->decim > 4096 || cfg->decim % 4 || cfg->n_taps < 1 || cfg->n_taps > 8192 ||
        cfg->n_channels < 1 || cfg->n_channels > 4096 || cfg->out_rate < 1 || cfg->out_rate >= (1 << 26) ||
        !(cfg->gain > 0.0))
        return rd_fail_msg(RD_ERR_ARG, "channelizer config out of range (decim: a multiple of 4)");
    const int t_pad = (cfg->n_taps + RD_CHAN_KC - 1) / RD_CHAN_KC * RD_CHAN_KC;
    const size_t span = (size_t)(RD_CHAN_TT - 1) * cfg->decim + t_pad + RD_CHAN_KC;
    if (2 * span + 16 > 160 * 1024)
        return rd_fail_msg(RD_ERR_ARG, "decim x 255 + n_taps samples do not fit the 160 KiB LDS");
    const int n_early = (t_pad - 1 + cfg->decim - 1) / cfg->decim;
    if (n_early > RD_CHAN_EARLY) return rd_fail_msg(RD_ERR_ARG, "n_taps / decim too large");
    rd_chan *h = new rd_chan();
    h->cfg = *cfg;
    const int T = cfg->n_taps;
    h->t_pad = t_pad;
    h->n_early = n_early;
    h->n_groups = (cfg->n_channels + 16 * RD_CHAN_RBG - 1) / (16 * RD_CHAN_RBG);
    const int n_q = t_pad / 8 + 1;  // the window starts one sample early (aligned): one more K step
    h->h_amat.assign((size_t)h->n_groups * n_q * (RD_CHAN_Q_BYTES / 2), 0);
    h->h_dc.assign((size_t)cfg->n_channels * RD_CHAN_DCN * 2, 0.0f);
    h->shifts.resize(cfg->n_channels);  // shift mod Fo in [0, Fo): all the output phasor needs
    for (int c = 0; c < cfg->n_channels; c++) h->shifts[c] = ((shift_hz[c] % cfg->out_rate) + cfg->out_rate) % cfg->out_rate;
    const double wide_rate = (double)cfg->out_rate * cfg->decim;
    // every |g_c[k]| <= max |h[k]|: scale the taps by the power of two that brings that just below 2^15 (f16: 11
    // significant bits from 2^-14 up), the kernel multiplies the sums back
    double hmax = 0.0;
    for (int k = 0; k < T; k++) hmax = std::max(hmax, std::fabs(taps[k]));
    int sexp = 0;
    if (hmax > 0.0) sexp = 14 - (int)std::ceil(std::log2(hmax));
    if (sexp > 60) sexp = 60;
    if (sexp < -60) sexp = -60;
    const double tap_scale = std::ldexp(1.0, sexp);
    h->tap_unscale = (float)std::ldexp(1.0, -sexp + 24);  // (+24: the samples enter as k 2^-24)
    std::vector<double> gr(t_pad), gi(t_pad);
    for (int c = 0; c < cfg->n_channels; c++) {
        for (int k = 0; k < t_pad; k++) {
            gr[k] = gi[k] = 0.0;
            if (k >= T) continue;
            // g_c[k] = h[k] e^{+j 2 pi shift k / Fw}; the phase through an exact integer remainder
            const __int128 prod = (__int128)shift_hz[c] * k;
            const long fw = (long)cfg->out_rate * cfg->decim;
            long r = (long)(prod % fw);
            if (r < 0) r += fw;
            const double ph = 2.0 * M_PI * ((double)r / wide_rate);
            gr[k] = (double)(float)(taps[k] * cos(ph));  // the fp32 taps are the definition's taps on the device
            gi[k] = (double)(float)(taps[k] * sin(ph));
        }
        // DC term: output t sees taps k <= D t (zero history before)
        double sr = 0.0, si = 0.0;
        int kdone = 0;
        for (int t = 0; t <= RD_CHAN_EARLY; t++) {
            const long kmax = t < RD_CHAN_EARLY ? (long)cfg->decim * t : (long)t_pad - 1;
            for (; kdone < t_pad && kdone <= kmax; kdone++) { sr += gr[kdone]; si += gi[kdone]; }
            // lut(b) = (b - 127.4) / 127.6 and the kernel sums g b: the constant is -127.4 (1 + j)(sr + j si)
            h->h_dc[((size_t)c * RD_CHAN_DCN + t) * 2] = (float)(-127.4 * (sr - si));
            h->h_dc[((size_t)c * RD_CHAN_DCN + t) * 2 + 1] = (float)(-127.4 * (sr + si));
        }
        {   // the rotation that takes the output phasor 32 output times on (exact remainder, float64 sin / cos)
            const long inc = (long)(((__int128)h->shifts[c] * 32) % cfg->out_rate);
            const double ph = -2.0 * M_PI * ((double)inc / (double)cfg->out_rate);
            h->h_dc[((size_t)c * RD_CHAN_DCN + RD_CHAN_EARLY + 1) * 2] = (float)cos(ph);
            h->h_dc[((size_t)c * RD_CHAN_DCN + RD_CHAN_EARLY + 1) * 2 + 1] = (float)sin(ph);
        }
        // rows 2c (re) and 2c+1 (im); kappa = 2 i + comp, window sample i = t_pad - 1 - k
        const int grp = c / (16 * RD_CHAN_RBG), rb = (c / 16) % RD_CHAN_RBG, r0 = 2 * (c % 16);
        for (int i = 0; i < 8 * n_q; i++) {   // window sample i of a column <-> tap k = t_pad - i (the window starts at D t - t_pad)
            const int k = t_pad - i;
            if (k < 0 || k >= t_pad) continue;  // (zero taps: the entries stay 0)
            const int q = i / 8, hh = (i % 8) / 4, s4 = i % 4;  // K step, lane half, sample within the lane's four
            for (int part = 0; part < 2; part++)
                for (int comp = 0; comp < 2; comp++) {
                    const double a = part == 0 ? (comp == 0 ? gr[k] : -gi[k]) : (comp == 0 ? gi[k] : gr[k]);
                    const double as = a * tap_scale;
                    const uint16_t hi = f16_rn(as), lo = f16_rn(as - f16_val(hi));
                    const uint16_t term[RD_CHAN_TERMS] = {hi, lo};
                    const int lane = 32 * hh + r0 + part;
                    for (int tm = 0; tm < RD_CHAN_TERMS; tm++) {
                        // element position inside the fragment: (I0 I1 Q0 Q1 | I2 I3 Q2 Q3), see the kernel's B fragments
                        const size_t at = (((((size_t)grp * n_q + q) * RD_CHAN_TERMS + tm) * RD_CHAN_RBG + rb) * 64 + lane) * 8 + 4 * (s4 / 2) + 2 * comp + (s4 % 2);
                        h->h_amat[at] = term[tm];
                    }
                }
        }
    }
    *out = h;
    return RD_OK;
}

extern "C" void rd_chan_destroy(rd_chan *h) {
    if (!h) return;
    // device memory belongs to the process that allocated it: a forked copy only drops its host state
    if (h->dev_ready && h->pid == getpid()) {
        if (h->device >= 0) hipSetDevice(h->device);
        hipFree(h->d_amat); hipFree(h->d_dc); hipFree(h->d_shifts); hipFree(h->d_wide);
    }
    delete h;
}

static int chan_alloc(rd_chan *h, size_t n_wide) {
    int rc = rd_ensure_device_public();
    if (rc) return rc;
    if (h->dev_ready && h->device >= 0) CHK(hipSetDevice(h->device));
    if (!h->dev_ready) {
        CHK(hipGetDevice(&h->device));
        h->pid = getpid();
        CHK(hipMalloc(&h->d_amat, h->h_amat.size() * sizeof(uint16_t)));
        CHK(hipMemcpy(h->d_amat, h->h_amat.data(), h->h_amat.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        CHK(hipMalloc(&h->d_dc, h->h_dc.size() * sizeof(float)));
        CHK(hipMemcpy(h->d_dc, h->h_dc.data(), h->h_dc.size() * sizeof(float), hipMemcpyHostToDevice));
        CHK(hipMalloc(&h->d_shifts, h->shifts.size() * sizeof(int64_t)));
        CHK(hipMemcpy(h->d_shifts, h->shifts.data(), h->shifts.size() * sizeof(int64_t), hipMemcpyHostToDevice));
        h->dev_ready = true;
    }
    if (n_wide > h->wide_cap) {
        if (h->d_wide) hipFree(h->d_wide);
        h->d_wide = nullptr;
        CHK(hipMalloc(&h->d_wide, 2 * n_wide + 16));
        h->wide_cap = n_wide;
    }
    return RD_OK;
}

extern "C" int rd_chan_input_ptr(rd_chan *h, size_t n_wide_samples, void **dev_ptr) {
    if (!h || !dev_ptr) return rd_fail_msg(RD_ERR_ARG, "null argument");
    int rc = chan_alloc(h, n_wide_samples);
    if (rc) return rc;
    h->wide_n = n_wide_samples;
    *dev_ptr = h->d_wide;
    return RD_OK;
}

extern "C" int rd_chan_upload(rd_chan *h, const uint8_t *wide_iq, size_t nbytes) {
    if (!h || !wide_iq) return rd_fail_msg(RD_ERR_ARG, "null argument");
    if (nbytes % 2) return rd_fail_msg(RD_ERR_ARG, "Incompatible array sizes: %zu bytes is not a whole number of IQ pairs", nbytes);
    int rc = chan_alloc(h, nbytes / 2);
    if (rc) return rc;
    CHK(hipMemcpy(h->d_wide, wide_iq, nbytes, hipMemcpyHostToDevice));
    h->wide_n = nbytes / 2;
    return RD_OK;
}

extern "C" int rd_chan_run(rd_chan *h, size_t n_out, void *dst_dev, size_t dst_stream_stride, void *hip_stream) {
    if (!h || !dst_dev) return rd_fail_msg(RD_ERR_ARG, "null argument");
    if (!h->dev_ready || !h->d_wide) return rd_fail_msg(RD_ERR_STATE, "no capture resident: rd_chan_upload first");
    if (n_out == 0) return RD_OK;
    if (n_out > h->wide_n / (size_t)h->cfg.decim || n_out > 0x7FFFFFFFull)
        return rd_fail_msg(RD_ERR_ARG, "n_out exceeds capture length / decim");
    if (dst_stream_stride < 2 * n_out || (dst_stream_stride & 1))
        return rd_fail_msg(RD_ERR_ARG, "destination stride too small for n_out samples");
    const int T = h->t_pad, D = h->cfg.decim;
    const size_t span = (size_t)(RD_CHAN_TT - 1) * D + T + RD_CHAN_KC;
    const size_t xs_bytes = (2 * span + 15 + 16) & ~(size_t)15;
    const size_t lds = xs_bytes;
    CHK(hipFuncSetAttribute((const void *)k_channelize, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const unsigned gx = (unsigned)((n_out + RD_CHAN_TT - 1) / RD_CHAN_TT);
    hipLaunchKernelGGL(k_channelize, dim3(gx, (unsigned)h->n_groups), dim3(256), lds, (hipStream_t)hip_stream, h->d_wide,
                       (long)h->wide_n, (const uint4 *)h->d_amat, (const float2 *)h->d_dc, h->d_shifts, T, D,
                       h->cfg.n_channels, h->n_early, (long)h->cfg.out_rate, (float)h->cfg.gain, h->tap_unscale, (long)n_out,
                       (uint8_t *)dst_dev, dst_stream_stride, (int)xs_bytes);
    CHK(hipGetLastError());
    return RD_OK;
}

extern "C" int rd_chan_run_host(rd_chan *h, size_t n_out, uint8_t *out_host, size_t nbytes) {
    if (!h || !out_host) return rd_fail_msg(RD_ERR_ARG, "null argument");
    const size_t need = (size_t)h->cfg.n_channels * n_out * 2;
    if (nbytes != need) return rd_fail_msg(RD_ERR_ARG, "Incompatible array sizes: got %zu bytes, expected %zu", nbytes, need);
    if (n_out == 0) return RD_OK;
    uint8_t *d = nullptr;
    CHK(hipMalloc(&d, need));
    int rc = rd_chan_run(h, n_out, d, 2 * n_out, nullptr);
    if (rc == RD_OK) {
        hipError_t e = hipMemcpy(out_host, d, need, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = rd_fail_msg(RD_ERR_DEVICE, "hipMemcpy: %s", hipGetErrorString(e));
    }
    hipFree(d);
    return rc;
}
