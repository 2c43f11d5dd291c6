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
// Kernel: VALU fp32 (8 T flops per output: 113 GFLOP for one second of 51 channels - compute-
// bound, the 81 MB of HBM traffic would take 10 us).  A workgroup of up to 16 waves stages the
// 63 D + T input samples of 64 output times in LDS as float2, with one pad slot per D samples so
// that the lanes' stride is 2(D+1) dwords (D = 100: 202 = 10 mod 32, a 64-bit read per lane
// spreads over all banks).  Each wave takes RD_CHAN_CPW = 4 channels with lane = output time; its
// taps come through a wave-private LDS buffer (16 taps x 4 channels per chunk, fetched one chunk
// ahead with one coalesced 16-byte load per lane) and are read back as broadcasts, so per tap one
// ds_read_b64 (samples) + two ds_read_b128 (taps) feed 16 fmas: 64 TFLOP/s for one second of
// capture (0.88 ms), limited by the CU's one LDS pipe.  Measured and not kept: taps through scalar loads (the compiler keeps two loads
// in flight per wave: 46 TFLOP/s); samples as half2 of the exact integers 10 k - 1274 with 1-4
// output times per lane (fewer LDS bytes per fma, but the conversions are 4-cycle-class
// instructions and the bigger tiles cost occupancy: 30-42 TFLOP/s); a first version of the same
// contraction on v_mfma_f32_32x32x2_f32 (real 128 x 1024 x N GEMM, A streamed through LDS, bytes
// converted on fetch; bit-identical results): 67 TFLOP/s (0.84 ms), 4 % better than this kernel
// although the pipe itself sustains 155 TFLOP/s (profiles/r01_ubench_mfma_f32_rate.txt) - half of
// its time is outside the MFMA loop (a fifth round of workgroups for 2112 tiles on 512 slots,
// staging, the epilogue's phasors) and was not reworked in this round.  MFMA is the right unit for this contraction (fp32 for 2-3x,
// bf16 with the taps split into three bf16 terms - the inputs are 8-bit integers, exact in
// bf16 - for the factor beyond); that kernel is the next step for this row.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <hip/hip_runtime.h>

#include "../../include/rtldavis_hip.h"

extern int rd_fail_msg(int code, const char *fmt, ...);  // rd_api.hip: sets rd_last_error
extern int rd_ensure_device_public(void);

#define RD_CHAN_TT 64      // output times per workgroup (one per lane)
#define RD_CHAN_CPW 4      // channels per wave (register tile)
#define RD_CHAN_KC 16      // taps per staged chunk: KC * CPW float2 = 512 B = one 16-byte load per lane 0..31
#define RD_CHAN_MAX_WAVES 16

struct rd_chan {
    rd_chan_config cfg;
    int n_ch_pad = 0;              // channels rounded up to RD_CHAN_CPW
    int t_pad = 0;                 // taps rounded up to RD_CHAN_KC (zero taps appended)
    std::vector<float> h_taps;     // [n_ch_pad / CPW][t_pad][CPW][2]  g_c[k] (re, im), c = group * CPW + q
    std::vector<int64_t> shifts;   // Hz, reduced mod out_rate
    float *d_taps = nullptr;
    int64_t *d_shifts = nullptr;
    uint8_t *d_wide = nullptr;     // resident capture, 2 bytes per sample
    size_t wide_cap = 0, wide_n = 0;
    bool dev_ready = false;
};

__device__ __forceinline__ int rd_chan_lds_index(int n_rel, int D) { return n_rel + n_rel / D; }

// x mod m for integer-valued 0 <= x < 2^53, 1 <= m < 2^26 (exact: one fma, one correction step)
__device__ __forceinline__ double rd_chan_mod(double x, double m) {
    double r = fma(-floor(x / m), m, x);
    if (r < 0.0) r += m;
    if (r >= m) r -= m;
    return r;
}

// T is the padded tap count (multiple of RD_CHAN_KC, zero taps at the end); taps layout
// [group][k][q] float2 with group = channel / CPW, q = channel % CPW.
__global__ __launch_bounds__(64 * RD_CHAN_MAX_WAVES) void k_channelize(const uint8_t *__restrict__ wide, long n_wide,
                                                                      const float2 *__restrict__ taps,
                                                                      const int64_t *shifts, int T, int D, int n_ch,
                                                                      int n_groups, long out_rate, float gain,
                                                                      long n_out, uint8_t *out, size_t out_stride,
                                                                      int xs_slots) {
    extern __shared__ float2 lds[];
    float2 *xs = lds;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float2 *tbuf = lds + xs_slots + wave * (2 * RD_CHAN_KC * RD_CHAN_CPW);  // wave-private, double-buffered
    const long t0 = (long)blockIdx.x * RD_CHAN_TT;
    // stage samples n_base .. n_base + span - 1, n_base = D t0 - (T - 1)
    const long n_base = (long)D * t0 - (T - 1);
    const int span = (RD_CHAN_TT - 1) * D + T;
    for (int i = threadIdx.x; i < span; i += blockDim.x) {
        const long n = n_base + i;
        float2 v = {0.0f, 0.0f};  // zero history before the capture (and zero padding after it)
        if (n >= 0 && n < n_wide) {
            const uint16_t iq = *(const uint16_t *)(wide + 2 * n);
            v.x = ((float)(iq & 0xFF) - 127.4f) * (1.0f / 127.6f);
            v.y = ((float)(iq >> 8) - 127.4f) * (1.0f / 127.6f);
        }
        xs[rd_chan_lds_index(i, D)] = v;
    }
    __syncthreads();
    const long t = t0 + lane;
    const int nwaves = blockDim.x >> 6;
    for (int grp = blockIdx.y * nwaves + wave; grp < n_groups; grp += gridDim.y * nwaves) {  // wave-uniform
        float ar[RD_CHAN_CPW], ai[RD_CHAN_CPW];
#pragma unroll
        for (int q = 0; q < RD_CHAN_CPW; q++) { ar[q] = 0.0f; ai[q] = 0.0f; }
        // chunk c = taps k in [c KC, (c+1) KC) of this group's 4 channels: 64 float2 = 32 float4
        const float4 *gsrc = (const float4 *)(taps + (size_t)grp * T * RD_CHAN_CPW);
        const int n_chunks = T / RD_CHAN_KC;
        float4 pre = {0.0f, 0.0f, 0.0f, 0.0f};
        if (lane < 32) pre = gsrc[lane];
        // tap k reads relative sample j = T-1-k of time 0, D n + j of time n: padded index
        // (D+1) n + j + j / D; j walks down, its quotient and remainder by D are kept in scalars
        const float2 *xl = xs + (D + 1) * lane;
        int j = T - 1, jq = (T - 1) / D, jr = (T - 1) % D;
        for (int c = 0; c < n_chunks; c++) {
            float2 *tb = tbuf + (c & 1) * (RD_CHAN_KC * RD_CHAN_CPW);
            if (lane < 32) ((float4 *)tb)[lane] = pre;
            if (c + 1 < n_chunks && lane < 32) pre = gsrc[(size_t)(c + 1) * 32 + lane];
#pragma unroll
            for (int kk = 0; kk < RD_CHAN_KC; kk++) {
                const float2 x = xl[j + jq];
                const float4 g01 = ((const float4 *)tb)[kk * 2], g23 = ((const float4 *)tb)[kk * 2 + 1];  // broadcasts
                const float gr[4] = {g01.x, g01.z, g23.x, g23.z}, gi[4] = {g01.y, g01.w, g23.y, g23.w};
#pragma unroll
                for (int q = 0; q < RD_CHAN_CPW; q++) {
                    ar[q] = __builtin_fmaf(gr[q], x.x, ar[q]);
                    ar[q] = __builtin_fmaf(-gi[q], x.y, ar[q]);
                    ai[q] = __builtin_fmaf(gr[q], x.y, ai[q]);
                    ai[q] = __builtin_fmaf(gi[q], x.x, ai[q]);
                }
                j--;
                if (jr == 0) { jr = D - 1; jq--; } else jr--;
            }
        }
        if (t < n_out) {
#pragma unroll
            for (int q = 0; q < RD_CHAN_CPW; q++) {
                const int ch = grp * RD_CHAN_CPW + q;
                if (ch >= n_ch) break;
                // phase = -2 pi frac(shift t / Fo), the remainders exact in float64: shifts[] holds
                // shift mod Fo in [0, Fo), Fo < 2^26, so (shift mod Fo)(t mod Fo) < 2^52
                const double fo = (double)out_rate;
                const double tm = rd_chan_mod((double)t, fo);
                const double rm = rd_chan_mod((double)shifts[ch] * tm, fo);
                float sn, cs;
                sincospif(-2.0f * (float)(rm / fo), &sn, &cs);
                const float zr = (ar[q] * cs - ai[q] * sn) * gain, zi = (ar[q] * sn + ai[q] * cs) * gain;
                const float qr = fminf(fmaxf(rintf(zr * 127.6f + 127.4f), 0.0f), 255.0f);
                const float qi = fminf(fmaxf(rintf(zi * 127.6f + 127.4f), 0.0f), 255.0f);
                const uint16_t o = (uint16_t)((uint32_t)qr | ((uint32_t)qi << 8));
                *(uint16_t *)(out + (size_t)ch * out_stride + 2 * t) = o;
            }
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

extern "C" int rd_chan_create(const rd_chan_config *cfg, const double *taps, const int64_t *shift_hz, rd_chan **out) {
    if (!cfg || !taps || !shift_hz || !out) return rd_fail_msg(RD_ERR_ARG, "null argument");
    if (cfg->decim < 1 || cfg->decim > 4096 || cfg->n_taps < 1 || cfg->n_taps > 8192 || cfg->n_channels < 1 ||
        cfg->n_channels > 4096 || cfg->out_rate < 1 || cfg->out_rate >= (1 << 26) || !(cfg->gain > 0.0))
        return rd_fail_msg(RD_ERR_ARG, "channelizer config out of range");
    const int t_pad = (cfg->n_taps + RD_CHAN_KC - 1) / RD_CHAN_KC * RD_CHAN_KC;
    const size_t lds = ((size_t)(RD_CHAN_TT - 1) * cfg->decim + t_pad);
    if ((lds + lds / cfg->decim + 2 + RD_CHAN_MAX_WAVES * 2 * RD_CHAN_KC * RD_CHAN_CPW) * sizeof(float2) > 160 * 1024)
        return rd_fail_msg(RD_ERR_ARG, "decim x 63 + n_taps samples do not fit the 160 KiB LDS");
    rd_chan *h = new rd_chan();
    h->cfg = *cfg;
    const int T = cfg->n_taps;
    h->t_pad = t_pad;
    h->n_ch_pad = (cfg->n_channels + RD_CHAN_CPW - 1) / RD_CHAN_CPW * RD_CHAN_CPW;
    h->h_taps.assign((size_t)h->n_ch_pad * t_pad * 2, 0.0f);
    h->shifts.resize(cfg->n_channels);  // shift mod Fo in [0, Fo): all the output phasor needs
    for (int c = 0; c < cfg->n_channels; c++) h->shifts[c] = ((shift_hz[c] % cfg->out_rate) + cfg->out_rate) % cfg->out_rate;
    const double wide_rate = (double)cfg->out_rate * cfg->decim;
    for (int c = 0; c < cfg->n_channels; c++)
        for (int k = 0; k < T; k++) {
            // g_c[k] = h[k] e^{+j 2 pi shift k / Fw}; the phase through an exact integer remainder
            const __int128 prod = (__int128)shift_hz[c] * k;
            const long fw = (long)cfg->out_rate * cfg->decim;
            long r = (long)(prod % fw);
            if (r < 0) r += fw;
            const double ph = 2.0 * M_PI * ((double)r / wide_rate);
            const size_t at = ((((size_t)(c / RD_CHAN_CPW) * t_pad + k) * RD_CHAN_CPW) + c % RD_CHAN_CPW) * 2;
            h->h_taps[at] = (float)(taps[k] * cos(ph));
            h->h_taps[at + 1] = (float)(taps[k] * sin(ph));
        }
    *out = h;
    return RD_OK;
}

extern "C" void rd_chan_destroy(rd_chan *h) {
    if (!h) return;
    if (h->dev_ready) { hipFree(h->d_taps); hipFree(h->d_shifts); hipFree(h->d_wide); }
    delete h;
}

static int chan_alloc(rd_chan *h, size_t n_wide) {
    int rc = rd_ensure_device_public();
    if (rc) return rc;
    if (!h->dev_ready) {
        CHK(hipMalloc(&h->d_taps, h->h_taps.size() * sizeof(float)));
        CHK(hipMemcpy(h->d_taps, h->h_taps.data(), h->h_taps.size() * sizeof(float), hipMemcpyHostToDevice));
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
    const size_t span = (size_t)(RD_CHAN_TT - 1) * D + T;
    const int groups = h->n_ch_pad / RD_CHAN_CPW;
    const int waves = groups < RD_CHAN_MAX_WAVES ? groups : RD_CHAN_MAX_WAVES;
    const size_t xs_slots = (span + span / D + 2) & ~(size_t)1;  // float2 slots, 16-byte aligned end
    const size_t lds = (xs_slots + (size_t)waves * 2 * RD_CHAN_KC * RD_CHAN_CPW) * sizeof(float2);
    static bool attr_set = false;
    if (!attr_set) {
        CHK(hipFuncSetAttribute((const void *)k_channelize, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    const unsigned gx = (unsigned)((n_out + RD_CHAN_TT - 1) / RD_CHAN_TT);
    const unsigned gy = (unsigned)((groups + waves - 1) / waves);
    hipLaunchKernelGGL(k_channelize, dim3(gx, gy), dim3(64 * waves), lds, (hipStream_t)hip_stream, h->d_wide,
                       (long)h->wide_n, (const float2 *)h->d_taps, h->d_shifts, T, D, h->cfg.n_channels, groups,
                       (long)h->cfg.out_rate, (float)h->cfg.gain, (long)n_out, (uint8_t *)dst_dev, dst_stream_stride,
                       (int)xs_slots);
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
