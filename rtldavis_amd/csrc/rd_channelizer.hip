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
// bound, the 81 MB of HBM traffic would take 10 us).  A workgroup stages the (TT-1) D + T input
// samples of TT = 64 output times in LDS as float2, with one pad slot per D samples so that the
// lanes' stride is 2(D+1) dwords (D = 100: 202 = 10 mod 32, a 64-bit read per lane spreads over all
// banks).  Each wave takes RD_CHAN_CPW = 4 channels with lane = output time: per tap one
// ds_read_b64 feeds 16 fmas, the four complex taps arrive through scalar loads (wave-uniform),
// the tap loop unrolled by 8 so that those loads are in flight together.  Measured: 1.9 ms for
// one second of capture (27 M samples -> 51 x 270 k), 30 TFLOP/s; a bf16-MFMA formulation (inputs
// are 8-bit integers, taps split into two bf16 terms) is the way to the next 10x and is not built.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <hip/hip_runtime.h>

#include "../../include/rtldavis_hip.h"

extern int rd_fail_msg(int code, const char *fmt, ...);  // rd_api.hip: sets rd_last_error
extern int rd_ensure_device_public(void);

#define RD_CHAN_TT 64   // output times per workgroup (one per lane)
#define RD_CHAN_CPW 4   // channels per wave and pass
#define RD_CHAN_WAVES 4

struct rd_chan {
    rd_chan_config cfg;
    int n_ch_pad = 0;              // channels rounded up to RD_CHAN_CPW
    std::vector<float> h_taps;     // [n_ch_pad][T][2]  g_c[k] (re, im)
    std::vector<int64_t> shifts;   // Hz
    float *d_taps = nullptr;
    int64_t *d_shifts = nullptr;
    uint8_t *d_wide = nullptr;     // resident capture, 2 bytes per sample
    size_t wide_cap = 0, wide_n = 0;
    bool dev_ready = false;
};

__device__ __forceinline__ int rd_chan_lds_index(int n_rel, int D) { return n_rel + n_rel / D; }

__global__ __launch_bounds__(64 * RD_CHAN_WAVES) void k_channelize(const uint8_t *__restrict__ wide, long n_wide,
                                                                  const float2 *__restrict__ taps, const int64_t *shifts,
                                                                  int T, int D, int n_ch, int n_ch_pad, long out_rate,
                                                                  float gain, long n_out, uint8_t *out,
                                                                  size_t out_stride) {
    extern __shared__ float2 xs[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long t0 = (long)blockIdx.x * RD_CHAN_TT;
    // stage samples n_base .. n_base + span - 1, n_base = D t0 - (T - 1)
    const long n_base = (long)D * t0 - (T - 1);
    const int span = (RD_CHAN_TT - 1) * D + T;
    for (int i = threadIdx.x; i < span; i += blockDim.x) {
        const long n = n_base + i;
        float2 v = {0.0f, 0.0f};
        if (n >= 0 && n < n_wide) {
            const uint16_t iq = *(const uint16_t *)(wide + 2 * n);
            v.x = ((float)(iq & 0xFF) - 127.4f) * (1.0f / 127.6f);
            v.y = ((float)(iq >> 8) - 127.4f) * (1.0f / 127.6f);
        }
        xs[rd_chan_lds_index(i, D)] = v;
    }
    __syncthreads();
    const long t = t0 + lane;
    // lane's newest sample (k = 0) sits at relative index D lane + T - 1
    for (int c0 = (blockIdx.y * RD_CHAN_WAVES + wave) * RD_CHAN_CPW; c0 < n_ch_pad;
         c0 += gridDim.y * RD_CHAN_WAVES * RD_CHAN_CPW) {
        float ar[RD_CHAN_CPW], ai[RD_CHAN_CPW];
#pragma unroll
        for (int j = 0; j < RD_CHAN_CPW; j++) { ar[j] = 0.0f; ai[j] = 0.0f; }
        const float2 *g = taps + (size_t)c0 * T;
        // tap k reads relative sample j = T-1-k of lane 0, D lane + j of this lane: padded index
        // (D+1) lane + j + j / D.  Walk j downwards in segments of constant j / D (no division in
        // the loop).
        const float2 *xl = xs + (D + 1) * lane;
        for (int jq = (T - 1) / D; jq >= 0; jq--) {
            const int j_hi = min(T - 1, jq * D + D - 1), j_lo = jq * D;
#pragma unroll 8
            for (int j = j_hi; j >= j_lo; j--) {
                const int k = T - 1 - j;
                const float2 x = xl[j + jq];
#pragma unroll
                for (int q = 0; q < RD_CHAN_CPW; q++) {
                    const float2 gk = g[(size_t)q * T + k];  // wave-uniform: scalar load
                    ar[q] = __builtin_fmaf(gk.x, x.x, ar[q]);
                    ar[q] = __builtin_fmaf(-gk.y, x.y, ar[q]);
                    ai[q] = __builtin_fmaf(gk.x, x.y, ai[q]);
                    ai[q] = __builtin_fmaf(gk.y, x.x, ai[q]);
                }
            }
        }
        if (t < n_out) {
#pragma unroll
            for (int j = 0; j < RD_CHAN_CPW; j++) {
                const int c = c0 + j;
                if (c >= n_ch) break;
                // phase = -2 pi frac(shift t / Fo), exact integer remainder
                const long sh = shifts[c];
                long r = (long)(((__int128)sh * t) % out_rate);
                if (r < 0) r += out_rate;
                float sn, cs;
                sincosf(-6.283185307179586f * ((float)r / (float)out_rate), &sn, &cs);
                const float zr = (ar[j] * cs - ai[j] * sn) * gain, zi = (ar[j] * sn + ai[j] * cs) * gain;
                const float qr = fminf(fmaxf(rintf(zr * 127.6f + 127.4f), 0.0f), 255.0f);
                const float qi = fminf(fmaxf(rintf(zi * 127.6f + 127.4f), 0.0f), 255.0f);
                const uint16_t o = (uint16_t)((uint32_t)qr | ((uint32_t)qi << 8));
                *(uint16_t *)(out + (size_t)c * out_stride + 2 * t) = o;
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
        cfg->n_channels > 4096 || cfg->out_rate < 1 || !(cfg->gain > 0.0))
        return rd_fail_msg(RD_ERR_ARG, "channelizer config out of range");
    const size_t lds = ((size_t)(RD_CHAN_TT - 1) * cfg->decim + cfg->n_taps);
    if ((lds + lds / cfg->decim + 1) * sizeof(float2) > 160 * 1024)
        return rd_fail_msg(RD_ERR_ARG, "decim x 63 + n_taps samples do not fit the 160 KiB LDS");
    rd_chan *h = new rd_chan();
    h->cfg = *cfg;
    const int T = cfg->n_taps;
    h->n_ch_pad = (cfg->n_channels + RD_CHAN_CPW - 1) / RD_CHAN_CPW * RD_CHAN_CPW;
    h->h_taps.assign((size_t)h->n_ch_pad * T * 2, 0.0f);
    h->shifts.assign(shift_hz, shift_hz + cfg->n_channels);
    const double wide_rate = (double)cfg->out_rate * cfg->decim;
    for (int c = 0; c < cfg->n_channels; c++)
        for (int k = 0; k < T; k++) {
            // g_c[k] = h[k] e^{+j 2 pi shift k / Fw}; the phase through an exact integer remainder
            const __int128 prod = (__int128)shift_hz[c] * k;
            const long fw = (long)cfg->out_rate * cfg->decim;
            long r = (long)(prod % fw);
            if (r < 0) r += fw;
            const double ph = 2.0 * M_PI * ((double)r / wide_rate);
            h->h_taps[((size_t)c * T + k) * 2] = (float)(taps[k] * cos(ph));
            h->h_taps[((size_t)c * T + k) * 2 + 1] = (float)(taps[k] * sin(ph));
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
    const int T = h->cfg.n_taps, D = h->cfg.decim;
    const size_t span = (size_t)(RD_CHAN_TT - 1) * D + T;
    const size_t lds = (span + span / D + 1) * sizeof(float2);
    static bool attr_set = false;
    if (!attr_set) {
        CHK(hipFuncSetAttribute((const void *)k_channelize, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    const unsigned gx = (unsigned)((n_out + RD_CHAN_TT - 1) / RD_CHAN_TT);
    const int groups = h->n_ch_pad / RD_CHAN_CPW;
    const unsigned gy = (unsigned)((groups + RD_CHAN_WAVES - 1) / RD_CHAN_WAVES);
    hipLaunchKernelGGL(k_channelize, dim3(gx, gy), dim3(64 * RD_CHAN_WAVES), lds, (hipStream_t)hip_stream, h->d_wide,
                       (long)h->wide_n, (const float2 *)h->d_taps, h->d_shifts, T, D, h->cfg.n_channels, h->n_ch_pad,
                       (long)h->cfg.out_rate, (float)h->cfg.gain, (long)n_out, (uint8_t *)dst_dev, dst_stream_stride);
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
