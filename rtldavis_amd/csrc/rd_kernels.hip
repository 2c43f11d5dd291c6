// rd_kernels.hip - gfx950 kernels of the rtldavis IQ -> bits -> packets path.
//
// Reference stages (py = /root/reference/src/rtldavis/dsp.py, go = /root/reference/dsp/dsp.go):
//   k_demod_bits   : LUT py:38-39 + rotate_fs4 py:46-49 + fir9 py:71-73 + discriminate
//                    numerator py:89 + quantize py:98 + bit pack (go:105-113), fused; fp32 with
//                    a rigorous guard band (rd_math.h) feeding
//   k_fixup        : exact integer re-evaluation of guard-band runs
//   k_search       : Demodulator._search py:171-188 (go:115-131) on packed bits, bit-parallel
//   k_slice        : Demodulator._slice py:190-246 (packing, RSSI/SNR windows)
//   k_disc/k_filt  : float64 discriminate / fir9 values for the state mirrors py:133-134
//   k_cplx_*       : the complex-input branch py:144-150 in float64
//   stage kernels  : one float64 kernel per reference stage function
//
// HBM-bound scan: no MFMA (there is no dense contraction).  The input is read once with
// 16-byte coalesced global->LDS loads; each lane then owns 32 consecutive samples so the
// 9-tap window lives in registers and one lane emits one packed 32-bit word.
#include "rd_internal.h"
#include "rd_math.h"

#define RD_WG 256
#define RD_WAVES (RD_WG / 64)
#define RD_LDS_WAVE (32 + RD_TILE_BYTES)  // 32 B halo + one 4 KiB tile, private to a wave

// Window bytes of one lane held in six dwordx4 registers: byte i of the window
// (sample t0-10 is byte 0) is byte 12+i of the 96-byte chunk starting 32 B before the run.
// The conversion is inline asm so that LLVM cannot rewrite "float(a) + float(b)" into an
// integer SDWA add followed by a conversion (it does: twice the instructions, all on the
// slow issue pipe).  Not volatile: it may be scheduled and CSE'd freely.
struct rd_reg_src {
    uint32_t q[24];
    __device__ __forceinline__ float f(int i) const {
        const int b = i + 12;
        const uint32_t d = q[b >> 2];
        float r;
        switch (b & 3) {
            case 0: asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(r) : "v"(d)); break;
            case 1: asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(r) : "v"(d)); break;
            case 2: asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(r) : "v"(d)); break;
            default: asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(r) : "v"(d)); break;
        }
        return r;
    }
};

// ------------------------------------------------------------------------------------------
// k_demod_bits: one wave = one 2048-sample tile per iteration, grid-stride over tiles.
// LDS is wave-private (no workgroup barrier anywhere): [halo 32 B][tile 4096 B].
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(RD_WG) void k_demod_bits(rd_layout lay, uint32_t tiles_per_stream,
                                                      uint32_t runs_per_stream, uint32_t *fix_list,
                                                      uint32_t fix_cap, uint32_t *counters) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[RD_WAVES][RD_LDS_WAVE];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    uint8_t *my = lds[wave];
    const uint64_t total = (uint64_t)lay.n_streams * tiles_per_stream;
    const uint64_t nwaves = (uint64_t)gridDim.x * RD_WAVES;
    for (uint64_t tile = (uint64_t)blockIdx.x * RD_WAVES + wave; tile < total; tile += nwaves) {
        const uint32_t s = (uint32_t)(tile / tiles_per_stream);
        const uint32_t ti = (uint32_t)(tile - (uint64_t)s * tiles_per_stream);
        const uint8_t *src = lay.iq + (size_t)s * lay.stream_stride + (size_t)ti * RD_TILE_BYTES;
        // global -> LDS, 16 B per lane, 1 KiB contiguous per instruction (LDS address is
        // wave-uniform base + lane*16).  Reads past the stream end land in the next stream
        // or the RD_INPUT_PAD tail; those samples are never turned into output.
#pragma unroll
        for (int j = 0; j < 4; j++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + j * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void *)(my + 32 + j * 1024), 16, 0, 0);
        // halo: the 32 bytes before the tile (previous tile, or the caller's history bytes).
        // With zero history there is nothing to read: run 0 is always re-evaluated exactly.
        const bool has_halo = (ti > 0) || lay.hist_mode;
        if (lane < 2)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(src + (has_halo ? -32 : 0) + lane * 16),
                (__attribute__((address_space(3))) void *)(my), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

        rd_reg_src win;
        const uint4 *chunk = (const uint4 *)(my + 64 * lane);
#pragma unroll
        for (int j = 0; j < 6; j++) {
            const uint4 v = chunk[j];
            win.q[4 * j + 0] = v.x; win.q[4 * j + 1] = v.y; win.q[4 * j + 2] = v.z; win.q[4 * j + 3] = v.w;
        }
        const rd_run_result r = rd_fast_run(win);

        const uint32_t run = ti * 64 + lane;
        const uint32_t t0 = run * RD_RUN;
        if (t0 < lay.n_samples) {
            uint32_t word = r.word;
            const uint32_t left = lay.n_samples - t0;
            if (left < RD_RUN) word &= (1u << left) - 1u;
            lay.bits[(size_t)s * lay.bits_stride + run] = word;
            const bool flag = !(r.nmin > rd_run_threshold(r.fmax)) || (run == 0 && !lay.hist_mode);
            if (flag) {
                const uint32_t idx = atomicAdd(&counters[RD_CNT_FIX], 1u);
                if (idx < fix_cap) fix_list[idx] = s * runs_per_stream + run;
            }
        }
        // the next iteration's LDS-DMA must not overtake this iteration's ds_reads
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

void rd_launch_demod(const rd_layout &lay, uint32_t *fix_list, uint32_t fix_cap, uint32_t *counters, hipStream_t st) {
    const uint32_t tps = (lay.n_samples + RD_TILE_SAMPLES - 1) / RD_TILE_SAMPLES;
    const uint32_t rps = (lay.n_samples + RD_RUN - 1) / RD_RUN;
    const uint64_t total = (uint64_t)lay.n_streams * tps;
    uint64_t wgs = (total + RD_WAVES - 1) / RD_WAVES;
    const uint64_t max_wgs = 256ull * 8;  // 8 workgroups of 4 waves per CU: 32 waves/CU
    if (wgs > max_wgs) wgs = max_wgs;
    if (wgs == 0) return;
    hipLaunchKernelGGL(k_demod_bits, dim3((unsigned)wgs), dim3(RD_WG), 0, st, lay, tps, rps, fix_list, fix_cap,
                       counters);
}

// ------------------------------------------------------------------------------------------
// k_fixup: exact bits for the listed runs (one lane per run).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fixup(rd_layout lay, uint32_t runs_per_stream, const uint32_t *fix_list,
                                               uint32_t fix_cap, const uint32_t *counters, int all) {
    uint64_t count;
    if (all) {
        count = (uint64_t)lay.n_streams * runs_per_stream;
    } else {
        count = counters[RD_CNT_FIX];
        if (count > fix_cap) count = fix_cap;  // overflow is detected and handled by the host
    }
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const uint64_t id = all ? i : fix_list[i];
        const uint32_t s = (uint32_t)(id / runs_per_stream);
        const uint32_t run = (uint32_t)(id - (uint64_t)s * runs_per_stream);
        const long t0 = (long)run * RD_RUN;
        rd_stream_view v;
        v.base = lay.iq + (size_t)s * lay.stream_stride;
        v.valid_from = lay.valid_from;
        v.n = lay.n_samples;
        const long left = (long)lay.n_samples - t0;
        lay.bits[(size_t)s * lay.bits_stride + run] = rd_exact_run(v, t0, left < RD_RUN ? (int)left : RD_RUN);
    }
}

void rd_launch_fixup(const rd_layout &lay, const uint32_t *fix_list, uint32_t fix_cap, uint32_t *counters, int all,
                     hipStream_t st) {
    const uint32_t rps = (lay.n_samples + RD_RUN - 1) / RD_RUN;
    if (rps == 0 || lay.n_streams == 0) return;
    uint64_t want = all ? (uint64_t)lay.n_streams * rps : fix_cap;
    uint64_t wgs = (want + 255) / 256;
    if (wgs > 256ull * 16) wgs = 256ull * 16;
    if (wgs == 0) wgs = 1;
    hipLaunchKernelGGL(k_fixup, dim3((unsigned)wgs), dim3(256), 0, st, lay, rps, fix_list, fix_cap, counters, all);
}

// ------------------------------------------------------------------------------------------
// k_search: bit-parallel preamble match.  One thread = 32 consecutive positions.
// match[p] = AND_m (bit[p + m*S] == preamble[m]).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t rd_word_at(const uint32_t *w, long nwords, long i) {
    return (i >= 0 && i < nwords) ? w[i] : 0u;
}

// 32 bits starting at bit offset o (may be negative / past the end: zeros there)
__device__ __forceinline__ uint32_t rd_bits32_at(const uint32_t *w, long nwords, long o) {
    const long wi = o >> 5;  // floor
    const uint32_t sh = (uint32_t)(o & 31);
    const uint32_t lo = rd_word_at(w, nwords, wi);
    const uint32_t hi = rd_word_at(w, nwords, wi + 1);
    return __builtin_amdgcn_alignbit(hi, lo, sh);
}

__global__ __launch_bounds__(256) void k_search(const uint32_t *bits, size_t bits_stride, int n_streams, long nwords,
                                                long base, long groups_per_stream, long p_lo, long p_hi,
                                                rd_devcfg cfg, rd_match *matches, uint32_t match_cap,
                                                uint32_t *counters) {
    const uint64_t total = (uint64_t)n_streams * groups_per_stream;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride) {
        const uint32_t s = (uint32_t)(g / groups_per_stream);
        const long gi = (long)(g - (uint64_t)s * groups_per_stream);
        const long p0 = base + 32 * gi;
        const uint32_t *w = bits + (size_t)s * bits_stride;
        uint32_t m = 0xFFFFFFFFu;
        for (int k = 0; k < cfg.P; k++) {
            const uint32_t v = rd_bits32_at(w, nwords, p0 + (long)k * cfg.S);
            m &= ((cfg.pre_mask >> k) & 1) ? v : ~v;
        }
        // keep positions inside [p_lo, p_hi]
        if (p0 < p_lo) m &= (p_lo - p0 >= 32) ? 0u : (0xFFFFFFFFu << (p_lo - p0));
        if (p0 + 31 > p_hi) m &= (p_hi < p0) ? 0u : (0xFFFFFFFFu >> (31 - (p_hi - p0)));
        while (m) {
            const int b = __builtin_ctz(m);
            m &= m - 1;
            const uint32_t idx = atomicAdd(&counters[RD_CNT_MATCH], 1u);
            if (idx < match_cap) {
                matches[idx].stream = (int32_t)s;
                matches[idx].pos = (int32_t)(p0 + b);
            }
        }
    }
}

void rd_launch_search(const uint32_t *bits, size_t bits_stride, int n_streams, long n_bits, long p_lo, long p_hi,
                      const rd_devcfg &cfg, rd_match *matches, uint32_t match_cap, uint32_t *counters,
                      hipStream_t st) {
    if (p_hi < p_lo || n_streams == 0) return;
    const long base = (p_lo >> 5) << 5;  // floor to a word boundary (p_lo may be negative)
    const long groups = ((p_hi - base) >> 5) + 1;
    const long nwords = (n_bits + 31) / 32;
    const uint64_t total = (uint64_t)n_streams * groups;
    uint64_t wgs = (total + 255) / 256;
    if (wgs > 256ull * 16) wgs = 256ull * 16;
    hipLaunchKernelGGL(k_search, dim3((unsigned)wgs), dim3(256), 0, st, bits, bits_stride, n_streams, nwords, base,
                       groups, p_lo, p_hi, cfg, matches, match_cap, counters);
}

// ------------------------------------------------------------------------------------------
// k_slice: one wave per match.  Packs packet_symbols bits at stride S (py:197-200) and
// evaluates the reference's RSSI/SNR windows (py:207-236) in float64.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double rd_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return __shfl(v, 0, 64);
}

__device__ __forceinline__ uint32_t rd_bit_at(const uint32_t *w, long nwords, long o) {
    return (rd_word_at(w, nwords, o >> 5) >> (o & 31)) & 1u;
}

template <class View>
__device__ __forceinline__ void rd_emit_record(const View &v, long f_origin, const uint32_t *w, long nwords, long pos,
                                               const rd_devcfg &cfg, int stream, int call, long q, rd_packet *recs,
                                               uint32_t rec_cap, uint32_t *counters, int lane) {
    // filtered[j] = f[f_origin + j - 1], j in [0, B]  (py:133,161: newest block only)
    const long ns = q - cfg.PL < 0 ? 0 : q - cfg.PL;
    const long pe = q + cfg.PL > cfg.B + 1 ? cfg.B + 1 : q + cfg.PL;
    double noise = 0.0, sig = 0.0;
    for (long j = ns + lane; j < pe; j += 64) {
        const rd_d2 f = rd_f_f64(v, f_origin + j - 1);
        const double p = f.x * f.x + f.y * f.y;
        if (j < q) noise += p; else sig += p;
    }
    noise = rd_wave_sum(noise);
    sig = rd_wave_sum(sig);
    uint32_t slot = 0;
    if (lane == 0) slot = atomicAdd(&counters[RD_CNT_REC], 1u);
    slot = __shfl(slot, 0, 64);
    if (slot >= rec_cap) return;
    rd_packet *o = &recs[slot];
    if (lane < RD_MAX_PKT_BYTES) {
        uint32_t byte = 0;
        if (lane < cfg.nbytes) {
            for (int k = 0; k < 8; k++) {
                const int i = lane * 8 + k;
                if (i < cfg.K) byte = (byte << 1) | rd_bit_at(w, nwords, pos + (long)i * cfg.S);
            }
        }
        o->data[lane] = (uint8_t)byte;
    }
    if (lane == 0) {
        o->stream = stream; o->call = call; o->index = (int32_t)q; o->nbytes = cfg.nbytes;
        const double noise_power = (q > ns) ? noise / (double)(q - ns) : 1e-9;
        const double signal_power = (pe > q) ? sig / (double)(pe - q) : __builtin_nan("");
        o->rssi = signal_power > 0 ? 10.0 * log10(signal_power) : -120.0;
        o->snr = noise_power > 0 ? 10.0 * log10(signal_power / noise_power) : 50.0;
    }
}

__global__ __launch_bounds__(256) void k_slice(rd_layout lay, const uint32_t *bits, size_t bits_stride, long nwords,
                                               rd_devcfg cfg, const rd_match *matches, uint32_t match_cap,
                                               int batch_mode, int n_calls, int call, rd_packet *recs,
                                               uint32_t rec_cap, uint32_t *counters) {
    const int lane = threadIdx.x & 63;
    uint32_t count = counters[RD_CNT_MATCH];
    if (count > match_cap) count = match_cap;
    const uint32_t nw = gridDim.x * (blockDim.x >> 6);
    for (uint32_t i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); i < count; i += nw) {
        const rd_match mt = matches[i];
        const uint32_t *w = bits + (size_t)mt.stream * bits_stride;
        rd_stream_view v;
        v.base = lay.iq + (size_t)mt.stream * lay.stream_stride;
        v.valid_from = lay.valid_from;
        v.n = lay.n_samples;
        if (batch_mode) {
            // call b reports absolute positions w_b <= p <= w_b + B, w_b = (b+1)B - L (py:194: q <= B)
            const long pl = (long)mt.pos + cfg.L;
            long b_hi = pl / cfg.B - 1;            // floor: pl >= B because p >= B - L
            long b_lo = (pl % cfg.B == 0) ? b_hi - 1 : b_hi;
            for (long b = b_lo; b <= b_hi; b++) {
                if (b < 0 || b >= n_calls) continue;
                const long q = (long)mt.pos - ((b + 1) * (long)cfg.B - cfg.L);
                rd_emit_record(v, b * (long)cfg.B, w, nwords, mt.pos, cfg, mt.stream, (int)b, q, recs, rec_cap,
                               counters, lane);
            }
        } else {
            rd_emit_record(v, 0, w, nwords, mt.pos, cfg, mt.stream, call, mt.pos, recs, rec_cap, counters, lane);
        }
    }
}

void rd_launch_slice(const rd_layout &lay, const uint32_t *bits, size_t bits_stride, long n_bits, const rd_devcfg &cfg,
                     const rd_match *matches, uint32_t match_cap, int batch_mode, int n_calls, int call,
                     rd_packet *recs, uint32_t rec_cap, uint32_t *counters, hipStream_t st) {
    uint32_t wgs = (match_cap + 3) / 4;
    if (wgs > 2048) wgs = 2048;
    if (wgs == 0) wgs = 1;
    hipLaunchKernelGGL(k_slice, dim3(wgs), dim3(256), 0, st, lay, bits, bits_stride, (n_bits + 31) / 32, cfg, matches,
                       match_cap, batch_mode, n_calls, call, recs, rec_cap, counters);
}

// ------------------------------------------------------------------------------------------
// float64 values of d and f (state mirrors / parse()'s frequency error, protocol.py:307-311)
// ------------------------------------------------------------------------------------------
template <class View>
__global__ __launch_bounds__(256) void k_disc(View v, long t0, long n, double *out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long t = t0 + i;
    out[i] = rd_disc_f64(rd_f_f64(v, t - 1), rd_f_f64(v, t));
}

template <class View>
__global__ __launch_bounds__(256) void k_filt(View v, long t0, long n, double *out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const rd_d2 f = rd_f_f64(v, t0 + i);
    out[2 * i] = f.x;
    out[2 * i + 1] = f.y;
}

static rd_stream_view make_view(const rd_layout &lay, int stream) {
    rd_stream_view v;
    v.base = lay.iq + (size_t)stream * lay.stream_stride;
    v.valid_from = lay.valid_from;
    v.n = lay.n_samples;
    return v;
}

void rd_launch_disc(const rd_layout &lay, int stream, long t0, long n, double *out, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_disc<rd_stream_view>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                       make_view(lay, stream), t0, n, out);
}

void rd_launch_filtered(const rd_layout &lay, int stream, long t0, long n, double *out, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_filt<rd_stream_view>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                       make_view(lay, stream), t0, n, out);
}

// ------------------------------------------------------------------------------------------
// streaming search window: out = (in >> n_block_bits) | (block << (n_win_bits - n_block_bits))
// i.e. np.roll(quantized, -B) followed by writing the newest block at the end (py:157,163-166)
// ------------------------------------------------------------------------------------------
__global__ void k_window_update(uint32_t *out, const uint32_t *in, long n_win_bits, const uint32_t *block,
                                long n_block_bits) {
    const long nw = (n_win_bits + 31) / 32;
    const long nbw = (n_block_bits + 31) / 32;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nw) return;
    const long keep = n_win_bits - n_block_bits;  // bits [0, keep) come from the old window
    const long o = 32 * i;
    const uint32_t oldv = rd_bits32_at(in, nw, o + n_block_bits);
    const uint32_t newv = rd_bits32_at(block, nbw, o - keep);
    uint32_t mask_old;
    if (o + 32 <= keep) mask_old = 0xFFFFFFFFu;
    else if (o >= keep) mask_old = 0u;
    else mask_old = (1u << (keep - o)) - 1u;
    uint32_t word = (oldv & mask_old) | (newv & ~mask_old);
    const long left = n_win_bits - o;
    if (left < 32) word &= (1u << left) - 1u;
    out[i] = word;
}

void rd_launch_window_update(uint32_t *win_out, const uint32_t *win_in, long n_win_bits, const uint32_t *block,
                             long n_block_bits, hipStream_t st) {
    const long nw = (n_win_bits + 31) / 32;
    hipLaunchKernelGGL(k_window_update, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, st, win_out, win_in,
                       n_win_bits, block, n_block_bits);
}

// ------------------------------------------------------------------------------------------
// complex128 input branch (py:144-150), float64 throughout
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_cplx_bits(rd_cplx_view v, uint32_t *bits) {
    const long run = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long t0 = run * 32;
    if (t0 >= v.n) return;
    rd_d2 prev = rd_f_f64(v, t0 - 1);
    uint32_t word = 0;
    for (int r = 0; r < 32 && t0 + r < v.n; r++) {
        const rd_d2 cur = rd_f_f64(v, t0 + r);
        word |= rd_signbit_f64(rd_disc_f64(prev, cur)) << r;
        prev = cur;
    }
    bits[run] = word;
}

void rd_launch_cplx_bits(const rd_cplx_layout &lay, uint32_t *bits, hipStream_t st) {
    const long runs = (lay.n + 31) / 32;
    if (runs <= 0) return;
    rd_cplx_view v = {lay.x, lay.valid_from, lay.n};
    hipLaunchKernelGGL(k_cplx_bits, dim3((unsigned)((runs + 255) / 256)), dim3(256), 0, st, v, bits);
}

void rd_launch_cplx_disc(const rd_cplx_layout &lay, long t0, long n, double *out, hipStream_t st) {
    if (n <= 0) return;
    rd_cplx_view v = {lay.x, lay.valid_from, lay.n};
    hipLaunchKernelGGL(k_disc<rd_cplx_view>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, v, t0, n, out);
}

void rd_launch_cplx_filtered(const rd_cplx_layout &lay, long t0, long n, double *out, hipStream_t st) {
    if (n <= 0) return;
    rd_cplx_view v = {lay.x, lay.valid_from, lay.n};
    hipLaunchKernelGGL(k_filt<rd_cplx_view>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, v, t0, n, out);
}

__global__ __launch_bounds__(256) void k_cplx_slice(rd_cplx_view v, const uint32_t *bits, long nwords, rd_devcfg cfg,
                                                    const rd_match *matches, uint32_t match_cap, int call,
                                                    rd_packet *recs, uint32_t rec_cap, uint32_t *counters) {
    const int lane = threadIdx.x & 63;
    uint32_t count = counters[RD_CNT_MATCH];
    if (count > match_cap) count = match_cap;
    const uint32_t nw = gridDim.x * (blockDim.x >> 6);
    for (uint32_t i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); i < count; i += nw) {
        const rd_match mt = matches[i];
        rd_emit_record(v, 0, bits, nwords, mt.pos, cfg, 0, call, mt.pos, recs, rec_cap, counters, lane);
    }
}

void rd_launch_cplx_slice(const rd_cplx_layout &lay, const uint32_t *bits, long n_bits, const rd_devcfg &cfg,
                          const rd_match *matches, uint32_t match_cap, int call, rd_packet *recs, uint32_t rec_cap,
                          uint32_t *counters, hipStream_t st) {
    uint32_t wgs = (match_cap + 3) / 4;
    if (wgs > 2048) wgs = 2048;
    if (wgs == 0) wgs = 1;
    rd_cplx_view v = {lay.x, lay.valid_from, lay.n};
    hipLaunchKernelGGL(k_cplx_slice, dim3(wgs), dim3(256), 0, st, v, bits, (n_bits + 31) / 32, cfg, matches,
                       match_cap, call, recs, rec_cap, counters);
}

// ------------------------------------------------------------------------------------------
// stage kernels: one per reference stage function, float64, on device arrays
// ------------------------------------------------------------------------------------------
__global__ void k_lut(const uint8_t *in, double *out, size_t n) {  // py:38-39
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[2 * i] = ((double)in[2 * i] - 127.4) / 127.6;
    out[2 * i + 1] = ((double)in[2 * i + 1] - 127.4) / 127.6;
}
void rd_launch_lut(const uint8_t *in, double *out, size_t n, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_lut, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, out, n);
}

__global__ void k_rotate(const double *in, double *out, size_t n) {  // py:46-49
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const rd_d2 y = rd_rot_f64(in[2 * i], in[2 * i + 1], (long)i);
    out[2 * i] = y.x;
    out[2 * i + 1] = y.y;
}
void rd_launch_rotate(const double *in, double *out, size_t n, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_rotate, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, out, n);
}

__global__ void k_fir9(const double *in, double *out, size_t n_out) {  // py:71-73
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    const double c[9] = {RD_C0, RD_C1, RD_C2, RD_C3, RD_C4, RD_C3, RD_C2, RD_C1, RD_C0};
    double ar = 0.0, ai = 0.0;
#pragma unroll
    for (int m = 0; m < 9; m++) {
        ar += c[m] * in[2 * (i + m)];
        ai += c[m] * in[2 * (i + m) + 1];
    }
    out[2 * i] = ar;
    out[2 * i + 1] = ai;
}
void rd_launch_fir9(const double *in, double *out, size_t n_out, hipStream_t st) {
    if (n_out) hipLaunchKernelGGL(k_fir9, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, st, in, out, n_out);
}

__global__ void k_discriminate(const double *in, double *out, size_t n_out) {  // py:80-90
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    const rd_d2 n = {in[2 * i], in[2 * i + 1]}, np = {in[2 * i + 2], in[2 * i + 3]};
    out[i] = rd_disc_f64(n, np);
}
void rd_launch_discriminate(const double *in, double *out, size_t n_out, hipStream_t st) {
    if (n_out)
        hipLaunchKernelGGL(k_discriminate, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, st, in, out, n_out);
}

__global__ void k_quantize(const double *in, uint8_t *out, size_t n) {  // py:98
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (uint8_t)rd_signbit_f64(in[i]);
}
void rd_launch_quantize(const double *in, uint8_t *out, size_t n, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_quantize, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, out, n);
}

__global__ void k_pack_bytes(const uint8_t *in01, uint32_t *words, size_t n) {  // go:105-113 Pack
    const size_t wi = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (wi * 32 >= n) return;
    uint32_t w = 0;
    for (int b = 0; b < 32 && wi * 32 + b < n; b++) w |= (uint32_t)(in01[wi * 32 + b] & 1u) << b;
    words[wi] = w;
}
void rd_launch_pack_bytes(const uint8_t *in01, uint32_t *words, size_t n, hipStream_t st) {
    const size_t nw = (n + 31) / 32;
    if (nw) hipLaunchKernelGGL(k_pack_bytes, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, st, in01, words, n);
}
