// rd_kernels.hip - gfx950 kernels of the rtldavis IQ -> bits -> packets path other than the fused demod kernel
// (rd_demod_mfma.hip).
//
// Reference stages (py = /root/reference/src/rtldavis/dsp.py, go = /root/reference/dsp/dsp.go):
//   k_tail         : everything behind the demod kernel in ONE launch (batch path, Davis shape): exact bits for the
//                    listed groups, Demodulator._search py:171-188, ._slice py:190-246 incl. order, dedupe, RSSI / SNR
//   k_fixup        : exact integer re-evaluation of the listed groups (every other shape, the streaming legacy path)
//   k_search       : Demodulator._search py:171-188 (go:115-131) on packed bits, bit-parallel
//   k_classify + k_rssi_u8, k_slice_rssi : Demodulator._slice py:190-246 without the final order (the host orders)
//   k_stream_block : one Demodulator.demodulate() call py:139-246 in one launch (streaming handle)
//   k_parse_select / k_freq_err : Parser.parse front half, protocol.py:290-311 (opt-in)
//   k_disc/k_filt  : float64 discriminate / fir9 values for the state mirrors py:133-134
//   k_cplx_*       : the complex-input branch py:144-150 in float64
//   stage kernels  : one float64 kernel per reference stage function
//   k_demod_bits   : the round-1 demod kernel (FIR on the VALU), diagnostic library only (RD_K1_IMPL=valu)
#include <algorithm>
#include <cstdlib>

#include <hip/hip_ext.h>

#include "rd_internal.h"
#include "rd_math.h"
#include "rd_mfma.h"

#ifdef RD_DIAG   // the round-1 demod kernel: A/B runs only
#define RD_WG 256
#define RD_WAVES (RD_WG / 64)
#define RD_LDS_WAVE (32 + RD_TILE_BYTES)  // 32 B halo + one 4 KiB tile, private to a wave
#define RD_PEND 128                       // guard-band run ids staged per wave before one atomic

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
            case 0: asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(r) : "v"(d)); break;
            case 1: asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(r) : "v"(d)); break;
            case 2: asm volatile("v_cvt_f32_ubyte2 %0, %1" : "=v"(r) : "v"(d)); break;
            default: asm volatile("v_cvt_f32_ubyte3 %0, %1" : "=v"(r) : "v"(d)); break;
        }
        return r;
    }
};

// Append a wave's staged ids to the global list with one atomic (count is wave-uniform).
__device__ __forceinline__ void rd_flush_pending(const uint32_t *pend, uint32_t count, uint32_t *fix_list,
                                                 uint32_t fix_cap, uint32_t *counters, int lane) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // staged ids written by other lanes
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&counters[RD_CNT_FIX], count);
    base = __builtin_amdgcn_readfirstlane(base);
    for (uint32_t i = lane; i < count; i += 64)
        if (base + i < fix_cap) fix_list[base + i] = pend[i];
}

// ------------------------------------------------------------------------------------------
// k_demod_bits: one wave = one 2048-sample tile per iteration, grid-stride over tiles.
// LDS is wave-private (no workgroup barrier anywhere): [halo 32 B][tile 4096 B].
// ------------------------------------------------------------------------------------------
// LDS image of a tile (wave-private).  One global_load_lds_dwordx4 writes 64 consecutive
// 16-byte slots (slot = lane), so the *source* chunk of each lane is permuted instead
// (cdna_hip_programming.md rule 21): slot 64j + l holds chunk 64j + 4(l%16) + l/16 of the
// tile.  Lane L then finds its four chunks 4L..4L+3 at slots 64(L/16) + 16i + L%16 - for a
// fixed i, 16 neighbouring lanes read 16 neighbouring slots: ds_read_b128 without bank
// conflicts (a plain linear image would be a 4-way conflict at the 64-byte lane stride).
__device__ __forceinline__ void rd_issue_tile_loads(const rd_layout &lay, uint32_t s, uint32_t ti, uint8_t *my,
                                                    int lane) {
    const uint8_t *src = lay.iq + (size_t)s * lay.stream_stride + (size_t)ti * RD_TILE_BYTES;
    const int perm = 4 * (lane & 15) + (lane >> 4);
    // the instruction offset advances the global and the LDS address alike: one address pair,
    // four immediates
    const __attribute__((address_space(1))) void *g0 = (const __attribute__((address_space(1))) void *)(src + perm * 16);
    __attribute__((address_space(3))) void *l0 = (__attribute__((address_space(3))) void *)(my + 32);
    __builtin_amdgcn_global_load_lds(g0, l0, 16, 0, 0);
    __builtin_amdgcn_global_load_lds(g0, l0, 16, 1024, 0);
    __builtin_amdgcn_global_load_lds(g0, l0, 16, 2048, 0);
    __builtin_amdgcn_global_load_lds(g0, l0, 16, 3072, 0);
    // halo: the 32 bytes before the tile (previous tile, or the caller's history bytes).
    // With zero history there is nothing to read: run 0 is always re-evaluated exactly.
    const bool has_halo = (ti > 0) || lay.hist_mode;
    if (lane < 2)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void *)(src + (has_halo ? -32 : 0) + lane * 16),
            (__attribute__((address_space(3))) void *)(my), 16, 0, 0);
}

// DBG (compile-time, 0 in the shipped instantiation): 1 = no global loads (compute on whatever
// LDS holds), 2 = loads and LDS reads only, no arithmetic - timing ablations, results are garbage.
template <int DBG, int NPK>
__global__ __launch_bounds__(RD_WG, 4) void k_demod_bits(rd_layout lay, uint32_t tiles_per_stream,
                                                      uint32_t runs_per_stream, uint32_t *fix_list,
                                                      uint32_t fix_cap, uint32_t *counters) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[RD_WAVES][RD_LDS_WAVE];
    // Guard-band run ids are staged per wave and appended to the global list RD_PEND at a
    // time: one returning atomic per ~100 ids instead of one per tile (a single counter
    // word sustains only ~90 atomics/us, MI355X_MICROARCH.md "dequeue").
    __shared__ uint32_t pend[RD_WAVES][RD_PEND];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar: tile bookkeeping stays on the SALU
    uint8_t *my = lds[wave];
    uint32_t *mypend = pend[wave];
    uint32_t npend = 0;  // wave-uniform
    const uint32_t nwaves = gridDim.x * RD_WAVES;
    // tile = s * tiles_per_stream + ti advances by nwaves per iteration: (s, ti) += (dq, dr) with carry
    const uint32_t dq = nwaves / tiles_per_stream, dr = nwaves % tiles_per_stream;
    // per-lane LDS addresses: own chunks i = 0..3 at own + 256 i; the previous lane's chunks
    // 2, 3 (this lane's halo) at prev + 512, prev + 768; lane 0 takes the halo region.
    const uint8_t *own = my + 32 + 16 * (64 * (lane >> 4) + (lane & 15));
    const int pl = lane - 1;
    const uint8_t *h0 = lane ? my + 32 + 16 * (64 * (pl >> 4) + 32 + (pl & 15)) : my;
    const uint8_t *h1 = lane ? h0 + 256 : my + 16;

    // The packed word of tile i is stored at the start of iteration i+1, BEFORE the next
    // tile's loads are issued: stores count in vmcnt too, and a store issued after the loads
    // would make every `s_waitcnt vmcnt(0)` wait for its write latency as well (-20 % measured).
    uint32_t st_word = 0;
    uint32_t *st_ptr = nullptr;
    const uint32_t first = blockIdx.x * RD_WAVES + wave;
    uint32_t s = first / tiles_per_stream, ti = first % tiles_per_stream;
    if (DBG != 1 && s < (uint32_t)lay.n_streams) rd_issue_tile_loads(lay, s, ti, my, lane);
    while (s < (uint32_t)lay.n_streams) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this tile has landed in LDS
        rd_reg_src win;
        {
            const uint4 a = *(const uint4 *)h0, b = *(const uint4 *)h1;
            win.q[0] = a.x; win.q[1] = a.y; win.q[2] = a.z; win.q[3] = a.w;
            win.q[4] = b.x; win.q[5] = b.y; win.q[6] = b.z; win.q[7] = b.w;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint4 v = *(const uint4 *)(own + 256 * i);
                win.q[8 + 4 * i] = v.x; win.q[9 + 4 * i] = v.y; win.q[10 + 4 * i] = v.z; win.q[11 + 4 * i] = v.w;
            }
        }
        // The window is in registers: the same LDS buffer can take the next tile while this
        // one is computed (~2000 VALU instructions cover the HBM latency).
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (st_ptr) *st_ptr = st_word;  // previous tile's output
        st_ptr = nullptr;
        uint32_t ns = s + dq, nti = ti + dr;
        if (nti >= tiles_per_stream) { nti -= tiles_per_stream; ns++; }
        if (DBG != 1 && ns < (uint32_t)lay.n_streams) rd_issue_tile_loads(lay, ns, nti, my, lane);

        rd_run_result r;
        if (DBG == 2) {
            uint32_t x = 0;
#pragma unroll
            for (int j = 0; j < 24; j++) x ^= win.q[j];
            r.word = x; r.fmax = 1.0f;
#pragma unroll
            for (int g = 0; g < RD_GROUPS; g++) r.nmin[g] = 1.0e30f;
        } else {
            r = rd_fast_run<NPK>(win);
        }

        const uint32_t run = ti * 64 + lane;
        const uint32_t t0 = run * RD_RUN;
        uint32_t gmask = 0;  // groups of 8 samples inside the guard band
        // only a stream's last tile can be ragged (wave-uniform test): everywhere else the whole
        // run is valid and none of the tail masking below is executed
        const bool ragged = (ti + 1 == tiles_per_stream) && (lay.n_samples % RD_TILE_SAMPLES) != 0;
        if (!ragged) {
            st_word = r.word;
            st_ptr = &lay.bits[(size_t)s * lay.bits_stride + run];
            gmask = (run == 0 && !lay.hist_mode) ? 0xFu : rd_guard_mask(r);
        } else if (t0 < lay.n_samples) {
            uint32_t word = r.word;
            const uint32_t left = lay.n_samples - t0;
            if (left < RD_RUN) word &= (1u << left) - 1u;
            st_word = word;
            st_ptr = &lay.bits[(size_t)s * lay.bits_stride + run];
            gmask = (run == 0 && !lay.hist_mode) ? 0xFu : rd_guard_mask(r);
            if (left < RD_RUN) gmask &= (1u << ((left + RD_GROUP - 1) / RD_GROUP)) - 1u;
        }
        const uint64_t fm = __ballot(gmask != 0);
        if (fm) {  // wave-uniform, ~60 % of the tiles on noise
            // one list entry per flagged run: (word index in the bits array) << 4 | group mask
            const uint32_t nf = (uint32_t)__popcll(fm);
            if (npend + nf > RD_PEND) {
                rd_flush_pending(mypend, npend, fix_list, fix_cap, counters, lane);
                npend = 0;
            }
            if (gmask)
                mypend[npend + __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32),
                                                         __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0))] =
                    ((uint32_t)((size_t)s * lay.bits_stride + run) << 4) | gmask;
            npend += nf;
        }
        s = ns;
        ti = nti;
    }
    if (st_ptr) *st_ptr = st_word;
    if (npend) rd_flush_pending(mypend, npend, fix_list, fix_cap, counters, lane);
}

#endif  // RD_DIAG (k_demod_bits)

// Environment switches of this file, read once per process (function-local static: thread-safe first use).
struct rd_k_params {
    int impl_mfma = 1, per_cu_valu = 7, dbg = 0, npk = RD_NPK_DEFAULT, slice_two = 1, search_cap = 32 / 4;
    rd_k_params() {
        const char *e = getenv("RD_SLICE_IMPL");  // "wave": the one-kernel slice on the batch path as well (A/B)
        slice_two = (e && e[0] == 'w') ? 0 : 1;
        e = getenv("RD_K2_WGS_PER_CU");  // tuning knob of k_search
        search_cap = e ? atoi(e) : 32 / 4;
        if (search_cap < 1 || search_cap > 64) search_cap = 32 / 4;
#ifdef RD_DIAG
        e = getenv("RD_K1_IMPL");  // "valu": the round-1 demod kernel (FIR on the VALU) for A/B runs
        impl_mfma = (e && e[0] == 'v') ? 0 : 1;
        e = getenv("RD_K1_WGS_PER_CU");
        per_cu_valu = e ? atoi(e) : 7;
        if (per_cu_valu < 1 || per_cu_valu > 8) per_cu_valu = 7;
        e = getenv("RD_K1_DEBUG");  // timing ablations with garbage results
        dbg = e ? atoi(e) : 0;
        npk = RD_NPK_DEFAULT;
        e = getenv("RD_K1_NPK");    // packed FIR steps per output, for tuning sweeps
        if (e) npk = atoi(e);
#endif
    }
};
static const rd_k_params &rd_k_get_params() {
    static const rd_k_params p;
    return p;
}

uint32_t rd_launch_demod(const rd_layout &lay, uint32_t *fix_list, uint32_t fix_cap, uint32_t *counters, hipStream_t st,
                         hipEvent_t ev_start, hipEvent_t ev_stop, uint32_t flags, uint32_t *chunk_out, uint32_t *bucket_cnt) {
#ifdef RD_DIAG
    const rd_k_params &P = rd_k_get_params();
    if (!P.impl_mfma) {
        const uint32_t tps = (lay.n_samples + RD_TILE_SAMPLES - 1) / RD_TILE_SAMPLES;
        const uint32_t rps = (lay.n_samples + RD_RUN - 1) / RD_RUN;
        const uint64_t total = (uint64_t)lay.n_streams * tps;
        uint64_t wgs = (total + RD_WAVES - 1) / RD_WAVES;
        // persistent grid: workgroups per CU (72 VGPRs and 18.5 KiB LDS admit 7); RD_K1_WGS_PER_CU overrides
        const uint64_t max_wgs = 256ull * P.per_cu_valu;
        if (wgs > max_wgs) wgs = max_wgs;
        if (wgs == 0) return 0;

    // With events given, the dispatch itself carries them (hipExtLaunchKernelGGL): its begin / end
    // timestamps, without the marker packets of hipEventRecord that idle the GPU for ~6 us each.
#define RD_LAUNCH_K1(D, N)                                                                                          \
    do {                                                                                                            \
        if (ev_start || ev_stop)                                                                                    \
            hipExtLaunchKernelGGL((k_demod_bits<D, N>), dim3((unsigned)wgs), dim3(RD_WG), 0, st, ev_start, ev_stop, \
                                  0, lay, tps, rps, fix_list, fix_cap, counters);                                   \
        else                                                                                                        \
            hipLaunchKernelGGL((k_demod_bits<D, N>), dim3((unsigned)wgs), dim3(RD_WG), 0, st, lay, tps, rps,        \
                               fix_list, fix_cap, counters);                                                        \
    } while (0)
        if (P.dbg == 1) { RD_LAUNCH_K1(1, RD_NPK_DEFAULT); return 0; }
        if (P.dbg == 2) { RD_LAUNCH_K1(2, RD_NPK_DEFAULT); return 0; }
        if (P.npk == 0) { RD_LAUNCH_K1(0, 0); return 0; }
        if (P.npk == 2) { RD_LAUNCH_K1(0, 2); return 0; }
        if (P.npk == 4) { RD_LAUNCH_K1(0, 4); return 0; }
        if (P.npk == 6) { RD_LAUNCH_K1(0, 6); return 0; }
        if (P.npk == 7) { RD_LAUNCH_K1(0, 7); return 0; }
        if (P.npk == 9) { RD_LAUNCH_K1(0, 9); return 0; }
        RD_LAUNCH_K1(0, RD_NPK_DEFAULT);
#undef RD_LAUNCH_K1
        return 0;
    }
#endif
    if (!bucket_cnt) flags &= ~RD_DEMOD_FIX_BUCKETS;
    const bool bucketed = rd_launch_demod_mfma(lay, fix_list, fix_cap, counters, st, ev_start, ev_stop, nullptr, flags, chunk_out, bucket_cnt);
    return bucketed ? RD_DEMOD_FIX_BUCKETS : 0u;
}

// ------------------------------------------------------------------------------------------
// k_fixup: exact bits for the listed runs (one lane per run).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fixup(rd_layout lay, uint32_t runs_per_stream, const uint32_t *fix_list,
                                               uint32_t fix_cap, const uint32_t *counters, int all,
                                               uint32_t *zero_next, uint32_t zero_words) {
    // the counters of the handle's NEXT run (double-buffered) are cleared here, saving a memset launch
    if (zero_next && blockIdx.x == 0)
        for (uint32_t i = threadIdx.x; i < zero_words; i += blockDim.x) zero_next[i] = 0;
    uint64_t count;
    if (all) {
        count = (uint64_t)lay.n_streams * runs_per_stream;
    } else {
        count = counters[RD_CNT_FIX];
        if (count > fix_cap) count = fix_cap;  // overflow is detected and handled by the host
    }
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        rd_stream_view v;
        v.valid_from = lay.valid_from;
        v.n = lay.n_samples;
        if (all) {  // every run, one output word each
            const uint32_t s = (uint32_t)(i / runs_per_stream);
            const uint32_t run = (uint32_t)(i - (uint64_t)s * runs_per_stream);
            const long t0 = (long)run * RD_RUN;
            v.base = lay.iq + (size_t)s * lay.stream_stride;
            const long left = (long)lay.n_samples - t0;
            lay.bits[(size_t)s * lay.bits_stride + run] = rd_exact_run(v, t0, left < RD_RUN ? (int)left : RD_RUN);
        } else {  // listed runs: word index << 4 | mask of the 8-sample groups to re-evaluate
            const uint32_t e = fix_list[i];
            const uint32_t widx = e >> 4;
            const uint32_t s = widx / (uint32_t)lay.bits_stride;
            const uint32_t run = widx - s * (uint32_t)lay.bits_stride;
            v.base = lay.iq + (size_t)s * lay.stream_stride;
            // Where the 39 us go (ablations): launch + list read 9, byte stores 6.5, IQ reads 11,
            // float64 arithmetic 14.  Tried and measured no faster: a per-lane "lowest flagged group
            // first" loop (25 % slower), expanding the entries into one (word, group) item per lane
            // through LDS (-2 us), reading the run's window from a
            // scratch area the demod kernel fills from its registers (same time; demod kernel +6 %).
            for (int g = 0; g < RD_GROUPS; g++) {
                if (!((e >> g) & 1)) continue;
                const long t0 = (long)run * RD_RUN + g * RD_GROUP;
                const long left = (long)lay.n_samples - t0;
                if (left <= 0) break;
                const int count = left < RD_GROUP ? (int)left : RD_GROUP;
                // samples t0-10 .. t0+9 = 40 bytes at a 4-byte aligned address: ten independent loads.
                // Dwords wholly before the first readable sample are not touched (no history there).
                const uint8_t *p = v.base + 2 * (t0 - 10);
                uint32_t dw[10];
                // dword d holds samples t0-10+2d and +1.  Readable: its last sample is at or after
                // valid_from (d >= floor(x / 2), x = valid_from - (t0-10)) and its first sample lies
                // before the input's padded end (d < ceil(z / 2), z = n_samples + 8 - (t0-10)).
                // Two 64-bit subtractions, then 32-bit compares against constants.
                const long x = v.valid_from - (t0 - 10), z = (long)lay.n_samples + 8 - (t0 - 10);
                const int d_lo = x <= 0 ? 0 : x >= 20 ? 10 : (int)(x / 2);
                const int d_hi = z <= 0 ? 0 : z >= 20 ? 10 : (int)((z + 1) / 2);
#pragma unroll
                for (int d = 0; d < 10; d++) dw[d] = (d >= d_lo && d < d_hi) ? *(const uint32_t *)(p + 4 * d) : 0u;
                ((uint8_t *)lay.bits)[(size_t)widx * 4 + g] = (uint8_t)rd_exact_group_dw(dw, t0, count, v.valid_from);
            }
        }
    }
}

void rd_launch_fixup(const rd_layout &lay, const uint32_t *fix_list, uint32_t fix_cap, uint32_t *counters, int all,
                     uint32_t *zero_next, hipStream_t st, uint32_t zero_words, uint64_t expect) {
    const uint32_t rps = (lay.n_samples + RD_RUN - 1) / RD_RUN;
    if (rps == 0 || lay.n_streams == 0) return;
    uint64_t want = all ? (uint64_t)lay.n_streams * rps : fix_cap;
    // a list that is 4 % full does not need a grid for all of it: 4096 workgroups of which 3400 find nothing to do
    // take longer to dispatch than the listed groups take to re-evaluate
    if (!all && expect > 0) want = std::min<uint64_t>(want, std::max<uint64_t>(expect + expect / 4, 256 * 256));
    uint64_t wgs = (want + 255) / 256;
    if (wgs > 256ull * 16) wgs = 256ull * 16;
    if (wgs == 0) wgs = 1;
    hipLaunchKernelGGL(k_fixup, dim3((unsigned)wgs), dim3(256), 0, st, lay, rps, fix_list, fix_cap, counters, all,
                       zero_next, zero_words);
}

// ------------------------------------------------------------------------------------------
// k_search: bit-parallel preamble match (py:171-188, go:115-131).
// match[p] = AND_k (bit[p + k*S] == preamble[k]).  One lane = 128 consecutive positions
// (4 output words); the words that cover them are loaded once and every tap is a funnel
// shift (v_alignbit_b32).  Persistent waves; matches are staged in LDS per wave and appended
// to the global list ~100 at a time (a single counter word sustains only ~90 atomics/us).
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

// 32-bit variant (bit offsets inside one stream's array always fit)
__device__ __forceinline__ uint32_t rd_bits32_at_i(const uint32_t *w, int nwords, int o) {
    const int wi = o >> 5;  // arithmetic shift: floor
    const uint32_t lo = (wi >= 0 && wi < nwords) ? w[wi] : 0u;
    const uint32_t hi = (wi + 1 >= 0 && wi + 1 < nwords) ? w[wi + 1] : 0u;
    return __builtin_amdgcn_alignbit(hi, lo, (uint32_t)(o & 31));
}

#ifndef RD_SEARCH_OUT
#define RD_SEARCH_OUT 4     // output words (32 positions each) per lane (a multiple of 4: 16-byte loads)
#endif
#define RD_MATCH_PEND 128   // staged matches per wave
#define RD_SEARCH_WAVES 4    // waves per workgroup (16 was tried to cut the end-of-kernel atomics: no gain)
#ifndef RD_SEARCH_UNROLL
#define RD_SEARCH_UNROLL 2
#endif

__device__ __forceinline__ void rd_flush_matches(const rd_match *pend, uint32_t count, rd_match *matches,
                                                 uint32_t match_cap, uint32_t *counters, int lane) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&counters[RD_CNT_MATCH], count);
    base = __builtin_amdgcn_readfirstlane(base);
    for (uint32_t i = lane; i < count; i += 64)
        if (base + i < match_cap) matches[base + i] = pend[i];
}

// S_ > 0: compile-time symbol length / preamble length (register funnel with constant
// shifts); S_ == 0: run-time cfg.S / cfg.P (words fetched per tap).
template <int S_, int P_, uint64_t PRE_>
__global__ __launch_bounds__(64 * RD_SEARCH_WAVES) void k_search(const uint32_t *bits, size_t bits_stride, int n_streams, long nwords,
                                                long base, long groups_per_stream, long p_lo, long p_hi,
                                                rd_devcfg cfg, rd_match *matches, uint32_t match_cap,
                                                uint32_t *counters) {
    __shared__ rd_match pend_all[RD_SEARCH_WAVES][RD_MATCH_PEND];
    const int lane = threadIdx.x & 63;
    rd_match *pend = pend_all[threadIdx.x >> 6];
    uint32_t npend = 0;  // wave-uniform
    // a wave covers 64 consecutive lane-groups (128 positions each) of ONE stream, so the
    // stream / group split is scalar arithmetic
    const uint32_t wgps = (uint32_t)((groups_per_stream + 63) / 64);  // wave-groups per stream
    const uint32_t nwave_groups = (uint32_t)n_streams * wgps;
    const uint32_t nwaves = gridDim.x * RD_SEARCH_WAVES;
    const uint32_t wave0 = __builtin_amdgcn_readfirstlane(blockIdx.x * RD_SEARCH_WAVES + (threadIdx.x >> 6));
    // RD_SEARCH_UNROLL wave-groups per trip: all their loads are issued before the first is used.
    // Everything per lane is 32-bit (positions and word indices of one stream fit), the stream /
    // group split of a wave-group is carried in scalars (no division in the loop): half of this
    // kernel's instructions used to be 64-bit index arithmetic and bounds tests.
    constexpr int NW = S_ > 0 ? ((RD_SEARCH_OUT + ((P_ > 0 ? P_ - 1 : 0) * (S_ > 0 ? S_ : 1) + 31) / 32 + 1 + 3) / 4) * 4 : 4;
    const int nwords_i = (int)nwords, gps = (int)groups_per_stream, plo = (int)p_lo, phi = (int)p_hi;
    const int base_i = (int)base, base_w = (int)(base >> 5);
    // 16-byte loads need every stream's words and the first word index 16-byte aligned
    const bool aligned16 = (bits_stride % 4 == 0) && (base_w % 4 == 0) && ((size_t)bits % 16 == 0);
    // wave-group wg = s * wgps + rem; a trip advances by step = nwaves * UNROLL, a slot by nwaves
    const uint32_t dq_n = nwaves / wgps, dr_n = nwaves % wgps;
    uint32_t s0 = wave0 / wgps, rem0 = wave0 % wgps;
    for (uint32_t wg0 = wave0; wg0 < nwave_groups; wg0 += nwaves * RD_SEARCH_UNROLL) {
        uint32_t r[RD_SEARCH_UNROLL][NW];
        uint32_t su[RD_SEARCH_UNROLL], remu[RD_SEARCH_UNROLL];
        {
            uint32_t sx = s0, rx = rem0;
#pragma unroll
            for (int u = 0; u < RD_SEARCH_UNROLL; u++) {
                su[u] = sx; remu[u] = rx;
                sx += dq_n; rx += dr_n;
                if (rx >= wgps) { rx -= wgps; sx++; }
            }
            s0 = sx; rem0 = rx;  // the next trip's first slot
        }
        if constexpr (S_ > 0) {
#pragma unroll
            for (int u = 0; u < RD_SEARCH_UNROLL; u++) {
                const uint32_t wg = wg0 + u * nwaves;
                if (wg >= nwave_groups) break;  // wave-uniform
                const int gi = (int)remu[u] * 64 + lane;
                const uint32_t *w = bits + (size_t)su[u] * bits_stride;
                const int w0 = base_w + RD_SEARCH_OUT * gi;
                if (aligned16 && gi < gps && w0 >= 0 && w0 + NW <= nwords_i) {
                    // interior: NW/4 coalesced 16-byte loads (lane i reads bytes 16i.. of the wave's span)
#pragma unroll
                    for (int j = 0; j < NW / 4; j++) {
                        const uint4 v4 = *(const uint4 *)(w + w0 + 4 * j);
                        r[u][4 * j] = v4.x; r[u][4 * j + 1] = v4.y; r[u][4 * j + 2] = v4.z; r[u][4 * j + 3] = v4.w;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < NW; j++) r[u][j] = gi < gps ? rd_word_at(w, nwords, (long)w0 + j) : 0u;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < RD_SEARCH_UNROLL; u++) {
            const uint32_t wg = wg0 + u * nwaves;
            if (wg >= nwave_groups) break;  // wave-uniform
            const uint32_t s = su[u];
            const int gi = (int)remu[u] * 64 + lane;
            uint32_t m[RD_SEARCH_OUT] = {};
            int p0 = 0;
            if (gi < gps) {
                p0 = base_i + 32 * RD_SEARCH_OUT * gi;
                const uint32_t *w = bits + (size_t)s * bits_stride;
#pragma unroll
                for (int o = 0; o < RD_SEARCH_OUT; o++) m[o] = 0xFFFFFFFFu;
                if constexpr (S_ > 0) {
                    // compile-time preamble: all[o] = AND of the taps that must be 1, any[o] = OR of
                    // the taps that must be 0 (two at a time with v_or3_b32); match = all & ~any
                    uint32_t any[RD_SEARCH_OUT] = {};
                    uint32_t held[RD_SEARCH_OUT];
                    bool have_held = false;
#pragma unroll
                    for (int k = 0; k < P_; k++) {
                        const int wj = (k * S_) >> 5, sh = (k * S_) & 31;
                        const bool one = (PRE_ >> k) & 1;
#pragma unroll
                        for (int o = 0; o < RD_SEARCH_OUT; o++) {
                            const uint32_t v = sh ? __builtin_amdgcn_alignbit(r[u][o + wj + 1], r[u][o + wj], sh) : r[u][o + wj];
                            if (one) m[o] &= v;
                            else if (have_held) any[o] = any[o] | held[o] | v;
                            else held[o] = v;
                        }
                        if (!one) have_held = !have_held;
                    }
#pragma unroll
                    for (int o = 0; o < RD_SEARCH_OUT; o++) {
                        if (have_held) any[o] |= held[o];
                        m[o] &= ~any[o];
                    }
                } else {
                    for (int k = 0; k < cfg.P; k++) {
                        const uint32_t x = ((cfg.pre_mask >> k) & 1) ? 0u : 0xFFFFFFFFu;
#pragma unroll
                        for (int o = 0; o < RD_SEARCH_OUT; o++)
                            m[o] &= rd_bits32_at(w, nwords, (long)p0 + 32 * o + (long)k * cfg.S) ^ x;
                    }
                }
                // keep positions inside [p_lo, p_hi]: only the first and the last words of a stream
                if (p0 < plo || p0 + 32 * RD_SEARCH_OUT - 1 > phi) {
#pragma unroll
                    for (int o = 0; o < RD_SEARCH_OUT; o++) {
                        const int q0 = p0 + 32 * o;
                        if (q0 < plo) m[o] &= (plo - q0 >= 32) ? 0u : (0xFFFFFFFFu << (plo - q0));
                        if (q0 + 31 > phi) m[o] &= (phi < q0) ? 0u : (0xFFFFFFFFu >> (31 - (phi - q0)));
                    }
                }
            }
#pragma unroll
            for (int o = 0; o < RD_SEARCH_OUT; o++) {
                uint32_t mm = m[o];
                uint64_t any = __ballot(mm != 0);
                while (any) {  // each round, every lane with matches left contributes its lowest one
                    const uint32_t nf = (uint32_t)__popcll(any);
                    if (npend + nf > RD_MATCH_PEND) {
                        rd_flush_matches(pend, npend, matches, match_cap, counters, lane);
                        npend = 0;
                    }
                    if (mm) {
                        const int bpos = __builtin_ctz(mm);
                        mm &= mm - 1;
                        rd_match e;
                        e.stream = (int32_t)s;
                        e.pos = (int32_t)(p0 + 32 * o + bpos);
                        pend[npend + __builtin_amdgcn_mbcnt_hi((uint32_t)(any >> 32),
                                                               __builtin_amdgcn_mbcnt_lo((uint32_t)any, 0))] = e;
                    }
                    npend += nf;
                    any = __ballot(mm != 0);
                }
            }
        }
    }
    // End of kernel: the workgroup's leftovers leave through ONE atomic (every wave flushing
    // its own few entries made 8192 serialized atomics = 90 us of a 110 us kernel).  What is
    // left is issue-bound: 64 funnel shifts (v_alignbit_b32, 4-cycle class) per 128 positions.
    __shared__ uint32_t left[RD_SEARCH_WAVES + 1];
    if (lane == 0) left[threadIdx.x >> 6] = npend;
    __syncthreads();
    if ((threadIdx.x >> 6) == 0) {
        // exclusive prefix of the wave counts (lane v < RD_SEARCH_WAVES holds wave v's)
        const uint32_t mine = lane < RD_SEARCH_WAVES ? left[lane] : 0u;
        uint32_t incl = mine;
#pragma unroll
        for (int o = 1; o < RD_SEARCH_WAVES; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o, 64);
            if (lane >= o) incl += up;
        }
        const uint32_t tot = __shfl(incl, RD_SEARCH_WAVES - 1, 64);
        if (tot) {
            uint32_t base_slot = 0;
            if (lane == 0) base_slot = atomicAdd(&counters[RD_CNT_MATCH], tot);
            base_slot = __builtin_amdgcn_readfirstlane(base_slot);
#pragma unroll 1
            for (int wv = 0; wv < RD_SEARCH_WAVES; wv++) {
                const uint32_t n = __shfl(mine, wv, 64), off = __shfl(incl - mine, wv, 64);
                for (uint32_t i = lane; i < n; i += 64)
                    if (base_slot + off + i < match_cap) matches[base_slot + off + i] = pend_all[wv][i];
            }
        }
    }
}

void rd_launch_search(const uint32_t *bits, size_t bits_stride, int n_streams, long n_bits, long p_lo, long p_hi,
                      const rd_devcfg &cfg, rd_match *matches, uint32_t match_cap, uint32_t *counters,
                      hipStream_t st) {
    if (p_hi < p_lo || n_streams == 0) return;
    const long base = (p_lo >> 5) << 5;  // floor to a word boundary (p_lo may be negative)
    const long nwords = (n_bits + 31) / 32;
    const long groups = (p_hi - base) / (32 * RD_SEARCH_OUT) + 1;
    const uint64_t total = (uint64_t)n_streams * groups;
    uint64_t wgs = (total + 64 * RD_SEARCH_WAVES - 1) / (64 * RD_SEARCH_WAVES);
    const int cap = rd_k_get_params().search_cap;  // 8 waves per SIMD unless RD_K2_WGS_PER_CU says otherwise
    if (wgs > 256ull * cap) wgs = 256ull * cap;
    // the Davis configuration (protocol.py:68-76): 14 samples/symbol, preamble 1100101110001001
    // (bit m of the mask = symbol m -> 0x91D3)
    if (cfg.S == 14 && cfg.P == 16 && cfg.pre_mask == 0x91D3ull)
        hipLaunchKernelGGL((k_search<14, 16, 0x91D3ull>), dim3((unsigned)wgs), dim3(64 * RD_SEARCH_WAVES), 0, st, bits, bits_stride,
                           n_streams, nwords, base, groups, p_lo, p_hi, cfg, matches, match_cap, counters);
    else
        hipLaunchKernelGGL((k_search<0, 0, 0>), dim3((unsigned)wgs), dim3(64 * RD_SEARCH_WAVES), 0, st, bits, bits_stride, n_streams,
                           nwords, base, groups, p_lo, p_hi, cfg, matches, match_cap, counters);
}


// ------------------------------------------------------------------------------------------
// k_slice_rssi: one WAVE per match.  Decides which call(s) report it (py:194), packs
// packet_symbols bits at stride S (py:197-200: lane k reads symbol k, a ballot packs them) and
// evaluates the reference's RSSI/SNR windows (py:207-236) with all 64 lanes.
// Record i belongs to match i.  A position on a block boundary is reported by two calls (q = B in
// call b, q = 0 in call b+1): its second record is appended at recs[match_cap + k], k counted by
// RD_CNT_REC.  A match that no call reports leaves stream = -1.  Per-call duplicates (py:203-205)
// are dropped by the host when it puts the records in the reference's order.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t rd_bit_at(const uint32_t *w, long nwords, long o) {
    return (rd_word_at(w, nwords, o >> 5) >> (o & 31)) & 1u;
}

__device__ __forceinline__ double rd_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;  // valid in lane 0
}

// filtered[j] = f[origin + j - 1], j in [0, B] (py:133,161: newest block only); origin is
// call*B in batch mode and 0 (the newest block's first sample) in streaming mode.
// rssi / snr of the packet at window index q: valid in lane 0.
template <class View>
__device__ __forceinline__ void rd_rssi_f64(const View &v, long origin, const rd_devcfg &cfg, long q, int lane,
                                            double &rssi, double &snr) {
    const long ns = q - cfg.PL < 0 ? 0 : q - cfg.PL;
    const long pe = q + cfg.PL > cfg.B + 1 ? cfg.B + 1 : q + cfg.PL;
    double noise = 0.0, sig = 0.0;
    for (long j = ns + lane; j < pe; j += 64) {
        const rd_d2 f = rd_f_f64(v, origin + j - 1);
        const double p = f.x * f.x + f.y * f.y;
        if (j < q) noise += p; else sig += p;
    }
    noise = rd_wave_sum(noise);
    sig = rd_wave_sum(sig);
    if (lane == 0) {
        const double noise_power = (q > ns) ? noise / (double)(q - ns) : 1e-9;
        const double signal_power = (pe > q) ? sig / (double)(pe - q) : __builtin_nan("");
        rssi = signal_power > 0 ? 10.0 * log10(signal_power) : -120.0;
        snr = noise_power > 0 ? 10.0 * log10(signal_power / noise_power) : 50.0;
    }
}

// uint8 input: the window means are evaluated in fp32 (|f|^2 to ~1e-6 relative, 4e-6 dB; the
// tolerance on RSSI/SNR is 1e-3 dB).  Each lane takes a contiguous slice of the window so the
// nine-sample FIR history slides through registers (15 conversions for 7 outputs).
// One pass of the RSSI window for a wave: lane handles PER outputs from window index j0.
// PH0 = phase (index mod 4) of the first sample it reads - wave-uniform because every lane's
// start differs by 8 samples - so the Fs/4 rotation and the dword/half selection are static.
template <int PH0, int PER>
__device__ __forceinline__ void rd_rssi_pass(const rd_stream_view &v, int n0, int j0, int q, int pe,
                                             float &noise, float &sig) {
    // (32-bit indices: samples of one stream; 64-bit index arithmetic doubled this kernel's instructions)
    const int vfrom = (int)v.valid_from, vn = (int)v.n;
    const float c[9] = {(float)RD_C0, (float)RD_C1, (float)RD_C2, (float)RD_C3, (float)RD_C4,
                        (float)RD_C3, (float)RD_C2, (float)RD_C1, (float)RD_C0};
    constexpr int ODD = PH0 & 1;
    constexpr int ND = (PER + 8 + ODD + 1) / 2;  // aligned dwords (two samples each) covering PER+8 samples
    const int ne = n0 - ODD;
    const uint32_t *src = (const uint32_t *)(v.base + 2 * (long)ne);
    uint32_t dw[ND];
#pragma unroll
    for (int d = 0; d < ND; d++) {
        const int n = ne + 2 * d;
        dw[d] = (n + 1 >= vfrom && n < vn + 8) ? src[d] : 0u;
    }
    float yr[PER + 8], yi[PER + 8];
#pragma unroll
    for (int k = 0; k < PER + 8; k++) {
        constexpr int dummy = 0; (void)dummy;
        const int i = k + ODD;
        const uint32_t h = dw[i >> 1] >> (16 * (i & 1));
        float a = ((float)(h & 0xFF) - 127.4f) * (1.0f / 127.6f);
        float b = ((float)((h >> 8) & 0xFF) - 127.4f) * (1.0f / 127.6f);
        if (n0 + k < vfrom) { a = 0.0f; b = 0.0f; }
        const int ph = (PH0 + k) & 3;  // static
        yr[k] = ph == 0 ? a : ph == 1 ? -b : ph == 2 ? -a : b;
        yi[k] = ph == 0 ? b : ph == 1 ? a : ph == 2 ? -b : -a;
    }
#pragma unroll
    for (int r = 0; r < PER; r++) {
        float fr = 0.0f, fi = 0.0f;
#pragma unroll
        for (int m = 0; m < 9; m++) {  // f[t] = sum_m c_m y[t-9+m]; y[t-9+m] is window index r+m
            fr = __builtin_fmaf(c[m], yr[r + m], fr);
            fi = __builtin_fmaf(c[m], yi[r + m], fi);
        }
        const float pw = fr * fr + fi * fi;
        const int j = j0 + r;
        if (j < pe) { if (j < q) noise += pw; else sig += pw; }
    }
}

__device__ __forceinline__ float rd_wave_sum_f32(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;  // valid in lane 0
}

// The RSSI windows on the matrix pipe: the 448 filter outputs of a packet's two windows are ONE 16-output block of
// 32 columns in rd_mfma.h's formulation (six MFMAs, exact; rd_demod_mfma.hip: rd_mf_block) instead of 144 fp32
// fma per lane.  Column n covers outputs A + 16 n + 1 .. A + 16 n + 16, lane (n, h) its half h; A is the
// window's first output rounded down to a multiple of four samples (8-byte aligned loads).  Applies when every
// sample the 32 columns touch is a real byte of the stream; otherwise the caller's fp32 path runs.
typedef _Float16 rd_k_h8 __attribute__((ext_vector_type(8)));
typedef float rd_k_f16v __attribute__((ext_vector_type(16)));
__device__ const rd_mf_taps g_rssi_taps = rd_mf_make_taps();

__device__ __forceinline__ rd_k_h8 rd_k_frag(uint2 d) {  // as rd_mf_frag: a byte read as an f16 pattern is k * 2^-24
    uint4 v;
    v.x = d.x & 0x00FF00FFu;
    v.y = __builtin_amdgcn_perm(0u, d.x, 0x0c030c01u);
    v.z = d.y & 0x00FF00FFu;
    v.w = __builtin_amdgcn_perm(0u, d.y, 0x0c030c01u);
    return __builtin_bit_cast(rd_k_h8, v);
}

// One RSSI job prepared for the matrix pipe: ok = every sample the 32 columns touch is a real byte of the stream
// (wave-uniform); p = this lane's first 8 window bytes.
struct rd_rssi_job {
    bool ok;
    int A, ns, pe, q, origin;
    const uint8_t *p;
};
__device__ __forceinline__ rd_rssi_job rd_rssi_prepare(const rd_stream_view &v, int origin, const rd_devcfg &cfg, int q, int lane) {
    rd_rssi_job j;
    j.origin = origin; j.q = q;
    j.ns = q - cfg.PL < 0 ? 0 : q - cfg.PL;
    j.pe = q + cfg.PL > cfg.B + 1 ? cfg.B + 1 : q + cfg.PL;
    const int t_lo = origin + j.ns - 1, t_hi = origin + j.pe - 2;  // filter outputs f[t] of the window
    j.A = (t_lo - 1) & ~3;
    j.ok = t_lo >= 1 && j.A - 8 >= (int)v.valid_from && j.A + 16 * 31 + 24 <= (int)v.n + 8 && t_hi <= j.A + 512;
    j.p = v.base + 2 * (long)(j.A - 8) + 32 * (lane & 31) + 8 * (lane >> 5);
    return j;
}
struct rd_rssi_data {
    uint2 d0, d1, d2;
};
__device__ __forceinline__ rd_rssi_data rd_rssi_fetch(const rd_rssi_job &j) {
    rd_rssi_data d;
    d.d0 = *(const uint2 *)j.p; d.d1 = *(const uint2 *)(j.p + 16); d.d2 = *(const uint2 *)(j.p + 32);
    return d;
}
// window sums of |f|^2 (noise: window index < q, signal: the rest) from the fetched bytes
__device__ __forceinline__ void rd_rssi_block(const rd_rssi_job &j, const rd_rssi_data &d, const rd_k_h8 (&Ahi)[3],
                                              const rd_k_h8 (&Alo)[3], int lane, float &noise, float &sig) {
    const int n = lane & 31, h = lane >> 5;
    const float dcv = -(float)RD_MF_DHI / 16777216.0f;
    const rd_k_f16v dc = {dcv, dcv, dcv, dcv, dcv, dcv, dcv, dcv, dcv, dcv, dcv, dcv, dcv, dcv, dcv, dcv};
    const rd_k_f16v zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const rd_k_h8 b0 = rd_k_frag(d.d0), b1 = rd_k_frag(d.d1), b2 = rd_k_frag(d.d2);
    rd_k_f16v ah = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ahi[0], b0, dc, 0, 0, 0);
    rd_k_f16v al = __builtin_amdgcn_mfma_f32_32x32x16_f16(Alo[0], b0, zero, 0, 0, 0);
    ah = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ahi[1], b1, ah, 0, 0, 0);
    al = __builtin_amdgcn_mfma_f32_32x32x16_f16(Alo[1], b1, al, 0, 0, 0);
    ah = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ahi[2], b2, ah, 0, 0, 0);
    al = __builtin_amdgcn_mfma_f32_32x32x16_f16(Alo[2], b2, al, 0, 0, 0);
    // |f|^2 = |g|^2 / (127.6 * 2^-24 S)^2: g is the filter output in rd_mfma.h's units, the rotation has modulus 1
    const float inv = (float)(1.0 / (127.6 * RD_MF_G_PER_BYTE * 127.6 * RD_MF_G_PER_BYTE));
    float sn = 0.0f, ss = 0.0f;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const float gr = __builtin_fmaf(ah[2 * r], 2048.0f, al[2 * r]), gi = __builtin_fmaf(ah[2 * r + 1], 2048.0f, al[2 * r + 1]);
        const float pw = __builtin_fmaf(gr, gr, gi * gi);
        const int jj = j.A + 16 * n + 8 * h + 1 + r - j.origin + 1;  // window index of output t
        if (jj >= j.ns && jj < j.pe) { if (jj < j.q) sn += pw; else ss += pw; }
    }
    noise = sn * inv;
    sig = ss * inv;
}

// Sum over the wave without the LDS crossbar: four rotations inside the rows of 16 lanes (DPP), then the four row
// sums through scalar registers - a dozen short-latency instructions instead of six dependent ds_bpermute.
__device__ __forceinline__ float rd_wave_sum_dpp(float v) {
#define RD_ROR(x, n) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (x)), 0x120 + (n), 0xF, 0xF, false))
    v += RD_ROR(v, 1);
    v += RD_ROR(v, 2);
    v += RD_ROR(v, 4);
    v += RD_ROR(v, 8);
#undef RD_ROR
    const int vi = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 16)) +
           __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 48));
}

// per-lane partial sums -> py:207-236's two figures (valid in lane 0)
__device__ __forceinline__ void rd_rssi_finish(float noise, float sig, int ns, int pe, int q, int lane, double &rssi, double &snr) {
    noise = rd_wave_sum_dpp(noise);
    sig = rd_wave_sum_dpp(sig);
    if (lane == 0) {
        const float noise_power = (q > ns) ? noise / (float)(q - ns) : 1e-9f;
        const float signal_power = (pe > q) ? sig / (float)(pe - q) : __builtin_nanf("");
        // 10*log10(x) = 3.0102999566 * log2(x)
        rssi = signal_power > 0 ? (double)(3.0102999566f * __builtin_amdgcn_logf(signal_power)) : -120.0;
        snr = noise_power > 0 ? (double)(3.0102999566f * __builtin_amdgcn_logf(signal_power / noise_power)) : 50.0;
    }
}

// uint8 input: the window means are evaluated in fp32 (|f|^2 to ~1e-6 relative; 10*log10 through
// v_log_f32, ~3e-6 dB; the tolerance on RSSI/SNR is 1e-3 dB).
__device__ __forceinline__ void rd_rssi_u8(const rd_stream_view &v, long origin64, const rd_devcfg &cfg, long q64,
                                           int lane, double &rssi, double &snr) {
    const int origin = (int)origin64, q = (int)q64;
    const int ns = q - cfg.PL < 0 ? 0 : q - cfg.PL;
    const int pe = q + cfg.PL > cfg.B + 1 ? cfg.B + 1 : q + cfg.PL;
    constexpr int PER = 8;  // outputs per lane per pass: 64 * 8 = 512 window positions per pass
    float noise = 0.0f, sig = 0.0f;
    for (int jb = ns; jb < pe; jb += 64 * PER) {
        const int j0 = jb + PER * lane;
        // outputs j0 .. j0+PER-1 are f[t], t = origin + j - 1, each using y[t-9 .. t-1]:
        // samples n0 .. n0 + PER + 7 with n0 = origin + j0 - 10
        const int n0 = origin + j0 - 10;
        const int ph0 = __builtin_amdgcn_readfirstlane(n0 & 3);
        if (j0 < pe) {
            switch (ph0) {
                case 0: rd_rssi_pass<0, PER>(v, n0, j0, q, pe, noise, sig); break;
                case 1: rd_rssi_pass<1, PER>(v, n0, j0, q, pe, noise, sig); break;
                case 2: rd_rssi_pass<2, PER>(v, n0, j0, q, pe, noise, sig); break;
                default: rd_rssi_pass<3, PER>(v, n0, j0, q, pe, noise, sig); break;
            }
        }
    }
    rd_rssi_finish(noise, sig, ns, pe, q, lane, rssi, snr);
}

#ifndef RD_SLICE_MIN_WGS
#define RD_SLICE_MIN_WGS 1  // (8 = 64 VGPRs was tried: 14 spills, 73 us instead of 58)
#endif
// where the RSSI windows read their samples from: the uint8 streams or the complex128 ring
struct rd_u8_src {
    rd_layout lay;
    __device__ __forceinline__ void rssi(int stream, long origin, const rd_devcfg &cfg, long q, int lane, double &r,
                                         double &s) const {
        rd_stream_view v;
        v.base = lay.iq + (size_t)stream * lay.stream_stride;
        v.valid_from = lay.valid_from;
        v.n = lay.n_samples;
        rd_rssi_u8(v, origin, cfg, q, lane, r, s);
    }
};
struct rd_cplx_src {
    rd_cplx_view v;
    __device__ __forceinline__ void rssi(int, long origin, const rd_devcfg &cfg, long q, int lane, double &r,
                                         double &s) const {
        rd_rssi_f64(v, origin, cfg, q, lane, r, s);
    }
};

// Lanes 0..31 hold data[lane] in `byte`, lane 0 holds rssi / snr.  Destinations: `dev` (device
// memory) and/or `host` (pinned host memory mapped into the device, used by the streaming handle:
// a block's few records need no copy afterwards).  Plain byte + header stores: assembling the 64
// bytes across 16 lanes for one coalesced store (four ds_bpermute + readfirstlanes) was measured
// 7 us slower per launch.
__device__ __forceinline__ void rd_store_record(rd_packet *dev, rd_packet *host, int lane, int stream, long call, long q,
                                                int nbytes, uint32_t byte, double rssi, double snr) {
#pragma unroll
    for (int dst = 0; dst < 2; dst++) {
        rd_packet *o = dst ? host : dev;
        if (!o) continue;
        if (lane < RD_MAX_PKT_BYTES) o->data[lane] = lane < nbytes ? (uint8_t)byte : (uint8_t)0;
        if (lane == 0) {
            o->stream = stream; o->call = (int32_t)call; o->index = (int32_t)q; o->nbytes = nbytes;
            o->rssi = rssi; o->snr = snr;
        }
    }
}

// a match that no call reports, or that the per-call dedupe is certain to drop: stream = -1
__device__ __forceinline__ void rd_store_void(rd_packet *dev, rd_packet *host, int lane) {
    if (lane == 0) {
        if (dev) dev->stream = -1;
        if (host) host->stream = -1;
    }
}

template <class Src>
__global__ __launch_bounds__(256, RD_SLICE_MIN_WGS) void k_slice_rssi(Src src, const uint32_t *bits, size_t bits_stride, long nwords,
                                                    rd_devcfg cfg, const rd_match *matches, uint32_t match_cap,
                                                    int batch_mode, int n_calls, int call, rd_packet *recs,
                                                    rd_packet *recs_host, uint32_t *counters) {
    const int lane = threadIdx.x & 63;
    uint32_t count = counters[RD_CNT_MATCH];
    if (count > match_cap) count = match_cap;
    const uint32_t nw = gridDim.x * (blockDim.x >> 6);
    for (uint32_t i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); i < count; i += nw) {
        const int stream = __builtin_amdgcn_readfirstlane(matches[i].stream);
        const long pos = __builtin_amdgcn_readfirstlane(matches[i].pos);
        // Which call(s) report this position (py:194, q <= B): in batch mode call b sees absolute
        // positions w_b <= p <= w_b + B with w_b = (b+1)B - L, so p is reported by one call, or by
        // two when it falls on a block boundary (q = B in call b, q = 0 in call b+1).
        long b0 = call, b1 = -1, q0 = pos, q1 = 0;
        bool ok0 = true, ok1 = false;
        if (batch_mode) {
            // p + L >= B because p >= B - L, and < 2^32 (n_samples < 2^31): 32-bit division
            const uint32_t pl = (uint32_t)(pos + cfg.L), bq = pl / (uint32_t)cfg.B, br = pl - bq * (uint32_t)cfg.B;
            b0 = (long)bq - 1;
            q0 = pos - ((b0 + 1) * (long)cfg.B - cfg.L);
            ok0 = b0 >= 0 && b0 < n_calls;
            b1 = b0 - 1;
            q1 = q0 + cfg.B;
            ok1 = br == 0 && b1 >= 0 && b1 < n_calls;
        }
        // symbols 64r + lane of the packet; byte bi = symbols 8bi .. 8bi+7, first symbol = MSB; a
        // last partial byte is right-aligned (the bits are shifted in one by one, py:197-200).
        // The same fetch yields the symbols of the packets one position earlier and later: when
        // one of them carries the same bits (an oversampled burst matches at 2-4 adjacent
        // positions) and precedes this one in the reference's order within the same call, this
        // record is certain to be dropped by the per-call dedupe (py:203-205) and is skipped here,
        // before its RSSI windows are read.
        const uint32_t *w = bits + (size_t)stream * bits_stride;
        uint32_t byte = 0;
        bool same_prev = true, same_next = true;
        for (int r = 0; r * 64 < cfg.K; r++) {
            const int k = 64 * r + lane;
            const uint32_t tri = k < cfg.K ? rd_bits32_at_i(w, (int)nwords, (int)pos - 1 + k * cfg.S) : 0u;
            const uint64_t mp = __ballot((tri & 1u) != 0), m = __ballot((tri & 2u) != 0), mn = __ballot((tri & 4u) != 0);
            same_prev &= mp == m;
            same_next &= mn == m;
            const int bi = lane - 8 * r;
            if (bi >= 0 && bi < 8) {
                const int have = cfg.K - 8 * lane;  // symbols in this byte
                const uint32_t rev = __builtin_bitreverse32((uint32_t)((m >> (8 * bi)) & 0xFF)) >> 24;
                byte = have >= 8 ? rev : have > 0 ? rev >> (8 - have) : 0u;
            }
        }
        // The reference's order inside a call is by (q % S, q) (py:171-188, phase-major search):
        // q-1 precedes q unless its phase is larger (it wraps to S-1 when q % S == 0, S > 1);
        // q+1 precedes q only when its phase is smaller (it wraps to 0 when q % S == S-1, S > 1).
        auto superseded = [&](long q) {
            const uint32_t S = (uint32_t)cfg.S, ph = (uint32_t)q % S;
            const bool prev_first = S == 1 || ph != 0;
            const bool next_first = S > 1 && ph == S - 1;
            return (same_prev && q >= 1 && prev_first) || (same_next && q + 1 <= cfg.B && next_first);
        };
        rd_packet *o = recs ? &recs[i] : nullptr, *oh = recs_host ? &recs_host[i] : nullptr;
        const bool use0 = ok0 && !superseded(q0), use1 = ok1 && !superseded(q1);
        if (!(use0 || use1)) {
            rd_store_void(o, oh, lane);
            continue;
        }
        const long pb = use0 ? b0 : b1, pq = use0 ? q0 : q1;
        double rssi = 0.0, snr = 0.0;
        src.rssi(stream, batch_mode ? pb * cfg.B : 0, cfg, pq, lane, rssi, snr);
        rd_store_record(o, oh, lane, stream, pb, pq, cfg.nbytes, byte, rssi, snr);
        if (use0 && use1) {
            uint32_t slot = 0;
            if (lane == 0) slot = atomicAdd(&counters[RD_CNT_REC], 1u);
            slot = __builtin_amdgcn_readfirstlane(slot);  // < match_cap: at most one per match
            src.rssi(stream, b1 * cfg.B, cfg, q1, lane, rssi, snr);
            const size_t xi = (size_t)match_cap + slot;
            rd_store_record(recs ? &recs[xi] : nullptr, recs_host ? &recs_host[xi] : nullptr, lane, stream, b1, q1,
                            cfg.nbytes, byte, rssi, snr);
        }
    }
}

__device__ __forceinline__ uint32_t rd_wave_incl_scan_u32(uint32_t v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(v, o, 64);
        if (lane >= o) v += up;
    }
    return v;
}

// ------------------------------------------------------------------------------------------
// The same work in two kernels for a compile-time (S, K) and the batch path: k_classify takes one LANE per match -
// the wave-per-match kernel above spends ~80 instructions and a chain of three memory latencies on a match that
// two times out of three is voided - and k_rssi one wave per SURVIVING packet (its two RSSI windows are 448 filter
// outputs).  A lane reads the words that hold its packet's K symbols for positions pos-1, pos, pos+1, shifts them
// to a common origin, and then everything is static: symbol k sits at bit k S + 1.
// ------------------------------------------------------------------------------------------
struct rd_task {
    uint32_t rec;     // record index (primary slot, or match_cap + extra slot)
    int32_t stream;
    int32_t call;     // reported by this call ...
    int32_t q;        // ... at this window index
};

template <int S_, int K_>
__global__ __launch_bounds__(256) void k_classify(const uint32_t *bits, size_t bits_stride, int nwords, rd_devcfg cfg,
                                                  const rd_match *matches, uint32_t match_cap, int n_calls,
                                                  rd_packet *recs, rd_task *tasks, uint32_t *counters) {
    constexpr int NBITS = (K_ - 1) * S_ + 3;          // bits pos-1 .. of the three packets
    constexpr int NU = (NBITS + 31) / 32;              // words of the aligned stream
    constexpr int NW = NU + 1;                         // words fetched (funnel)
    constexpr int NBYTES = (K_ + 7) / 8;
    static_assert(K_ % 8 == 0 && NBYTES <= RD_MAX_PKT_BYTES, "whole bytes only");
    __shared__ uint32_t s_tot[4], s_base;
    const int lane = threadIdx.x & 63;
    uint32_t count = counters[RD_CNT_MATCH];
    if (count > match_cap) count = match_cap;
    const uint32_t nthreads = gridDim.x * blockDim.x;
    // trips are uniform over the WORKGROUP (barriers inside): the workgroup's first index decides
    for (uint32_t g0 = blockIdx.x * blockDim.x; g0 < count; g0 += nthreads) {
        const uint32_t i = g0 + threadIdx.x;
        const bool live = i < count;
        int stream = 0, pos = 0;
        if (live) { stream = matches[i].stream; pos = matches[i].pos; }
        // calls that report this position (py:194, q <= B), as in k_slice_rssi
        const uint32_t pl = (uint32_t)(pos + cfg.L), bq = pl / (uint32_t)cfg.B, br = pl - bq * (uint32_t)cfg.B;
        const int b0 = (int)bq - 1;
        const int q0 = pos - ((b0 + 1) * cfg.B - cfg.L);
        const bool ok0 = live && b0 >= 0 && b0 < n_calls;
        const int b1 = b0 - 1, q1 = q0 + cfg.B;
        const bool ok1 = live && br == 0 && b1 >= 0 && b1 < n_calls;
        // words holding bits pos-1 ..: zero outside the array (rd_bits32_at_i)
        const uint32_t *w = bits + (size_t)stream * bits_stride;
        const int o = pos - 1, wi0 = o >> 5;
        const uint32_t sh = (uint32_t)(o & 31);
        uint32_t W[NW];
#pragma unroll
        for (int j = 0; j < NW; j++) {
            const int wi = wi0 + j;
            W[j] = (live && wi >= 0 && wi < nwords) ? w[wi] : 0u;
        }
        uint32_t u[NU];
#pragma unroll
        for (int j = 0; j < NU; j++) u[j] = __builtin_amdgcn_alignbit(W[j + 1], W[j], sh);
        // symbol k of position pos-1 / pos / pos+1 = bit k S / k S + 1 / k S + 2 of u.
        // y = u ^ (u >> 1): bit kS set <=> prev differs from cur, bit kS + 1 set <=> cur differs from next
        uint32_t dprev = 0, dnext = 0;
#pragma unroll
        for (int j = 0; j < NU; j++) {
            const uint32_t nxt = j + 1 < NU ? u[j + 1] : 0u;
            const uint32_t y = u[j] ^ __builtin_amdgcn_alignbit(nxt, u[j], 1);
            uint32_t mp = 0, mn = 0;  // static masks of this word
#pragma unroll
            for (int k = 0; k < K_; k++) {
                if (((k * S_) >> 5) == j) mp |= 1u << ((k * S_) & 31);
                if (((k * S_ + 1) >> 5) == j) mn |= 1u << ((k * S_ + 1) & 31);
            }
            dprev |= y & mp;
            dnext |= y & mn;
        }
        const bool same_prev = dprev == 0, same_next = dnext == 0;
        // the packet's bytes: byte bi = symbols 8 bi .. 8 bi + 7, first symbol = MSB (py:197-200)
        uint32_t dw[(NBYTES + 3) / 4];
#pragma unroll
        for (int m = 0; m < (NBYTES + 3) / 4; m++) dw[m] = 0;
#pragma unroll
        for (int k = 0; k < K_; k++) {
            const int bit = k * S_ + 1, bi = k >> 3;
            const uint32_t b = (u[bit >> 5] >> (bit & 31)) & 1u;
            dw[bi >> 2] |= b << (8 * (bi & 3) + 7 - (k & 7));
        }
        auto superseded = [&](int q) {
            const uint32_t S = (uint32_t)S_, ph = (uint32_t)q % S;
            const bool prev_first = S == 1 || ph != 0;
            const bool next_first = S > 1 && ph == S - 1;
            return (same_prev && q >= 1 && prev_first) || (same_next && q + 1 <= cfg.B && next_first);
        };
        const bool use0 = ok0 && !superseded(q0), use1 = ok1 && !superseded(q1);
        const int ntask = (use0 ? 1 : 0) + (use1 ? 1 : 0);  // records of this match: task t owns record t (dense)
        // task slots: ONE atomic per workgroup and trip (a counter word sustains ~90 atomics per microsecond: one per
        // wave was 12 of this kernel's 20 us).  All waves of a workgroup make the same number of trips.
        const uint32_t incl = rd_wave_incl_scan_u32((uint32_t)ntask, lane);
        const uint32_t tot = __shfl(incl, 63, 64);
        const int wv = threadIdx.x >> 6;
        __syncthreads();  // the previous trip's s_tot / s_base have been read
        if (lane == 0) s_tot[wv] = tot;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t sum = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) sum += s_tot[k];
            s_base = sum ? atomicAdd(&counters[RD_CNT_TASKS], sum) : 0u;
        }
        __syncthreads();
        uint32_t base = s_base;
#pragma unroll
        for (int k = 0; k < 4; k++) base += k < wv ? s_tot[k] : 0u;
        base = __builtin_amdgcn_readfirstlane(base);
        if (ntask) {
            uint32_t t = base + incl - (uint32_t)ntask;
            const int pb = use0 ? b0 : b1, pq = use0 ? q0 : q1;
            auto put = [&](uint32_t ri, int call, int q) {
                uint32_t *r = (uint32_t *)&recs[ri];
                const uint4 hdr = {(uint32_t)stream, (uint32_t)call, (uint32_t)q, (uint32_t)NBYTES};
                *(uint4 *)r = hdr;
                uint32_t d8[RD_MAX_PKT_BYTES / 4];
#pragma unroll
                for (int m = 0; m < RD_MAX_PKT_BYTES / 4; m++) d8[m] = m < (NBYTES + 3) / 4 ? dw[m] : 0u;
#pragma unroll
                for (int m = 0; m < RD_MAX_PKT_BYTES / 16; m++)
                    *(uint4 *)(r + 4 + 4 * m) = uint4{d8[4 * m], d8[4 * m + 1], d8[4 * m + 2], d8[4 * m + 3]};
                const rd_task tk = {ri, stream, call, q};
                tasks[t++] = tk;
            };
            put(t, pb, pq);
            if (use0 && use1) put(t, b1, q1);  // a position on a block boundary: reported by both calls
        }
    }
}

// One wave per surviving packet.  The task entry two steps ahead and the window bytes of the next step are
// fetched while the current packet's block runs on the matrix pipe: the kernel is a chain of latencies otherwise.
__global__ __launch_bounds__(256) void k_rssi_u8(rd_layout lay, rd_devcfg cfg, const rd_task *tasks, uint32_t task_cap,
                                                 rd_packet *recs, const uint32_t *counters) {
    const int lane = threadIdx.x & 63;
    const uint32_t nw = gridDim.x * (blockDim.x >> 6);
    const uint32_t i0 = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    rd_k_h8 Ahi[3], Alo[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        Ahi[d] = *(const rd_k_h8 *)g_rssi_taps.v[0][d][lane];
        Alo[d] = *(const rd_k_h8 *)g_rssi_taps.v[1][d][lane];
    }
    rd_task t_cur = {0, 0, 0, 0}, t_nxt = {0, 0, 0, 0};
    if (i0 < task_cap) t_cur = tasks[i0];                 // (any index below task_cap is readable)
    if (i0 + nw < task_cap) t_nxt = tasks[i0 + nw];
    uint32_t count = counters[RD_CNT_TASKS];
    if (count > task_cap) count = task_cap;
    auto view = [&](int stream) {
        rd_stream_view v;
        v.base = lay.iq + (size_t)stream * lay.stream_stride;
        v.valid_from = lay.valid_from;
        v.n = lay.n_samples;
        return v;
    };
    auto job_of = [&](const rd_task &t) {
        const int stream = __builtin_amdgcn_readfirstlane(t.stream), call = __builtin_amdgcn_readfirstlane(t.call),
                  q = __builtin_amdgcn_readfirstlane(t.q);
        return rd_rssi_prepare(view(stream), call * cfg.B, cfg, q, lane);
    };
    rd_rssi_job j_cur = {};
    rd_rssi_data d_cur = {};
    if (i0 < count) {
        j_cur = job_of(t_cur);
        if (j_cur.ok) d_cur = rd_rssi_fetch(j_cur);
    }
    for (uint32_t i = i0; i < count; i += nw) {
        const rd_task t_now = t_cur;
        const rd_rssi_job j_now = j_cur;
        const rd_rssi_data d_now = d_cur;
        // next step's bytes, the entry after that
        if (i + nw < count) {
            j_cur = job_of(t_nxt);
            if (j_cur.ok) d_cur = rd_rssi_fetch(j_cur);
        }
        t_cur = t_nxt;
        if (i + 2 * nw < task_cap) t_nxt = tasks[i + 2 * nw];
        const uint32_t rec = __builtin_amdgcn_readfirstlane(t_now.rec);
        double rssi = 0.0, snr = 0.0;
        if (j_now.ok) {
            float noise, sig;
            rd_rssi_block(j_now, d_now, Ahi, Alo, lane, noise, sig);
            rd_rssi_finish(noise, sig, j_now.ns, j_now.pe, j_now.q, lane, rssi, snr);
        } else {  // a window that reaches outside the stream: the fp32 path with its per-sample checks
            const int stream = __builtin_amdgcn_readfirstlane(t_now.stream);
            rd_rssi_u8(view(stream), (long)__builtin_amdgcn_readfirstlane(t_now.call) * cfg.B, cfg,
                       (long)__builtin_amdgcn_readfirstlane(t_now.q), lane, rssi, snr);
        }
        if (lane == 0) { recs[rec].rssi = rssi; recs[rec].snr = snr; }
    }
}

// ------------------------------------------------------------------------------------------
// k_tail (round 4): the WHOLE tail of a batch run in ONE launch (Davis shape).  Until round 3 a run's tail was four
// launches - k_fixup, k_search and two slice / RSSI kernels with the order on the device: 102 us of kernels that are each a launch, a ramp and a
// chain of dependent latencies, plus the gaps between dependent kernels of one stream.  Streams are independent
// (dsp.py:131-135 holds all state per instance), so a workgroup that OWNS RD_FT_STREAMS consecutive streams can take
// them through every stage without waiting for anybody:
//   0. exact bits for the 8-sample groups the demod kernel listed for its streams (k_fixup's arithmetic; the demod
//      kernel's waves put their entries into per-group buckets, rd_demod_mfma.hip: rd_mf_flush_buckets);
//   1. Demodulator._search (py:171-188) over its streams' bits - k_search's funnel-shift test - the matches into
//      per-stream lists in LDS;
//   2. Demodulator._slice (py:190-205): calls (q <= B, py:194), packet bytes (py:197-200), the exact per-call dedupe
//      (py:203-205) and the reference's order (phase-major inside a call, py:175-186), by rank among the stream's
//      survivors; all in LDS (one lane per match, strips of 32 for long streams);
//   3. RSSI / SNR (py:207-236) on the matrix pipe as k_rssi_u8 does, and the finished records at their FINAL position:
//      the number of records of the groups in front comes from a word per group, published (agent scope) as soon as a
//      group knows its total and read - spinning if need be - by the one wave per workgroup that needs it, while the
//      other waves already evaluate windows.  A group only ever waits for groups with LOWER numbers, workgroups are
//      dispatched in ascending order: the lowest unfinished group never waits, so the chain always advances (and a
//      spin limit turns a bug into an error flag instead of a hang).
// A stream with more matches than its list holds, more records than the output array, or a group whose fix-up bucket
// overflowed raises RD_CNT_OVF bits (1, 2, 8; 16 = the spin limit): the host falls back to the separate kernels.
// ------------------------------------------------------------------------------------------
#define RD_FT_WG 256
#ifndef RD_FT_DEPTH
#define RD_FT_DEPTH 5   /* register buffers of the search phase (units in flight + 1) */
#endif
struct rd_ft_args {
    rd_layout lay;
    rd_devcfg cfg;
    int n_calls;
    int p_hi;                  // last position to report; the first is 0 (PRE_ starts with a one: zero bits in front of a stream match nothing)
    int skip_fix;              // 1: the bits are final already (second pass after an exact re-evaluation of everything)
    const uint32_t *fixb;      // [groups][fix_bcap] entries (word index << 4 | group mask)
    const uint32_t *fixcnt;    // [groups] entries written (more than fix_bcap: the bucket overflowed)
    uint32_t fix_bcap;
    uint32_t bcap;             // matches a stream's list holds (a multiple of 32)
    uint32_t bucket_limit;     // <= bcap (test hook: a smaller limit makes the overflow path run on ordinary inputs)
    uint32_t *gstate;          // [3][groups]: (seq << 20 | records), matches, fix-up entries of each group
    uint32_t seq;              // 1 .. 4095, different from the previous launch's on this buffer
    rd_packet *recs;
    uint32_t rec_cap;
    uint32_t *counters;
    uint32_t *zero_next;       // the handle's next counter set (cleared by workgroup 0) or null
    uint32_t zero_words;
    uint64_t *stamps;          // diagnostic library: [groups][8] s_memrealtime stamps of the phases (else null)
    int abl;                   // diagnostic library, WRONG results: 1 search without the test, 2 without the loads, 3 without either
};
#ifdef RD_DIAG
#define RD_FT_STAMP(k) do { if (a.stamps && (tid & 63) == 0) { uint64_t t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : : "memory"); \
                                                               a.stamps[(size_t)grp * 8 + (k)] = t_; } } while (0)
#else
#define RD_FT_STAMP(k) do { } while (0)
#endif

// The preamble test of one output word with three-input logic (v_bitop3_b32: any boolean function of three operands in
// one instruction): tap t contributes its shifted word or the complement, by the preamble's symbol t, and every
// instruction folds two more taps into the running AND - 8 logic instructions per word for 16 taps, where and / or3 /
// andn chains took 15 (the compiler's best), next to the 15 funnel shifts that produce the taps.
template <int S_, int T>
__device__ __forceinline__ uint32_t rd_tap(const uint32_t *r, int o) {
    constexpr int wj = (T * S_) >> 5, sh = (T * S_) & 31;
    return sh ? __builtin_amdgcn_alignbit(r[o + wj + 1], r[o + wj], sh) : r[o + wj];
}
// truth table of A' & B' & C' (X' = X when the symbol is one, else its complement); operands a, b, c = 0xF0, 0xCC, 0xAA
constexpr int rd_tt3(bool pa, bool pb, bool pc) { return (pa ? 0xF0 : 0x0F) & (pb ? 0xCC : 0x33) & (pc ? 0xAA : 0x55); }
template <int S_, int P_, uint64_t PRE_, int T>
__device__ __forceinline__ uint32_t rd_match_from(uint32_t m, const uint32_t *r, int o) {
    if constexpr (T >= P_) {
        return m;
    } else if constexpr (T + 1 == P_) {
        const uint32_t v = rd_tap<S_, T>(r, o);
        return __builtin_amdgcn_bitop3_b32(m, v, v, rd_tt3(true, (PRE_ >> T) & 1, (PRE_ >> T) & 1));
    } else {
        const uint32_t n = __builtin_amdgcn_bitop3_b32(m, rd_tap<S_, T>(r, o), rd_tap<S_, T + 1>(r, o),
                                                       rd_tt3(true, (PRE_ >> T) & 1, (PRE_ >> (T + 1)) & 1));
        return rd_match_from<S_, P_, PRE_, T + 2>(n, r, o);
    }
}
template <int S_, int P_, uint64_t PRE_>
__device__ __forceinline__ uint32_t rd_match_word(const uint32_t *r, int o) {
    static_assert(P_ >= 3, "three taps in the first instruction");
    const uint32_t m = __builtin_amdgcn_bitop3_b32(rd_tap<S_, 0>(r, o), rd_tap<S_, 1>(r, o), rd_tap<S_, 2>(r, o),
                                                   rd_tt3(PRE_ & 1, (PRE_ >> 1) & 1, (PRE_ >> 2) & 1));
    return rd_match_from<S_, P_, PRE_, 3>(m, r, o);
}

// a workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every global store in flight
// (s_waitcnt vmcnt(0)) - here the stores that publish a group's totals, whose acknowledgement nobody in the
// workgroup needs
__device__ __forceinline__ void rd_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int S_, int P_, uint64_t PRE_, int K_>
__global__ __launch_bounds__(RD_FT_WG, 4) void k_tail(rd_ft_args a) {
    constexpr int G = RD_FT_STREAMS;
    static_assert(PRE_ & 1, "positions below 0 are skipped because the preamble starts with a one");
    static_assert(RD_FT_WG == 64 * G, "one wave per stream in the classify strips");
    extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];
    // s_t: per match {flags, call b0, key 0, key 1, bytes[3], result}; s_pos: the matches' positions; s_ord: the
    // group's surviving tasks in final order (stream << 16 | match << 1 | which); s_res: their RSSI / SNR
    uint32_t *s_t = s_dyn;
    int32_t *s_pos = (int32_t *)(s_dyn + (size_t)G * a.bcap * 8);
    uint32_t *s_ord = s_dyn + (size_t)G * a.bcap * 9;
    __shared__ uint32_t s_cnt[G], s_kept[G];
    __shared__ uint32_t s_ready;   // 0: not yet; else 1 + the number of records of the groups in front
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = blockIdx.x, n_groups = gridDim.x;
    const int stream0 = grp * G;
    const int n_here = a.lay.n_streams - stream0 < G ? a.lay.n_streams - stream0 : G;
    const int nwords = (int)((a.lay.n_samples + 31) / 32);
    if (grp == 0 && a.zero_next)
        for (uint32_t i = tid; i < a.zero_words; i += RD_FT_WG) a.zero_next[i] = 0;
    if (tid < G) { s_cnt[tid] = 0; s_kept[tid] = 0; }
    if (tid == 0) s_ready = 0;
    // The SIMD arbiter serves the oldest wave first: of the four workgroups that share a CU the first dispatched ran its
    // search in 30 us, the last in 48, and every group behind it waited for its total.  Workgroups are dispatched in
    // ascending order, a quarter of the grid per round: quarter q's waves raise and lower their priority in rotation
    // (search unit k runs at priority (q + k) & 3), so that the four waves of a SIMD take turns at the front.
    const int prio_q = (int)(((unsigned)grp * 4u) / (unsigned)n_groups);
    if (wave == 0) RD_FT_STAMP(0);

    // ---- 0. exact bits for the listed groups of this workgroup's streams (k_fixup's listed branch) ----
    uint32_t n_fix_raw = 0;
    if (!a.skip_fix) {
        // One lane per (entry, listed 8-sample group), packed densely: the ~50 items of a group of streams fit ONE wave.
        // (A lane per entry that took its groups one after the other - a stream's first run lists all four - held the
        // workgroup at the barrier for four dependent rounds of loads and float64 arithmetic, 12 us; a lane per
        // (entry, group) slot kept all four waves busy with the arithmetic for a handful of active lanes each.)
        // An entry is asked for together with the count - a slot past the count holds an old run's entry or nothing, and
        // is not used: one memory round trip instead of two.
        const uint32_t *bk = a.fixb + (size_t)grp * a.fix_bcap;
        const uint32_t e_first = (uint32_t)tid < a.fix_bcap ? bk[tid] : 0u;
        n_fix_raw = a.fixcnt[grp];
        const uint32_t n_fix = n_fix_raw < a.fix_bcap ? n_fix_raw : a.fix_bcap;
        uint32_t *s_items = s_t;  // (the match table's space: 4 x 256 items at the least, not in use yet)
        __shared__ uint32_t s_nitems;
        static_assert(RD_GROUPS == 4, "an entry lists up to four groups");
        for (uint32_t i0 = 0; i0 < n_fix; i0 += RD_FT_WG) {
            if (tid == 0) s_nitems = 0;
            __syncthreads();
            const uint32_t i = i0 + (uint32_t)tid;
            const uint32_t e = i < n_fix ? (i0 == 0 ? e_first : bk[i]) : 0u;
            const uint32_t k = (uint32_t)__popc(e & 0xFu);
            if (k) {
                uint32_t at = atomicAdd(&s_nitems, k);
                for (int g = 0; g < RD_GROUPS; g++)
                    if ((e >> g) & 1) s_items[at++] = ((uint32_t)tid << 2) | (uint32_t)g;
                s_items[4 * RD_FT_WG + tid] = e;   // (the entries themselves: 256 words behind the 1024 item slots)
            }
            __syncthreads();
            const uint32_t n_items = s_nitems;
            for (uint32_t it = (uint32_t)tid; it < n_items; it += RD_FT_WG) {
                const uint32_t item = s_items[it];
                const uint32_t ee = s_items[4 * RD_FT_WG + (item >> 2)];
                const int g = (int)(item & 3u);
                const uint32_t widx = ee >> 4;
                const uint32_t s = widx / (uint32_t)a.lay.bits_stride;
                const uint32_t run = widx - s * (uint32_t)a.lay.bits_stride;
                const uint8_t *base = a.lay.iq + (size_t)s * a.lay.stream_stride;
                const long t0 = (long)run * RD_RUN + g * RD_GROUP;
                const long left = (long)a.lay.n_samples - t0;
                if (left <= 0) continue;
                const int count = left < RD_GROUP ? (int)left : RD_GROUP;
                const uint8_t *p = base + 2 * (t0 - 10);
                uint32_t dw[10];
                const long x = a.lay.valid_from - (t0 - 10), z = (long)a.lay.n_samples + 8 - (t0 - 10);
                const int d_lo = x <= 0 ? 0 : x >= 20 ? 10 : (int)(x / 2);
                const int d_hi = z <= 0 ? 0 : z >= 20 ? 10 : (int)((z + 1) / 2);
#pragma unroll
                for (int d = 0; d < 10; d++) dw[d] = (d >= d_lo && d < d_hi) ? *(const uint32_t *)(p + 4 * d) : 0u;
                ((uint8_t *)a.lay.bits)[(size_t)widx * 4 + g] = (uint8_t)rd_exact_group_dw(dw, t0, count, a.lay.valid_from);
            }
            if (i0 + RD_FT_WG < n_fix) __syncthreads();   // (the item list is rewritten by the next batch of entries)
        }
    }
    // (the byte stores above are read below by other waves of this workgroup only: the barrier's workgroup-scope
    // release / acquire covers them - all waves of a workgroup share the CU's vector L1)
    __syncthreads();
    if (wave == 0) RD_FT_STAMP(1);

    // ---- 1. preamble search over this workgroup's streams: a wave = 64 lanes x 128 positions of one stream ----
    // RD_FT_DEPTH register buffers in rotation: the words of the units behind the one being tested are in flight.  (With
    // one unit's loads waited for at the top of every trip this phase took 41 us; with two or four units in flight 38: loads
    // alone take 24.6 us = 5.6 TB/s of bits, the test alone 21.7, and together they add up rather than overlap - the memory
    // path follows the clock the arithmetic pulls down.  profiles/r04_tail_stamps.txt)
    {
        constexpr int NW = ((RD_SEARCH_OUT + ((P_ - 1) * S_ + 31) / 32 + 1 + 3) / 4) * 4;
        const int gps = a.p_hi / (32 * RD_SEARCH_OUT) + 1;       // lane-groups per stream
        const int wgps = (gps + 63) / 64;                         // wave-groups per stream
        const int units = n_here * wgps;
        // (unit u = wave-group u % wgps of stream u / wgps; the split is carried along in scalars - one division per
        // unit was a fifth of this loop's bookkeeping)
        auto fetch = [&](int g, int rem, uint32_t (&r)[NW]) {
            const int gi = rem * 64 + lane;
            const uint32_t *w = a.lay.bits + (size_t)(stream0 + g) * a.lay.bits_stride;
            // Every lane loads, whatever its position: the lanes past the stream's last lane-group (the last wave-group
            // of a stream only) read the stream's last NW words instead and report nothing.  No branch, no exec mask: the
            // compiler's wait-count bookkeeping gave up on the guarded form of this ("wait for everything" in front of
            // every prefetch), and the loads then ran between the tests instead of under them.  (The host checks that a
            // reported position's window always lies inside the stream: p_hi / 32 + NW <= nwords.)
            int w0 = RD_SEARCH_OUT * gi;
            w0 = w0 + NW <= nwords ? w0 : nwords - NW;
#pragma unroll
            for (int j = 0; j < NW / 4; j++) {
                const uint4 v4 = *(const uint4 *)(w + w0 + 4 * j);
                r[4 * j] = v4.x; r[4 * j + 1] = v4.y; r[4 * j + 2] = v4.z; r[4 * j + 3] = v4.w;
            }
        };
        auto test = [&](int g, int rem, const uint32_t (&r)[NW]) {
            const int gi = rem * 64 + lane;
            uint32_t m[RD_SEARCH_OUT] = {};
            const int p0 = 32 * RD_SEARCH_OUT * gi;
            if (gi < gps) {
#pragma unroll
                for (int o = 0; o < RD_SEARCH_OUT; o++) m[o] = rd_match_word<S_, P_, PRE_>(r, o);
                if (p0 + 32 * RD_SEARCH_OUT - 1 > a.p_hi) {  // only a stream's last words
#pragma unroll
                    for (int o = 0; o < RD_SEARCH_OUT; o++) {
                        const int q0 = p0 + 32 * o;
                        if (q0 + 31 > a.p_hi) m[o] &= (a.p_hi < q0) ? 0u : (0xFFFFFFFFu >> (31 - (a.p_hi - q0)));
                    }
                }
            }
            uint32_t anym = 0;
#pragma unroll
            for (int o = 0; o < RD_SEARCH_OUT; o++) anym |= m[o];
            if (__ballot(anym != 0)) {  // wave-uniform; 3 wave-groups in 10 on noise
                uint32_t mine = 0;
#pragma unroll
                for (int o = 0; o < RD_SEARCH_OUT; o++) mine += (uint32_t)__popc(m[o]);
                uint32_t incl = mine;
#pragma unroll
                for (int sh = 1; sh < 64; sh <<= 1) {
                    const uint32_t up = (uint32_t)__shfl_up((int)incl, sh, 64);
                    if (lane >= sh) incl += up;
                }
                const uint32_t tot = (uint32_t)__shfl((int)incl, 63, 64);
                uint32_t slot0 = 0;
                if (lane == 0) slot0 = atomicAdd(&s_cnt[g], tot);
                uint32_t slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot0) + incl - mine;
#pragma unroll
                for (int o = 0; o < RD_SEARCH_OUT; o++) {
                    uint32_t mm = m[o];
                    while (mm) {  // (lane-divergent, a few iterations at most)
                        const int bpos = __builtin_ctz(mm);
                        mm &= mm - 1;
                        if (slot < a.bcap) s_pos[(size_t)g * a.bcap + slot] = p0 + 32 * o + bpos;
                        slot++;
                    }
                }
            }
        };
        // a wave's units: wave, wave + G, ...; (g, rem) of the unit to fetch next and of the unit to test next.  RD_FT_DEPTH
        // register buffers in rotation: the words of the DEPTH - 1 units behind the one being tested are in flight.  A
        // fetch past the wave's last unit re-reads that unit (no branch around a load: the compiler's wait counts stay
        // exact only in straight-line code).
        constexpr int DEPTH = RD_FT_DEPTH;
        uint32_t rb[DEPTH][NW];
        const int dq = G / wgps, dr = G % wgps;
        int fg = wave / wgps, fr = wave % wgps, fu = wave;
        int tg = fg, tr = fr, prio_k = 0;
        auto step = [&](int &gg, int &rr) { gg += dq; rr += dr; if (rr >= wgps) { rr -= wgps; gg++; } };
        auto fetch_next = [&](uint32_t (&r)[NW]) {
#ifdef RD_DIAG
            if (a.abl & 2) return;
#endif
            fetch(fg, fr, r);
            if (fu + G < units) { fu += G; step(fg, fr); }   // (scalar: past the last unit the position stays where it is)
        };
        auto test_next = [&](const uint32_t (&r)[NW]) {
#ifdef RD_DIAG
            if (a.abl & 1) { if (r[0] == 0x12345678u) s_cnt[0] = 1; step(tg, tr); return; }
#endif
            switch ((prio_q + prio_k++) & 3) {   // (the operand of s_setprio is an immediate)
                case 0: __builtin_amdgcn_s_setprio(0); break;
                case 1: __builtin_amdgcn_s_setprio(1); break;
                case 2: __builtin_amdgcn_s_setprio(2); break;
                default: __builtin_amdgcn_s_setprio(3); break;
            }
            test(tg, tr, r);
            step(tg, tr);
        };
        if (wave < units) {
#pragma unroll
            for (int j = 0; j < DEPTH - 1; j++) fetch_next(rb[j]);
            for (int u = wave; u < units; u += DEPTH * G) {
#pragma unroll
                for (int j = 0; j < DEPTH; j++) {
                    if (u + j * G < units) {   // wave-uniform
                        fetch_next(rb[(j + DEPTH - 1) % DEPTH]);
                        test_next(rb[j]);
                    }
                }
            }
        }
    }
    __builtin_amdgcn_s_setprio(0);
    if (wave == 0) RD_FT_STAMP(6);   // (wave 0's own end of the search: the barrier's wait is 2 - 6)
    rd_barrier_lds();
    if (wave == 0) RD_FT_STAMP(2);

    // ---- 2. slice: one wave per stream, lanes = matches (strips of 64) ----
    constexpr int NBITS = (K_ - 1) * S_ + 1;
    constexpr int NU = (NBITS + 31) / 32;
    constexpr int NWS = NU + 1;
    static_assert(K_ % 8 == 0 && (K_ + 7) / 8 <= 12 && NWS <= 36, "packet bytes fit a task entry");
    const int g = wave;                       // the stream of this wave
    const int stream = stream0 + g;
    const uint32_t nmatch = g < n_here ? s_cnt[g] : 0u;
    const bool ovf = nmatch > a.bucket_limit;
    const uint32_t count = ovf ? 0u : nmatch;  // (overflow: the host runs the separate kernels on this input)
    uint32_t *tg = s_t + (size_t)g * a.bcap * 8;
    const int B = a.cfg.B, L = a.cfg.L;
    for (uint32_t j = lane; j < count; j += 64) {
        const int pos = s_pos[(size_t)g * a.bcap + j];
        const uint32_t *w = a.lay.bits + (size_t)stream * a.lay.bits_stride;
        const int wi0 = pos >> 5;
        const uint32_t sh = (uint32_t)(pos & 31);
        uint32_t W[NWS];
#pragma unroll
        for (int i = 0; i < NWS; i++) W[i] = (wi0 + i < nwords) ? w[wi0 + i] : 0u;
        // calls that report this position (py:194, q <= B)
        const uint32_t pl = (uint32_t)(pos + L), bq = pl / (uint32_t)B, br = pl - bq * (uint32_t)B;
        const int b0 = (int)bq - 1;
        const int q0 = pos - ((b0 + 1) * B - L);
        const bool ok0 = b0 >= 0 && b0 < a.n_calls;
        const int b1 = b0 - 1, q1 = q0 + B;
        const bool ok1 = br == 0 && b1 >= 0 && b1 < a.n_calls;
        // the packet's bytes: byte bi = symbols 8 bi .. 8 bi + 7, first symbol = MSB (py:197-200)
        uint32_t uu[NU];
#pragma unroll
        for (int i = 0; i < NU; i++) uu[i] = __builtin_amdgcn_alignbit(W[i + 1], W[i], sh);
        uint32_t dw[3] = {0, 0, 0};
#pragma unroll
        for (int k = 0; k < K_; k++) {
            const int bit = k * S_, bi = k >> 3;
            const uint32_t b = (uu[bit >> 5] >> (bit & 31)) & 1u;
            dw[bi >> 2] |= b << (8 * (bi & 3) + 7 - (k & 7));
        }
        // keys: (phase, q), the order of py:171-188 inside a call; task 0 = (b0, q0), task 1 = (b0 - 1, q0 + B)
        const uint32_t ph0 = (uint32_t)q0 % (uint32_t)S_, ph1 = (uint32_t)q1 % (uint32_t)S_;
        uint4 *e = (uint4 *)(tg + 8 * j);
        e[0] = uint4{(ok0 ? 1u : 0u) | (ok1 ? 2u : 0u), (uint32_t)b0, (ph0 << 24) | (uint32_t)q0, (ph1 << 24) | (uint32_t)q1};
        e[1] = uint4{dw[0], dw[1], dw[2], 0u};
    }
    if (wave == 0) RD_FT_STAMP(7);   // (wave 0: its stream's entries are in LDS)
    // (a wave reads what the same wave wrote: LDS operations of a wave complete in order, no barrier needed)
    // pass 1: a task is a duplicate when a task of the same call with the same bytes precedes it (py:203-205)
    for (uint32_t j = lane; j < ((count + 63) & ~63u); j += 64) {
        const bool live = j < count;
        const uint4 me = live ? *(const uint4 *)(tg + 8 * j) : uint4{0, 0, 0, 0};
        const uint4 md = live ? *(const uint4 *)(tg + 8 * j + 4) : uint4{0, 0, 0, 0};
        const int b0 = (int)me.y, b1 = b0 - 1;
        const uint32_t k0 = me.z, k1 = me.w;
        bool dup0 = false, dup1 = false;
        // four entries per trip, all eight broadcast reads issued before the first is used (a trip per entry waits for
        // the LDS once per entry); the slots of a trip past the stream's count are read (they exist) and ignored
        for (uint32_t i0 = 0; i0 < count; i0 += 4) {
            uint4 xa[4], da[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t i = (i0 + u) < a.bcap ? i0 + u : i0;
                xa[u] = *(const uint4 *)(tg + 8 * i);
                da[u] = *(const uint4 *)(tg + 8 * i + 4);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t i = i0 + u;
                const uint4 x = xa[u], d = da[u];
                const int ob0 = (int)x.y;
                const bool lower = i < j;  // equal keys = the same position twice: the lower slot counts as the earlier one
                if (i < count && d.x == md.x && d.y == md.y && d.z == md.z) {
                    if ((x.x & 1u) && ob0 == b0 && (x.z < k0 || (x.z == k0 && lower))) dup0 = true;
                    if ((x.x & 2u) && ob0 - 1 == b0 && (x.w < k0 || (x.w == k0 && lower))) dup0 = true;
                    if ((x.x & 1u) && ob0 == b1 && (x.z < k1 || (x.z == k1 && lower))) dup1 = true;
                    if ((x.x & 2u) && ob0 - 1 == b1 && (x.w < k1 || (x.w == k1 && lower))) dup1 = true;
                }
            }
        }
        if (live) tg[8 * j + 7] = (((me.x & 1u) && !dup0) ? 1u : 0u) | (((me.x & 2u) && !dup1) ? 2u : 0u);
    }
    // pass 2: rank among the surviving tasks of the stream
    uint32_t kept_here = 0;
    for (uint32_t j = lane; j < ((count + 63) & ~63u); j += 64) {
        const bool live = j < count;
        const uint4 me = live ? *(const uint4 *)(tg + 8 * j) : uint4{0, 0, 0, 0};
        const uint32_t mk = live ? tg[8 * j + 7] : 0u;
        const int b0 = (int)me.y, b1 = b0 - 1;
        const uint32_t k0 = me.z, k1 = me.w;
        uint32_t r0 = 0, r1 = 0;
        for (uint32_t i0 = 0; i0 < count; i0 += 4) {
            uint4 xa[4];
            uint32_t ka[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t i = (i0 + u) < a.bcap ? i0 + u : i0;
                xa[u] = *(const uint4 *)(tg + 8 * i);
                ka[u] = tg[8 * i + 7];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint4 x = xa[u];
                const uint32_t xk = (i0 + u) < count ? ka[u] & 3u : 0u;
                const int ob0 = (int)x.y;
                auto before = [](int ca, uint32_t ka_, int cb, uint32_t kb) { return ca < cb || (ca == cb && ka_ < kb); };
                if (xk & 1u) { r0 += before(ob0, x.z, b0, k0) ? 1u : 0u; r1 += before(ob0, x.z, b1, k1) ? 1u : 0u; }
                if (xk & 2u) { r0 += before(ob0 - 1, x.w, b0, k0) ? 1u : 0u; r1 += before(ob0 - 1, x.w, b1, k1) ? 1u : 0u; }
            }
        }
        kept_here += (uint32_t)__popcll(__ballot((mk & 1u) != 0)) + (uint32_t)__popcll(__ballot((mk & 2u) != 0));
        if (live) tg[8 * j + 7] = mk | (r0 << 2) | (r1 << 17);   // (ranks < 2 bcap <= 2^15)
    }
    if (lane == 0) s_kept[g] = kept_here;
    if (ovf && lane == 0) atomicOr(&a.counters[RD_CNT_OVF], 1u);
    rd_barrier_lds();
    // the group's list: its streams' tasks one stream after the other, each stream's in rank order
    uint32_t off = 0, total = 0, mtotal = 0;
#pragma unroll
    for (int i = 0; i < G; i++) {
        off += i < g ? s_kept[i] : 0u;
        total += s_kept[i];
        mtotal += i < n_here ? s_cnt[i] : 0u;
    }
    for (uint32_t j = lane; j < count; j += 64) {
        const uint32_t res = tg[8 * j + 7];
        if (res & 1u) s_ord[off + ((res >> 2) & 0x7FFFu)] = ((uint32_t)g << 16) | (j << 1);
        if (res & 2u) s_ord[off + (res >> 17)] = ((uint32_t)g << 16) | (j << 1) | 1u;
    }
    // Publish the group's totals: three self-validating words (this launch's sequence number in the top twelve bits),
    // relaxed agent-scope stores - no release fence: at agent scope that is a write-back of the XCD's whole L2
    // (buffer_wbl2), once per workgroup, and an acquire on the reading side an invalidate per poll; the first version
    // of this kernel had both and took 165 us instead of 60.  Nothing else travels between workgroups.
    const uint32_t tag = (a.seq & 0xFFFu) << 20;
    if (tid == 0) {
        __hip_atomic_store(&a.gstate[n_groups + grp], tag | (mtotal < 0xFFFFFu ? mtotal : 0xFFFFFu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&a.gstate[2 * n_groups + grp], tag | (n_fix_raw < 0xFFFFFu ? n_fix_raw : 0xFFFFFu), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&a.gstate[grp], tag | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (the word the others wait for: last)
        if (n_fix_raw > a.fix_bcap) atomicOr(&a.counters[RD_CNT_OVF], 8u);
    }
    rd_barrier_lds();
    if (wave == 0) RD_FT_STAMP(3);

    // ---- 3. RSSI / SNR and the records ----
    if (wave == G - 1) {
        // records of the groups in front: one word each, valid once it carries this launch's sequence number
        uint32_t polls = 0;
        bool gave_up = false;
        auto tagged = [&](const uint32_t *p) -> uint32_t {  // the value behind this launch's tag (spins until it is there)
            for (;;) {
                const uint32_t v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((v & 0xFFF00000u) == tag) return v & 0xFFFFFu;
                if (++polls > (1u << 22)) { gave_up = true; return 0u; }  // (seconds: never, unless something is broken)
                __builtin_amdgcn_s_sleep(8);
            }
        };
        // (sixteen words per lane asked for before the first is looked at: one memory round trip per 1024 groups in
        // front, where a loop of dependent polls took one per 64)
        uint32_t sum = 0;
        for (int j0 = 0; j0 < grp; j0 += 1024) {
            uint32_t v[16];
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const int j = j0 + lane + 64 * k;
                v[k] = j < grp ? __hip_atomic_load(&a.gstate[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : tag;
            }
#pragma unroll
            for (int k = 0; k < 16; k++) {
                if ((v[k] & 0xFFF00000u) != tag) v[k] = tag | tagged(&a.gstate[j0 + lane + 64 * k]);  // not there yet: poll
                sum += v[k] & 0xFFFFFu;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += (uint32_t)__shfl_xor((int)sum, o, 64);
        if (lane == 0) __hip_atomic_store(&s_ready, sum + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (grp == n_groups - 1) {  // the last group leaves the run's totals for the host
            // (its prefix has seen every other group's first word, and a group stores its other two words BEFORE that one:
            // they are there, as a rule - all loads of a round go out before the first is looked at, a word that is not
            // there yet is polled.  A dependent load per word made this group the last to finish by 8 us)
            uint32_t mt = 0, ft = 0;
            for (int i0 = 0; i0 < n_groups; i0 += 512) {
                uint32_t vm[8], vf[8];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int i = i0 + lane + 64 * k;
                    vm[k] = i < n_groups ? __hip_atomic_load(&a.gstate[n_groups + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : tag;
                    vf[k] = i < n_groups ? __hip_atomic_load(&a.gstate[2 * n_groups + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : tag;
                }
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int i = i0 + lane + 64 * k;
                    if ((vm[k] & 0xFFF00000u) != tag) vm[k] = tagged(&a.gstate[n_groups + i]);
                    if ((vf[k] & 0xFFF00000u) != tag) vf[k] = tagged(&a.gstate[2 * n_groups + i]);
                    mt += vm[k] & 0xFFFFFu;
                    ft += vf[k] & 0xFFFFFu;
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                mt += (uint32_t)__shfl_xor((int)mt, o, 64);
                ft += (uint32_t)__shfl_xor((int)ft, o, 64);
            }
            if (lane == 0) {
                a.counters[RD_CNT_TASKS] = sum + total;
                a.counters[RD_CNT_MATCH] = mt;
                a.counters[RD_CNT_FIX] = ft;
                if (sum + total > a.rec_cap) atomicOr(&a.counters[RD_CNT_OVF], 2u);
            }
        }
        if (__ballot(gave_up) && lane == 0) atomicOr(&a.counters[RD_CNT_OVF], 16u);
        RD_FT_STAMP(4);
    }
    // every wave evaluates its share of the windows; the results wait in LDS for the group's first record number
    float *s_res = (float *)(s_dyn + (size_t)G * a.bcap * 11);
    if ((uint32_t)wave < total) {
        rd_k_h8 Ahi[3], Alo[3];
#pragma unroll
        for (int d = 0; d < 3; d++) {
            Ahi[d] = *(const rd_k_h8 *)g_rssi_taps.v[0][d][lane];
            Alo[d] = *(const rd_k_h8 *)g_rssi_taps.v[1][d][lane];
        }
        struct task { int stream, call, q; };
        auto task_at = [&](uint32_t r) {
            const uint32_t o = s_ord[r];
            const uint32_t *e = s_t + ((size_t)(o >> 16) * a.bcap + ((o >> 1) & 0x7FFFu)) * 8;
            task t;
            t.stream = stream0 + (int)(o >> 16);
            t.call = (int)e[1] - (int)(o & 1u);
            t.q = (int)(e[2] & 0xFFFFFFu) + ((o & 1u) ? B : 0);
            return t;
        };
        auto view = [&](int s) {
            rd_stream_view v;
            v.base = a.lay.iq + (size_t)s * a.lay.stream_stride;
            v.valid_from = a.lay.valid_from;
            v.n = a.lay.n_samples;
            return v;
        };
        auto job_of = [&](const task &t) {
            return rd_rssi_prepare(view(__builtin_amdgcn_readfirstlane(t.stream)), __builtin_amdgcn_readfirstlane(t.call) * B, a.cfg,
                                   __builtin_amdgcn_readfirstlane(t.q), lane);
        };
        task t_cur = task_at((uint32_t)wave);
        rd_rssi_job j_cur = job_of(t_cur);
        rd_rssi_data d_cur = {};
        if (j_cur.ok) d_cur = rd_rssi_fetch(j_cur);
        for (uint32_t r = (uint32_t)wave; r < total; r += G) {
            const task t_now = t_cur;
            const rd_rssi_job j_now = j_cur;
            const rd_rssi_data d_now = d_cur;
            if (r + G < total) {  // the next task's bytes are in flight under this one's arithmetic
                t_cur = task_at(r + G);
                j_cur = job_of(t_cur);
                if (j_cur.ok) d_cur = rd_rssi_fetch(j_cur);
            }
            double rssi = 0.0, snr = 0.0;
            if (j_now.ok) {
                float noise, sig;
                rd_rssi_block(j_now, d_now, Ahi, Alo, lane, noise, sig);
                rd_rssi_finish(noise, sig, j_now.ns, j_now.pe, j_now.q, lane, rssi, snr);
            } else {  // a window that reaches outside the stream: the fp32 path with its per-sample checks
                rd_rssi_u8(view(__builtin_amdgcn_readfirstlane(t_now.stream)), (long)__builtin_amdgcn_readfirstlane(t_now.call) * B, a.cfg,
                           (long)__builtin_amdgcn_readfirstlane(t_now.q), lane, rssi, snr);
            }
            // (both figures are float values widened to double - 3.0103 * v_log_f32 or the constants of py:233,236)
            if (lane == 0) { s_res[2 * r] = (float)rssi; s_res[2 * r + 1] = (float)snr; }
        }
    }
    rd_barrier_lds();
    // the records, one lane each, at the group's first record number + rank (known by now, as a rule)
    if ((uint32_t)tid < total) {
        uint32_t spins = 0, v = 0;
        while (!(v = __hip_atomic_load(&s_ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) && ++spins < (1u << 24))
            __builtin_amdgcn_s_sleep(2);
        const uint32_t first = v - 1u;
        for (uint32_t r = (uint32_t)tid; r < total; r += RD_FT_WG) {
            const uint32_t o = s_ord[r];
            const uint32_t *e = s_t + ((size_t)(o >> 16) * a.bcap + ((o >> 1) & 0x7FFFu)) * 8;
            const uint4 x = *(const uint4 *)e, d = *(const uint4 *)(e + 4);
            const uint32_t at = first + r;
            if (at < a.rec_cap) {
                const double rssi = (double)s_res[2 * r], snr = (double)s_res[2 * r + 1];
                uint4 *out = (uint4 *)&a.recs[at];
                out[0] = uint4{(uint32_t)(stream0 + (int)(o >> 16)), (uint32_t)((int)x.y - (int)(o & 1u)),
                               (uint32_t)((int)(x.z & 0xFFFFFFu) + ((o & 1u) ? B : 0)), (uint32_t)a.cfg.nbytes};
                out[1] = uint4{d.x, d.y, d.z, 0u};
                out[2] = uint4{0u, 0u, 0u, 0u};
                const uint2 rb = __builtin_bit_cast(uint2, rssi), sb = __builtin_bit_cast(uint2, snr);
                out[3] = uint4{rb.x, rb.y, sb.x, sb.y};
            }
        }
    }
    if (wave == 0) RD_FT_STAMP(5);
}

#ifdef RD_DIAG
static uint64_t *g_ft_stamps = nullptr;
static uint32_t g_ft_groups = 0;
// stamps of the last k_tail launch (diagnostic library): 8 uint64 per group (s_memrealtime, 100 MHz)
extern "C" int rd_diag_read_tail_stamps(uint64_t *out, uint32_t cap_groups, uint32_t *n_groups) {
    if (!g_ft_stamps || !n_groups) return RD_ERR_STATE;
    if (hipDeviceSynchronize() != hipSuccess) return RD_ERR_DEVICE;
    *n_groups = g_ft_groups;
    const uint32_t n = g_ft_groups < cap_groups ? g_ft_groups : cap_groups;
    if (n && hipMemcpy(out, g_ft_stamps, (size_t)n * 64, hipMemcpyDeviceToHost) != hipSuccess) return RD_ERR_DEVICE;
    return RD_OK;
}
#endif

size_t rd_tail_fused_lds(uint32_t bcap) { return (size_t)RD_FT_STREAMS * bcap * (8 + 1 + 2 + 4) * sizeof(uint32_t); }

// returns 1 when the fused tail was launched, 0 when the shape is not the one it is built for
int rd_launch_tail_fused(const rd_layout &lay, const rd_devcfg &cfg, int n_calls, long p_lo, long p_hi, const rd_ft_bufs &fb,
                         uint32_t bucket_limit, uint32_t seq, int skip_fix, rd_packet *recs, uint32_t rec_cap, uint32_t *counters,
                         hipStream_t st, hipEvent_t ev_stop, uint32_t *zero_next, uint32_t zero_words) {
    const long n_bits = lay.n_samples;
    if (!(cfg.S == 14 && cfg.P == 16 && cfg.K == 80 && cfg.pre_mask == 0x91D3ull) || n_bits >= (1l << 30) || cfg.B >= (1 << 24) ||
        lay.n_streams <= 0 || p_lo > 0 || p_hi < 0 || p_hi >= (1l << 30) || !fb.gstate || !fb.fixb || fb.bcap == 0 || fb.bcap % 32 ||
        fb.bcap > RD_BUCKET_MAX || lay.bits_stride % 4 || ((size_t)lay.bits % 16) || lay.hist_mode ||
        p_hi / 32 + 16 > (n_bits + 31) / 32)   // (a reported position's 12-word window inside the stream: k_tail's search loads)
        return 0;
    rd_ft_args a;
    a.lay = lay; a.cfg = cfg; a.n_calls = n_calls; a.p_hi = (int)p_hi; a.skip_fix = skip_fix;
    a.fixb = fb.fixb; a.fixcnt = fb.fixcnt; a.fix_bcap = fb.fix_bcap;
    a.bcap = fb.bcap; a.bucket_limit = bucket_limit < fb.bcap ? bucket_limit : fb.bcap;
    a.gstate = fb.gstate; a.seq = seq;
    a.recs = recs; a.rec_cap = rec_cap; a.counters = counters; a.zero_next = zero_next; a.zero_words = zero_words;
    a.stamps = nullptr;
    a.abl = 0;
    const uint32_t groups = (uint32_t)(lay.n_streams + RD_FT_STREAMS - 1) / RD_FT_STREAMS;
#ifdef RD_DIAG
    if (const char *e = getenv("RD_FT_ABL")) a.abl = atoi(e);
    if (getenv("RD_FT_STAMPS")) {
        if (groups > g_ft_groups) { if (g_ft_stamps) hipFree(g_ft_stamps); g_ft_stamps = nullptr; if (hipMalloc(&g_ft_stamps, (size_t)groups * 64) != hipSuccess) g_ft_stamps = nullptr; }
        g_ft_groups = g_ft_stamps ? groups : 0;
        a.stamps = g_ft_stamps;
    }
#endif
    // at least 33 KiB per workgroup: FOUR workgroups per CU (one wave of each per SIMD) and never five - the register
    // budget would admit a fifth, and the CUs that got one ran their search a third slower than the rest while every
    // group behind them waited for their totals
    const size_t lds = std::max<size_t>(rd_tail_fused_lds(fb.bcap), 33 * 1024);
    static bool attr_set = false;
    if (lds > 48 * 1024 && !attr_set) {
        hipFuncSetAttribute((const void *)k_tail<14, 16, 0x91D3ull, 80>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        attr_set = true;
    }
    if (ev_stop)
        hipExtLaunchKernelGGL((k_tail<14, 16, 0x91D3ull, 80>), dim3(groups), dim3(RD_FT_WG), (unsigned)lds, st, nullptr, ev_stop, 0, a);
    else
        hipLaunchKernelGGL((k_tail<14, 16, 0x91D3ull, 80>), dim3(groups), dim3(RD_FT_WG), (unsigned)lds, st, a);
    return 1;
}

static uint32_t rd_slice_grid(uint32_t match_cap) {
    uint32_t wgs = (match_cap + 3) / 4;
    if (wgs > 4096) wgs = 4096;
    return wgs ? wgs : 1;
}

int rd_launch_slice(const rd_layout &lay, const uint32_t *bits, size_t bits_stride, long n_bits, const rd_devcfg &cfg,
                    const rd_match *matches, uint32_t match_cap, int batch_mode, int n_calls, int call,
                    rd_packet *recs, rd_packet *recs_host, uint32_t *counters, hipStream_t st, hipEvent_t ev_stop,
                    void *tasks) {
    rd_u8_src src;
    src.lay = lay;
    const int two = rd_k_get_params().slice_two;  // RD_SLICE_IMPL=wave: the one-kernel form (A/B)
    // the Davis shape in the batch path: lane-per-match classification, then one wave per surviving packet
    if (two && tasks && batch_mode && recs && !recs_host && cfg.S == 14 && cfg.K == 80 && n_bits < (1l << 30)) {
        const uint32_t cg = std::min<uint32_t>((match_cap + 255) / 256, 1024);
        hipLaunchKernelGGL((k_classify<14, 80>), dim3(cg ? cg : 1), dim3(256), 0, st, bits, bits_stride,
                           (int)((n_bits + 31) / 32), cfg, matches, match_cap, n_calls, recs, (rd_task *)tasks, counters);
        // one resident generation of waves (4 per SIMD): each works through its ~6 packets with the next one's
        // bytes in flight; a larger grid means generations of waves that each pay the whole latency chain
        const uint32_t rg = std::min<uint32_t>(rd_slice_grid(match_cap), 1024);
        if (ev_stop)
            hipExtLaunchKernelGGL(k_rssi_u8, dim3(rg), dim3(256), 0, st, nullptr, ev_stop, 0, lay, cfg,
                                  (const rd_task *)tasks, 2 * match_cap, recs, counters);
        else
            hipLaunchKernelGGL(k_rssi_u8, dim3(rg), dim3(256), 0, st, lay, cfg, (const rd_task *)tasks,
                               2 * match_cap, recs, counters);
        return 1;  // records: one per task, dense from index 0 (RD_CNT_TASKS of them)
    }
    if (ev_stop)  // the dispatch records the event itself (no marker packet behind the kernel)
        hipExtLaunchKernelGGL(k_slice_rssi<rd_u8_src>, dim3(rd_slice_grid(match_cap)), dim3(256), 0, st, nullptr,
                              ev_stop, 0, src, bits, bits_stride, (n_bits + 31) / 32, cfg, matches, match_cap,
                              batch_mode, n_calls, call, recs, recs_host, counters);
    else
        hipLaunchKernelGGL(k_slice_rssi<rd_u8_src>, dim3(rd_slice_grid(match_cap)), dim3(256), 0, st, src, bits,
                           bits_stride, (n_bits + 31) / 32, cfg, matches, match_cap, batch_mode, n_calls, call, recs,
                           recs_host, counters);
    return 0;  // records: one slot per match (RD_CNT_MATCH, void ones marked) + the block-boundary twins behind match_cap
}

// ------------------------------------------------------------------------------------------
// Parser.parse front half (protocol.py:282-311), batch mode.
// k_parse_select: one lane per final record: bit-reverse every byte (:290, :79-83), CRC-16-CCITT
// (poly 0x1021, init 0; crc.py:19-26) over data[2:] must be 0 (:297); survivors are compacted.
// k_freq_err: one wave per survivor: mean of discriminated[index : index + preamble_length]
// (:304-311) in float64, where discriminated covers absolute samples [(call-1)B, (call+1)B).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t rd_swap_bits8(uint32_t b) {  // protocol.py:79-83
    b = ((b & 0xF0) >> 4) | ((b & 0x0F) << 4);
    b = ((b & 0xCC) >> 2) | ((b & 0x33) << 2);
    b = ((b & 0xAA) >> 1) | ((b & 0x55) << 1);
    return b;
}

__global__ __launch_bounds__(256) void k_parse_select(const rd_packet *recs, uint32_t match_cap, rd_parsed *parsed,
                                                      uint32_t *counters, int dense) {
    // records: [0, matches) one per match (stream < 0: reported by no call) and
    // [match_cap, match_cap + RD_CNT_REC) the second records of block-boundary positions
    uint32_t nprim = counters[RD_CNT_MATCH], nextra = counters[RD_CNT_REC];
    if (nprim > match_cap) nprim = match_cap;
    if (nextra > match_cap) nextra = match_cap;
    if (dense) {  // the two-kernel slice writes one record per task, densely, from index 0
        nprim = counters[RD_CNT_TASKS];
        if (nprim > 2 * match_cap) nprim = 2 * match_cap;
        nextra = 0;
    }
    const uint32_t rec_cap = 2 * match_cap;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    bool ok = false;
    uint8_t sw[RD_MAX_PKT_BYTES];
    int nb = 0;
    if ((i < nprim || (!dense && i >= match_cap && i - match_cap < nextra)) && recs[i].stream >= 0) {
        const rd_packet *r = &recs[i];
        nb = r->nbytes;
        uint32_t crc = 0;
#pragma unroll
        for (int k = 0; k < RD_MAX_PKT_BYTES; k++) {
            const uint32_t b = k < nb ? rd_swap_bits8(r->data[k]) : 0u;
            sw[k] = (uint8_t)b;
            if (k >= 2 && k < nb) {  // crc.py:24-25, bitwise form
                crc ^= b << 8;
#pragma unroll
                for (int j = 0; j < 8; j++) crc = (crc & 0x8000) ? ((crc << 1) ^ 0x1021) & 0xFFFF : (crc << 1) & 0xFFFF;
            }
        }
        ok = nb > 2 && crc == 0;
    }
    const uint64_t km = __ballot(ok);
    if (!km) return;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&counters[RD_CNT_PARSED], (uint32_t)__popcll(km));
    base = __builtin_amdgcn_readfirstlane(base);
    if (!ok) return;
    const uint32_t dst = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(km >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)km, 0));
    if (dst >= rec_cap) return;
    const rd_packet *r = &recs[i];
    rd_parsed *o = &parsed[dst];
    o->stream = r->stream; o->call = r->call; o->index = r->index; o->freq_err = 0;
    o->id = sw[2] & 7; o->nbytes = nb - 2;
#pragma unroll
    for (int k = 0; k < RD_MAX_PKT_BYTES; k++) o->data[k] = (k + 2 < RD_MAX_PKT_BYTES && k + 2 < nb) ? sw[k + 2] : 0;
    o->rssi = r->rssi; o->snr = r->snr;
}

__global__ __launch_bounds__(256) void k_freq_err(rd_layout lay, rd_devcfg cfg, rd_parsed *parsed, uint32_t rec_cap,
                                                  const uint32_t *counters) {
    const int lane = threadIdx.x & 63;
    uint32_t count = counters[RD_CNT_PARSED];
    if (count > rec_cap) count = rec_cap;
    const uint32_t nw = gridDim.x * (blockDim.x >> 6);
    for (uint32_t i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); i < count; i += nw) {
        rd_parsed *o = &parsed[i];
        rd_stream_view v;
        v.base = lay.iq + (size_t)o->stream * lay.stream_stride;
        v.valid_from = lay.valid_from;
        v.n = lay.n_samples;
        // discriminated[j] of call b is d[(b-1)B + j], j in [0, 2B) (py:134,156,162)
        const long j0 = o->index;
        long j1 = j0 + cfg.PL;
        if (j1 > 2L * cfg.B) j1 = 2L * cfg.B;
        const long t_base = ((long)o->call - 1) * cfg.B;
        double sum = 0.0;
        for (long j = j0 + lane; j < j1; j += 64) {
            const long t = t_base + j;
            sum += rd_disc_f64(rd_f_f64(v, t - 1), rd_f_f64(v, t));
        }
        sum = rd_wave_sum(sum);
        if (lane == 0) {
            const double mean = sum / (double)(j1 - j0);
            o->freq_err = -(int32_t)((mean * cfg.fs) / (2.0 * 3.141592653589793));  // int(): toward zero
        }
    }
}

void rd_launch_parse(const rd_layout &lay, const rd_devcfg &cfg, const rd_packet *recs, uint32_t match_cap,
                     rd_parsed *parsed, uint32_t *counters, hipStream_t st, int dense) {
    const uint32_t rec_cap = 2 * match_cap;
    hipLaunchKernelGGL(k_parse_select, dim3((rec_cap + 255) / 256), dim3(256), 0, st, recs, match_cap, parsed,
                       counters, dense);
    uint32_t wgs = (rec_cap + 3) / 4;
    if (wgs > 2048) wgs = 2048;
    hipLaunchKernelGGL(k_freq_err, dim3(wgs), dim3(256), 0, st, lay, cfg, parsed, rec_cap, counters);
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
                                long n_block_bits, size_t win_stride, size_t block_stride) {
    const long nw = (n_win_bits + 31) / 32;
    const long nbw = (n_block_bits + 31) / 32;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nw) return;
    out += (size_t)blockIdx.y * win_stride;  // one stream per grid row
    in += (size_t)blockIdx.y * win_stride;
    block += (size_t)blockIdx.y * block_stride;
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
                             long n_block_bits, int n_streams, size_t win_stride, size_t block_stride,
                             hipStream_t st) {
    const long nw = (n_win_bits + 31) / 32;
    hipLaunchKernelGGL(k_window_update, dim3((unsigned)((nw + 255) / 256), (unsigned)n_streams), dim3(256), 0, st,
                       win_out, win_in, n_win_bits, block, n_block_bits, win_stride, block_stride);
}

// ------------------------------------------------------------------------------------------
// k_stream_block: ONE launch per demodulate() call of the streaming handle (py:139-246 for one block per stream).
// The multi-launch form (copy, three ring copies, memset, demod, fix-up, window update, search, slice, counter copy,
// event) spends 70 us on launch overheads and idle gaps for 16 KB of input; here one workgroup per stream does it all:
//   0  roll the raw ring (py:140,154) and take the new block straight from pinned host memory
//   1  the block's sign bits, EXACTLY (rd_exact_group_dw, one 8-sample group per thread: 8192 samples need no fast path)
//   2  quantized window = previous block's bits | new bits (py:157,163-166), kept in LDS and written out for the mirrors
//   3  preamble search over positions 0 .. B (py:171-188, q <= B: py:194)
//   4  slice + RSSI / SNR, one wave per match (py:190-246), records into mapped host memory
//   5  per stream: match count, then - behind a system-scope fence - the sequence number the host polls for
// Production shape class only (compile-time S, P, K; buffer_length = 2 block_size; block_size a multiple of 32, at most
// 16384): everything else keeps the multi-launch form.  Every position 0 .. B can be a match (a degenerate input makes
// it so): the match list in LDS and the stream's region of the mapped record array hold B + 1 entries.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t rd_lds_bits32(const uint32_t *w, int nwords, int o) {  // bits o .. o+31 (zeros outside)
    if (o < 0) return o <= -32 ? 0u : (w[0] << (-o));
    const int wi = o >> 5;
    const uint32_t lo = wi < nwords ? w[wi] : 0u, hi = wi + 1 < nwords ? w[wi + 1] : 0u;
    return __builtin_amdgcn_alignbit(hi, lo, (uint32_t)(o & 31));
}

#ifdef RD_DIAG
// diagnostic library: s_memrealtime stamps of the streaming blocks' phases (stream 0's workgroup), RD_SB_STAMPS=1
#define RD_SB_STAMP(k) do { if (a.stamps && threadIdx.x == 0 && blockIdx.x == 0) { uint64_t t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : : "memory"); \
                                                                                 a.stamps[(k)] = t_; } } while (0)
// (the same from whichever workgroup gets there: k_stream_block_cplx's last workgroup)
#define RD_SBL_STAMP(k) do { if (a.stamps && threadIdx.x == 0) { uint64_t t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : : "memory"); \
                                                              a.stamps[(k)] = t_; } } while (0)
#else
#define RD_SB_STAMP(k) do { } while (0)
#define RD_SBL_STAMP(k) do { } while (0)
#endif
#ifdef RD_DIAG
static uint64_t *g_sb_stamps = nullptr;
static uint64_t *rd_sb_stamp_buffer() {
    if (!getenv("RD_SB_STAMPS")) return nullptr;
    if (!g_sb_stamps && (hipMalloc(&g_sb_stamps, 64) != hipSuccess || hipMemset(g_sb_stamps, 0, 64) != hipSuccess)) g_sb_stamps = nullptr;
    return g_sb_stamps;
}
// stamps of the last streaming one-launch block (diagnostic library): 8 uint64 (s_memrealtime, 100 MHz)
extern "C" int rd_diag_read_sb_stamps(uint64_t *out) {
    if (!g_sb_stamps || !out) return RD_ERR_STATE;
    if (hipDeviceSynchronize() != hipSuccess) return RD_ERR_DEVICE;
    return hipMemcpy(out, g_sb_stamps, 64, hipMemcpyDeviceToHost) == hipSuccess ? RD_OK : RD_ERR_DEVICE;
}
#endif

template <int S_, int P_, uint64_t PRE_, int K_>
__global__ __launch_bounds__(RD_SB_THREADS) void k_stream_block(rd_sb_args a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t sb_lds[];
    const int B = a.cfg.B, nwin = (2 * B) / 32, nbw = B / 32;
    uint8_t *s_iq = sb_lds;                                      // samples -16 .. B-1 as bytes: 32 + 2 B
    uint32_t *s_win = (uint32_t *)(sb_lds + 32 + 2 * (size_t)B);  // the 2 B-bit window
    int32_t *s_match = (int32_t *)(s_win + nwin);                // B + 1 positions
    __shared__ uint32_t s_nm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int stream = blockIdx.x;
    uint8_t *ring = a.ring + (size_t)stream * a.ring_stride;     // [hdr 32 B][previous block][newest block]
    const uint8_t *in = a.in + (size_t)stream * 2 * (size_t)B;
    const bool have_hist = a.seen_before > 0;
    RD_SB_STAMP(0);
    if (tid == 0) s_nm = 0;
    // ---- 0: the roll.  Every load first (two 16-byte pieces per thread cover 2 B <= 32 KiB), then a barrier, then the
    // stores: the old newest block is read whole before it is overwritten, the old previous block's tail before the old
    // newest block lands on it.
    const int pieces = (2 * B) / 16;
    uint4 nw[2], oc[2], ph = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int i = tid + RD_SB_THREADS * k;
        nw[k] = uint4{0, 0, 0, 0}; oc[k] = uint4{0, 0, 0, 0};
        if (i < pieces) {
            nw[k] = *(const uint4 *)(in + 16 * (size_t)i);  // pinned host memory
            if (have_hist) oc[k] = *(const uint4 *)(ring + 32 + 2 * (size_t)B + 16 * (size_t)i);
        }
    }
    if (have_hist && tid < 2) ph = *(const uint4 *)(ring + 2 * (size_t)B + 16 * (size_t)tid);  // last 32 bytes of the old previous block
    // the previous block's bits: the upper half of the old window becomes the lower half of the new one
    const uint32_t *win_in = a.win_in + (size_t)stream * nwin;
    uint32_t *win_out = a.win_out + (size_t)stream * nwin;
    uint32_t oldw = 0;
    if (tid < nbw) oldw = win_in[nbw + tid];
    __syncthreads();
    RD_SB_STAMP(1);
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int i = tid + RD_SB_THREADS * k;
        if (i < pieces) {
            *(uint4 *)(ring + 32 + 2 * (size_t)B + 16 * (size_t)i) = nw[k];
            *(uint4 *)(s_iq + 32 + 16 * (size_t)i) = nw[k];
            if (have_hist) {
                *(uint4 *)(ring + 32 + 16 * (size_t)i) = oc[k];
                if (i >= pieces - 2) *(uint4 *)(s_iq + 16 * (i - (pieces - 2))) = oc[k];  // samples -16 .. -1
            }
        }
    }
    if (have_hist && tid < 2) *(uint4 *)(ring + 16 * (size_t)tid) = ph;
    if (!have_hist && tid < 2) *(uint4 *)(s_iq + 16 * tid) = uint4{0, 0, 0, 0};
    if (tid < nbw) s_win[tid] = oldw;
    rd_barrier_lds();   // (LDS only: the ring stores drain under the arithmetic - 2 us of waiting for their acknowledgement
                        // otherwise; the barriers in front of the slice, which reads the ring, wait for them)
    RD_SB_STAMP(2);
    // ---- 1: exact sign bits, one 8-sample group per thread (dsp.py:38-98 in exact integer arithmetic) ----
    const long vfrom = have_hist ? -16 : 0;  // (ten samples of history are all a group needs)
    for (int g = tid; g < B / 8; g += RD_SB_THREADS) {
        const int t0 = 8 * g;
        uint32_t dw[10];
#pragma unroll
        for (int d = 0; d < 10; d++) {
            const int smp = t0 - 10 + 2 * d;  // first sample of the dword
            dw[d] = (smp + 1 >= -16 && smp + 1 < B) ? *(const uint32_t *)(s_iq + 32 + 2 * smp) : 0u;
        }
        ((uint8_t *)(s_win + nbw))[g] = (uint8_t)rd_exact_group_dw(dw, (long)t0, 8, vfrom);
    }
    __syncthreads();
    RD_SB_STAMP(3);
    // ---- 2: the window goes out for the state mirrors (rd_copy_quantized) and the next call ----
    for (int i = tid; i < nwin; i += RD_SB_THREADS) win_out[i] = s_win[i];
    // ---- 3: search, one 32-position word per thread; positions 0 .. B ----
    for (int o = tid; o <= nbw; o += RD_SB_THREADS) {
        uint32_t m = 0xFFFFFFFFu;
#pragma unroll
        for (int k = 0; k < P_; k++) {
            const uint32_t v = rd_lds_bits32(s_win, nwin, 32 * o + k * S_);
            m &= ((PRE_ >> k) & 1) ? v : ~v;
        }
        if (o == nbw) m &= 1u;  // position B only
        while (m) {
            const int bpos = __builtin_ctz(m);
            m &= m - 1;
            const uint32_t slot = atomicAdd(&s_nm, 1u);
            s_match[slot] = 32 * o + bpos;  // (at most B + 1 of them)
        }
    }
    __syncthreads();
    RD_SB_STAMP(4);
    const uint32_t nm = s_nm;
    // ---- 4: slice + RSSI / SNR, one wave per match (k_slice_rssi's logic for batch_mode = 0) ----
    rd_packet *recs = a.recs_host + (size_t)stream * (size_t)(B + 1);
    rd_stream_view v;
    v.base = ring + 32 + 2 * (size_t)B;
    v.valid_from = a.seen_before <= 0 ? 0 : a.seen_before == 1 ? -(long)B : -(long)(B + 16);
    v.n = B;
    {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // the ring stores above are read below (same CU)
        for (uint32_t i = (uint32_t)wave; i < nm; i += RD_SB_THREADS / 64) {
            const int pos = __builtin_amdgcn_readfirstlane(s_match[i]);
            uint32_t byte = 0;
            bool same_prev = true, same_next = true;
            for (int r = 0; r * 64 < K_; r++) {
                const int k = 64 * r + lane;
                const uint32_t tri = k < K_ ? rd_lds_bits32(s_win, nwin, pos - 1 + k * S_) : 0u;
                const uint64_t mp = __ballot((tri & 1u) != 0), mm = __ballot((tri & 2u) != 0), mn = __ballot((tri & 4u) != 0);
                same_prev &= mp == mm;
                same_next &= mn == mm;
                const int bi = lane - 8 * r;
                if (bi >= 0 && bi < 8) {
                    const int have = K_ - 8 * lane;
                    const uint32_t rev = __builtin_bitreverse32((uint32_t)((mm >> (8 * bi)) & 0xFF)) >> 24;
                    byte = have >= 8 ? rev : have > 0 ? rev >> (8 - have) : 0u;
                }
            }
            const uint32_t phs = (uint32_t)pos % (uint32_t)S_;
            const bool prev_first = S_ == 1 || phs != 0, next_first = S_ > 1 && phs == S_ - 1;
            const bool superseded = (same_prev && pos >= 1 && prev_first) || (same_next && pos + 1 <= B && next_first);
            if (superseded) {
                rd_store_void(nullptr, &recs[i], lane);
                continue;
            }
            double rssi = 0.0, snr = 0.0;
            rd_rssi_u8(v, 0, a.cfg, (long)pos, lane, rssi, snr);
            rd_store_record(nullptr, &recs[i], lane, stream, (long)a.seen_before, (long)pos, a.cfg.nbytes, byte, rssi, snr);
        }
    }
    // ---- 5: count, fence, flag ----
    __syncthreads();
    if (tid == 0) {
        a.cnt_host[stream] = nm;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // system scope: records and count before the flag
        __hip_atomic_store(&a.flag_host[stream], a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    RD_SB_STAMP(5);
}

int rd_launch_stream_block(const rd_sb_args &a_in, int n_streams, hipStream_t st) {
    const rd_devcfg &c = a_in.cfg;
    if (!(c.S == 14 && c.P == 16 && c.K == 80 && c.pre_mask == 0x91D3ull) || c.L != 2 * c.B || c.B % 32 || c.B < 2048 || c.B > 16384)
        return 0;
    rd_sb_args a = a_in;
    a.stamps = nullptr;
#ifdef RD_DIAG
    a.stamps = rd_sb_stamp_buffer();
#endif
    const size_t lds = 32 + 2 * (size_t)c.B + (size_t)(2 * c.B / 32) * 4 + ((size_t)c.B + 1) * 4;
    hipLaunchKernelGGL((k_stream_block<14, 16, 0x91D3ull, 80>), dim3((unsigned)n_streams), dim3(RD_SB_THREADS), lds, st, a);
    return 1;
}

// ------------------------------------------------------------------------------------------
// k_stream_block_cplx: the same ONE launch for the complex-input branch (py:144-150) - what pyrtlsdr's sdr.stream()
// feeds the live receiver (/root/reference/src/rtldavis/runners/rtlsdr.py:100-103).  Single stream; the raw ring holds
// complex128 [hdr 16][previous block][newest block]; the new block comes from the slot's mapped host buffer as
// complex128 (16 bytes per sample) or - a uint8 block on a handle that has seen complex input - as bytes through the
// LUT (py:26,38-39).  The block is 128 KB over the link: ONE workgroup asked for it at 10.7 GB/s (12 of its 19 us;
// a CU has only so many reads in flight), so block_size / (8 T) workgroups of T threads each take 8 T samples:
//   0  every load of the piece at once: the new samples from pinned host memory, the old newest block from the ring
//      (py:140,154: the roll); the 16 samples in front of the piece for the filter's history
//   1  ring stores (they drain under the arithmetic) and the ROTATED samples (py:46-49) into LDS, sample i in 16-byte
//      chunk i + i/8: a thread of step 2 reads the 17 consecutive samples of its 8-sample group, lanes 8 samples apart
//      = chunk stride 9, no bank asked twice (the same loads from global memory touched 64 cache lines per instruction)
//   2  signs from a float64 sum in tap order (rd_f_f64's), one 8-sample group per thread.  The sign of py:80-90's
//      quotient is the sign of its numerator - the denominator is a sum of squares + 1e-10 - so no float64 divide
//      (finite input; a NaN numerator has no sign to agree on)
//   3  the piece's sign words into the window; then the workgroup counts itself in.  The LAST one to arrive goes on
//      alone: window, search, slice, RSSI / SNR, records, flag (k_stream_block's steps 2 - 5)
// What one workgroup writes and another reads in the same launch: the window's words travel as relaxed agent-scope
// atomics (sc1: written through to, read from the level all XCDs share); the ring (the RSSI windows reach into other
// pieces and the previous block) is written with plain stores and ONE agent-scope release per wave in front of the
// workgroup's increment of the arrival counter (a write-back of this XCD's few dirty L2 lines: 0.6 us; the same ring as
// 32 write-through stores per thread took 6 us of issue) and read by the last workgroup with sc1 loads.  Nobody waits
// for anybody: a workgroup that is not the last simply ends.  The ring's last 16 samples (next block's filter
// history, and what the header copies) are read AND written by workgroup 0 only, whichever piece they belong to: no
// other workgroup's store can land on them before they are read.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int rd_sbc_chunk(int slot) { return slot + (slot >> 3); }   // (slot = sample - piece start + 16)
static int rd_sbc_chunk_host(int slot) { return slot + (slot >> 3); }

__device__ __forceinline__ double rd_load_coh(const double *p) {
    return __builtin_bit_cast(double, __hip_atomic_load((const uint64_t *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

template <int S_, int P_, uint64_t PRE_, int K_>
__global__ __launch_bounds__(RD_SBC_THREADS) void k_stream_block_cplx(rd_sbc_args a) {
    constexpr int T = RD_SBC_THREADS, C = 8 * T, PER = 8;
    extern __shared__ __attribute__((aligned(16))) uint8_t sb_lds[];
    const int B = a.cfg.B, nwin = (2 * B) / 32, nbw = B / 32;
    uint32_t *s_win = (uint32_t *)sb_lds;           // the 2 B-bit window (the last workgroup)
    uint4 *s_y = (uint4 *)(s_win + nwin);           // rotated samples -16 .. C-1 of the piece, padded (rd_sbc_chunk)
    int32_t *s_match = (int32_t *)s_y;              // B + 1 positions: over the samples, once the signs are made
    __shared__ uint32_t s_nm, s_last;
    __shared__ __attribute__((aligned(4))) uint8_t s_bytes[T];
    __shared__ double s_part[2][2][T / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int w = blockIdx.x, NW = gridDim.x, s0 = w * C;
    double *ring = a.ring;                          // [hdr 32 doubles][previous block 2 B][newest block 2 B]
    uint4 *r_hdr = (uint4 *)ring, *r_prev = (uint4 *)(ring + 32), *r_cur = (uint4 *)(ring + 32 + 2 * (size_t)B);  // one uint4 = one sample
    const bool have_hist = a.seen_before > 0;
    if (w == 0) RD_SB_STAMP(0);
    struct smp { uint64_t re, im; };   // one complex128 sample as bit patterns
    const int in_is_u8 = a.in_is_u8;
    const void *in = a.in;
    auto fetch_in = [in, in_is_u8](int i) -> smp {
        if (in_is_u8) {  // py:26,38-39
            const uint8_t *q = (const uint8_t *)in + 2 * (size_t)i;
            const double re = ((double)q[0] - 127.4) / 127.6, im = ((double)q[1] - 127.4) / 127.6;
            return smp{__builtin_bit_cast(uint64_t, re), __builtin_bit_cast(uint64_t, im)};
        }
        const uint4 r = ((const uint4 *)in)[i];  // pinned host memory
        return smp{((uint64_t)r.y << 32) | r.x, ((uint64_t)r.w << 32) | r.z};
    };
    auto fetch_ring = [](const uint4 *p) -> smp {
        const uint4 r = *p;
        return smp{((uint64_t)r.y << 32) | r.x, ((uint64_t)r.w << 32) | r.z};
    };
    auto store_coh = [](uint4 *p, smp v) {   // (plain 16-byte store: the release fence in front of the count writes the L2 back)
        *p = uint4{(uint32_t)v.re, (uint32_t)(v.re >> 32), (uint32_t)v.im, (uint32_t)(v.im >> 32)};
    };
    auto rotated = [](smp x, int n) -> uint4 {  // py:46-49, x * j^(n mod 4) (rd_rot_f64), on the bit patterns: a negation flips bit 63
        const uint64_t neg = 1ull << 63;
        const int ph = n & 3;
        const uint64_t re = ph == 0 ? x.re : ph == 1 ? x.im ^ neg : ph == 2 ? x.re ^ neg : x.im;
        const uint64_t im = ph == 0 ? x.im : ph == 1 ? x.re : ph == 2 ? x.im ^ neg : x.re ^ neg;
        return uint4{(uint32_t)re, (uint32_t)(re >> 32), (uint32_t)im, (uint32_t)(im >> 32)};
    };
    // ---- 0: the piece's loads, all of them before the first store ----
    const uint4 zero4 = {0, 0, 0, 0};
    const smp zero = {0, 0};
    smp nw[PER], oc[PER], tl_new = zero, tl_old = zero, ph = zero, hist = zero;
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int i = s0 + tid + T * k;
        nw[k] = zero; oc[k] = zero;
        if (i < B) {
            nw[k] = fetch_in(i);
            if (have_hist && i < B - 16) oc[k] = fetch_ring(&r_cur[i]);
        }
    }
    if (tid < 16) {
        if (w == 0) {
            tl_new = fetch_in(B - 16 + tid);
            if (have_hist) { tl_old = fetch_ring(&r_cur[B - 16 + tid]); ph = fetch_ring(&r_prev[B - 16 + tid]); }
            hist = tl_old;                      // (zeros without history: the zero state of py:133)
        } else {
            hist = fetch_in(s0 - 16 + tid);     // (the neighbour's samples, once more over the link)
        }
    }
    const int wj = s0 / 32 + tid;               // this thread's word of the piece (tid < C / 32)
    uint32_t oldw = 0;
    if (tid < C / 32 && wj < nbw) oldw = a.win_in[nbw + wj];
    // ---- 1: the old newest block moves down (it has been here a while), the new block follows as the link delivers it.
    // A thread overwrites only what it has read itself.
    if (have_hist) {
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const int i = s0 + tid + T * k;
            if (i < B - 16) store_coh(&r_prev[i], oc[k]);
        }
    }
    if (tid < 16) s_y[rd_sbc_chunk(tid)] = (w > 0 || have_hist) ? rotated(hist, s0 - 16 + tid) : zero4;
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int i = s0 + tid + T * k;
        if (i < B) {
            if (i < B - 16) store_coh(&r_cur[i], nw[k]);
            s_y[rd_sbc_chunk(i - s0 + 16)] = rotated(nw[k], i);
        }
    }
    if (w == 0 && tid < 16) {
        if (have_hist) { store_coh(&r_hdr[tid], ph); store_coh(&r_prev[B - 16 + tid], tl_old); }
        store_coh(&r_cur[B - 16 + tid], tl_new);
    }
    if (tid < C / 32 && wj < nbw) __hip_atomic_store(&a.win_out[wj], oldw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    rd_barrier_lds();   // (LDS only: the ring stores drain under the arithmetic)
    if (w == 0) RD_SB_STAMP(1);
    // ---- 2: sign bits in float64, one 8-sample group per thread (py:52-98; taps summed in order m = 0..8) ----
    {
        const int t0 = s0 + 8 * tid;
        uint32_t byte = 0;
        if (t0 < B) {
            const double c[9] = {RD_C0, RD_C1, RD_C2, RD_C3, RD_C4, RD_C3, RD_C2, RD_C1, RD_C0};
            rd_d2 y[17];
#pragma unroll
            for (int i = 0; i < 17; i++) {  // samples t0 - 10 + i
                const uint4 x = s_y[rd_sbc_chunk(8 * tid + 6 + i)];
                y[i].x = __builtin_bit_cast(double, uint2{x.x, x.y});
                y[i].y = __builtin_bit_cast(double, uint2{x.z, x.w});
            }
            rd_d2 prev = {0.0, 0.0};
#pragma unroll
            for (int j = 0; j < 9; j++) {  // f[t0 - 1 + j]
                rd_d2 f = {0.0, 0.0};
                if (have_hist || t0 - 1 + j >= 0) {
#pragma unroll
                    for (int m = 0; m < 9; m++) {
                        f.x += c[m] * y[j + m].x;
                        f.y += c[m] * y[j + m].y;
                    }
                }
                if (j > 0) byte |= rd_signbit_f64(prev.y * f.x - prev.x * f.y) << (j - 1);   // (py:80-90: the numerator's sign)
                prev = f;
            }
        }
        s_bytes[tid] = (uint8_t)byte;
    }
    rd_barrier_lds();
    if (w == 0) RD_SB_STAMP(2);
    // ---- 3: the piece's words, then the count ----
    if (tid < C / 32 && wj < nbw)
        __hip_atomic_store(&a.win_out[nbw + wj], ((const uint32_t *)s_bytes)[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // the piece's ring stores are plain stores into this XCD's L2: one agent-scope release per wave (a write-back of the
    // L2's dirty lines - few here - and a wait for every store) puts them where the last workgroup's sc1 loads read.
    // (32 write-through stores per thread instead cost 6 us of issue: the vector memory queue backs up behind them.)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (tid == 0) {
        const uint32_t before = __hip_atomic_fetch_add(a.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = before + 1u == (uint32_t)NW ? 1u : 0u;
        s_nm = 0;
    }
    __syncthreads();
    if (w == 0) RD_SB_STAMP(3);
    if (!s_last) return;
    // ---- the last workgroup: the window (every piece's words are in place), for the mirrors it is already out ----
    if (tid == 0) __hip_atomic_store(a.sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (the next launch counts from 0)
    for (int i = tid; i < nwin; i += T) s_win[i] = __hip_atomic_load(&a.win_out[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    RD_SBL_STAMP(4);
    // ---- search, one 32-position word per thread; positions 0 .. B ----
    for (int o = tid; o <= nbw; o += T) {
        uint32_t m = 0xFFFFFFFFu;
#pragma unroll
        for (int k = 0; k < P_; k++) {
            const uint32_t x = rd_lds_bits32(s_win, nwin, 32 * o + k * S_);
            m &= ((PRE_ >> k) & 1) ? x : ~x;
        }
        if (o == nbw) m &= 1u;  // position B only
        while (m) {
            const int bpos = __builtin_ctz(m);
            m &= m - 1;
            const uint32_t slot = atomicAdd(&s_nm, 1u);
            s_match[slot] = 32 * o + bpos;  // (at most B + 1 of them)
        }
    }
    __syncthreads();
    RD_SBL_STAMP(5);
    const uint32_t nm = s_nm;
    // ---- slice: a wave per match decides whether the match is superseded (a neighbour's identical bytes come first) ----
    rd_packet *recs = a.recs_host;
    auto bytes_of = [&](int pos, bool &superseded) -> uint32_t {   // lane bi < 10: byte bi of the packet at `pos`
        uint32_t byte = 0;
        bool same_prev = true, same_next = true;
        for (int r = 0; r * 64 < K_; r++) {
            const int k = 64 * r + lane;
            const uint32_t tri = k < K_ ? rd_lds_bits32(s_win, nwin, pos - 1 + k * S_) : 0u;
            const uint64_t mp = __ballot((tri & 1u) != 0), mm = __ballot((tri & 2u) != 0), mn = __ballot((tri & 4u) != 0);
            same_prev &= mp == mm;
            same_next &= mn == mm;
            const int bi = lane - 8 * r;
            if (bi >= 0 && bi < 8) {
                const int have = K_ - 8 * lane;
                const uint32_t rev = __builtin_bitreverse32((uint32_t)((mm >> (8 * bi)) & 0xFF)) >> 24;
                byte = have >= 8 ? rev : have > 0 ? rev >> (8 - have) : 0u;
            }
        }
        const uint32_t phs = (uint32_t)pos % (uint32_t)S_;
        const bool prev_first = S_ == 1 || phs != 0, next_first = S_ > 1 && phs == S_ - 1;
        superseded = (same_prev && pos >= 1 && prev_first) || (same_next && pos + 1 <= B && next_first);
        return byte;
    };
    for (uint32_t i = (uint32_t)wave; i < nm; i += T / 64) {
        const int pos = __builtin_amdgcn_readfirstlane(s_match[i]);
        bool superseded;
        (void)bytes_of(pos, superseded);
        if (superseded) {
            rd_store_void(nullptr, &recs[i], lane);
            if (lane == 0) s_match[i] = pos | 0x40000000;
        }
    }
    __syncthreads();
    // ---- RSSI / SNR of the others, one after the other, the whole workgroup on each (rd_rssi_f64's sums: |f|^2 over
    // [q - PL, q) and [q, q + PL), clipped to the newest block's filtered outputs, py:216-246).  A thread takes NINE
    // consecutive outputs: their 17 samples are asked for at once and slide through the taps - one round trip to where
    // the other workgroups' ring stores went, where an output per lane and trip (rd_rssi_f64) makes one per 64 outputs.
    const long vfrom = a.seen_before <= 0 ? 0 : a.seen_before == 1 ? -(long)B : -(long)(B + 16);
    const double *cur = ring + 32 + 2 * (size_t)B;   // sample 0 of the newest block
    uint32_t parity = 0;
    for (uint32_t i = 0; i < nm; i++) {
        const int pm = s_match[i];   // (uniform: every thread reads the same word)
        if (pm & 0x40000000) continue;
        const long q = pm;
        const long ns = q - a.cfg.PL < 0 ? 0 : q - a.cfg.PL;
        const long pe = q + a.cfg.PL > B + 1 ? B + 1 : q + a.cfg.PL;
        double noise = 0.0, sig = 0.0;
        constexpr int R = 9;
        for (long j0 = ns + (long)tid * R; j0 < pe; j0 += (long)T * R) {
            // outputs j0 .. j0 + 8 are f[j0 - 1 .. j0 + 7]: samples j0 - 10 .. j0 + 6
            const double c[9] = {RD_C0, RD_C1, RD_C2, RD_C3, RD_C4, RD_C3, RD_C2, RD_C1, RD_C0};
            rd_d2 y[R + 8];
#pragma unroll
            for (int k = 0; k < R + 8; k++) {
                const long n = j0 - 10 + k;
                y[k] = rd_d2{0.0, 0.0};
                if (n >= vfrom && n < B) y[k] = rd_rot_f64(rd_load_coh(cur + 2 * n), rd_load_coh(cur + 2 * n + 1), n);
            }
#pragma unroll
            for (int r = 0; r < R; r++) {
                const long j = j0 + r;
                rd_d2 f = {0.0, 0.0};
                if (j - 1 >= vfrom) {
#pragma unroll
                    for (int m = 0; m < 9; m++) {
                        f.x += c[m] * y[r + m].x;
                        f.y += c[m] * y[r + m].y;
                    }
                }
                const double p = f.x * f.x + f.y * f.y;
                if (j < pe) { if (j < q) noise += p; else sig += p; }
            }
        }
        noise = rd_wave_sum(noise);
        sig = rd_wave_sum(sig);
        if (lane == 0) { s_part[parity][0][wave] = noise; s_part[parity][1][wave] = sig; }
        __syncthreads();
        if (wave == 0) {
            bool superseded;
            const uint32_t byte = bytes_of((int)q, superseded);
            double rssi = 0.0, snr = 0.0;
            if (lane == 0) {
                double nsum = 0.0, ssum = 0.0;
#pragma unroll
                for (int k = 0; k < T / 64; k++) { nsum += s_part[parity][0][k]; ssum += s_part[parity][1][k]; }
                const double noise_power = (q > ns) ? nsum / (double)(q - ns) : 1e-9;
                const double signal_power = (pe > q) ? ssum / (double)(pe - q) : __builtin_nan("");
                rssi = signal_power > 0 ? 10.0 * log10(signal_power) : -120.0;
                snr = noise_power > 0 ? 10.0 * log10(signal_power / noise_power) : 50.0;
            }
            rd_store_record(nullptr, &recs[i], lane, 0, (long)a.seen_before, q, a.cfg.nbytes, byte, rssi, snr);
        }
        parity ^= 1u;   // (the next match's partial sums go to the other set: one barrier per match)
    }
    // ---- count, fence, flag ----
    __syncthreads();
    if (tid == 0) {
        a.cnt_host[0] = nm;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // system scope: records and count before the flag
        __hip_atomic_store(&a.flag_host[0], a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    RD_SBL_STAMP(6);
}

int rd_launch_stream_block_cplx(const rd_sbc_args &a_in, hipStream_t st) {
    const rd_devcfg &c = a_in.cfg;
    if (!(c.S == 14 && c.P == 16 && c.K == 80 && c.pre_mask == 0x91D3ull) || c.L != 2 * c.B || c.B % 32 || c.B < 2048 || c.B > 8192 ||
        !a_in.sync)
        return 0;
    rd_sbc_args a = a_in;
    a.stamps = nullptr;
#ifdef RD_DIAG
    a.stamps = rd_sb_stamp_buffer();
#endif
    constexpr int C = 8 * RD_SBC_THREADS;
    const int chunks = rd_sbc_chunk_host(C + 16) + 1;
    const size_t lds = (size_t)(2 * c.B / 32) * 4 + std::max((size_t)chunks * 16, ((size_t)c.B + 1) * 4);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)k_stream_block_cplx<14, 16, 0x91D3ull, 80>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL((k_stream_block_cplx<14, 16, 0x91D3ull, 80>), dim3((unsigned)((c.B + C - 1) / C)), dim3(RD_SBC_THREADS), lds, st, a);
    return 1;
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

void rd_launch_cplx_slice(const rd_cplx_layout &lay, const uint32_t *bits, long n_bits, const rd_devcfg &cfg,
                          const rd_match *matches, uint32_t match_cap, int call, rd_packet *recs, rd_packet *recs_host,
                          uint32_t *counters, hipStream_t st) {
    rd_cplx_src src;
    src.v = rd_cplx_view{lay.x, lay.valid_from, lay.n};
    hipLaunchKernelGGL(k_slice_rssi<rd_cplx_src>, dim3(rd_slice_grid(match_cap)), dim3(256), 0, st, src, bits,
                       (size_t)0, (n_bits + 31) / 32, cfg, matches, match_cap, 0, 0, call, recs, recs_host, counters);
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
