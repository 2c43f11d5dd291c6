// rd_demod_mfma.hip - k_demod_mfma: the fused IQ -> sign-bit kernel with the FIR on the matrix pipe.
//
// Same contract as k_demod_bits (rd_kernels.hip): reads lay.iq once, writes the packed sign bits and
// appends the 8-sample groups whose sign is not certain to the fix-up list (k_fixup re-evaluates those
// exactly).  Reference stages: LUT py:38-39 + rotate_fs4 py:46-49 + fir9 py:71-73 + discriminate
// numerator py:89 + quantize py:98 (py = /root/reference/src/rtldavis/dsp.py).  Arithmetic and its
// error bound: rd_mfma.h.
//
// One wave = one 2048-sample tile per iteration.  Column n = lane & 31 owns samples a0 .. a0+63
// (a0 = tile + 64 n), its window is the 144 bytes from 16 bytes before them; lane half h = lane >> 5
// supplies bytes 8h..8h+7 of every 16-byte k-step of that window and receives, per 16-output block
// b = 0..3, re/im of g[a0 + 16 b + 8 h + 1 + r], r = 0..7.  A lane therefore decides the eight signs
// of the aligned group [a0 + 16 b + 8 h, +8): six numerators from its own registers, the first two
// with g[base-1], g[base] of the lane that precedes it in time (other half, same or previous block;
// for block 0 of half 0 the last block of the previous column), exchanged through a per-wave LDS
// buffer.  A wave works through `chunk` consecutive tiles so that the very first group of a tile finds
// its predecessors in the previous iteration; at the start of a chunk (and of a stream) that one
// group is put on the fix-up list instead.
#include <cstdlib>
#include <cstring>

#include <hip/hip_ext.h>

#include "rd_internal.h"
#include "rd_math.h"
#include "rd_mfma.h"

#define RD_MF_WG 256
#define RD_MF_WAVES (RD_MF_WG / 64)
// LDS image of a tile, wave-private: [16 B: the chunk before the tile][4 groups of 72 slots]; group i
// holds the 64 chunks (16 B) of columns 8i..8i+7, chunk e of column n' at slot n' + 8 e.  One
// global_load_lds_dwordx4 fills a group (lane l supplies source chunk 8 (l & 7) + (l >> 3): a
// contiguous KiB).  A lane reads chunk e of its column at own + 128 e for every e (uniform immediates),
// and for a fixed e the 32 lanes of a half-wave fall on 16 different 8-byte bank pairs twice
// (groups 0/2 against 1/3: the 72-slot group stride is 8 mod 16): a 2-way conflict, the minimum for
// 32 lanes that all read the same half of their 16-byte chunks.
#define RD_MF_GROUP_BYTES 1152
#define RD_MF_IMG_BYTES (16 + 4 * RD_MF_GROUP_BYTES)   // 4624
#define RD_MF_IMG_PAD 4624
// predecessor exchange: three buffers of 64 x 16 B, then two carry slots (tile parity)
// + the offset constant (four copies); a multiple of 32 so that XOR 16 toggles between the two carry slots
#define RD_MF_XB_BYTES (3 * 1024 + 32 + 32)
#define RD_MF_PEND 32
// tiles whose words are stored together (a multiple of 4 that divides the default chunk)
#ifndef RD_MF_STAGE_TILES
#define RD_MF_STAGE_TILES 4
#endif

typedef _Float16 rd_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 rd_h2 __attribute__((ext_vector_type(2)));
typedef float rd_f16v __attribute__((ext_vector_type(16)));
typedef uint32_t rd_u4v __attribute__((ext_vector_type(4)));
typedef float rd_f4v __attribute__((ext_vector_type(4)));
typedef uint32_t rd_u2v __attribute__((ext_vector_type(2)));

__device__ const rd_mf_taps g_mf_taps = rd_mf_make_taps();
static const rd_mf_taps h_mf_taps = rd_mf_make_taps();

extern "C" void rd_debug_mfma_taps(uint16_t *out) { memcpy(out, &h_mf_taps, sizeof h_mf_taps); }

// the lane's 8 window bytes of one k-step -> B fragment (element order RD_MF_ELEM).  A byte in the low
// half of a 16-bit lane IS the f16 subnormal k * 2^-24: isolating the bytes is the whole conversion.
__device__ __forceinline__ rd_h8 rd_mf_frag(rd_u2v d) {
    rd_u4v v;
    v.x = d.x & 0x00FF00FFu;                                // bytes 0, 2
    v.y = __builtin_amdgcn_perm(0u, d.x, 0x0c030c01u);      // bytes 1, 3 (selector 0x0c = 0x00)
    v.z = d.y & 0x00FF00FFu;
    v.w = __builtin_amdgcn_perm(0u, d.y, 0x0c030c01u);
    return __builtin_bit_cast(rd_h8, v);
}

// -(ar cr + ai ci): numerator of py:89 for n = (ar, ai), n+ = (cr, ci) in the g frame; t1 = ai ci, the product
// whose size sets the part of the error bound that scales with the products (rd_mfma.h).
__device__ __forceinline__ float rd_mf_num(float ar, float ai, float cr, float ci, float &t1) {
    t1 = ai * ci;
    return __builtin_fmaf(-ar, cr, -t1);
}
// The guard value of a GROUP: min |num| - 2^-21 max |t1| <= |num_i| - 2^-21 |t1_i| for every sample of the group,
// so a group that passes with it passes sample by sample (rd_mfma.h); two three-operand min / max per pair of
// samples and one fma per group instead of an fma per sample.
__device__ __forceinline__ float rd_mf_guard(float nmin, float tmax) { return __builtin_fmaf(-4.76837158e-7f, tmax, nmin); }
// min(m, a, b) without abs (r may be negative: then the group is inside the band anyway)
__device__ __forceinline__ float rd_min3(float m, float a, float b) {
    float o;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(o) : "v"(m), "v"(a), "v"(b));
    return o;
}


// LDS traffic of the predecessor exchange and of the pending list goes through inline asm: the compiler
// cannot tell that these addresses do not alias the image the LDS-DMA of the NEXT tile is writing, and
// would drain vmcnt (i.e. wait for the prefetch) in front of every compiler-visible LDS access.
template <int OFF>
__device__ __forceinline__ void rd_lds_write16(uint32_t addr, rd_f4v v) {
    asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(addr), "v"(v), "i"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ rd_f4v rd_lds_read16(uint32_t addr) {  // result valid after rd_lds_wait
    rd_f4v r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "i"(OFF) : "memory");
    return r;
}
__device__ __forceinline__ void rd_lds_wait(rd_f4v &r) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r) : : "memory"); }
__device__ __forceinline__ void rd_lds_write4(uint32_t addr, uint32_t v) {
    asm volatile("ds_write_b32 %0, %1" : : "v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ rd_u4v rd_lds_read16u(uint32_t addr) {
    rd_u4v r;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r) : "v"(addr) : "memory");
    return r;
}
__device__ __forceinline__ uint32_t rd_lds_read4(uint32_t addr) {
    uint32_t r;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r) : "v"(addr) : "memory");
    return r;
}
__device__ __forceinline__ uint32_t rd_lds_addr(const void *p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)p;
}

struct rd_mf_state {
    uint32_t W;       // sign bits, first sample at the top (reversed at the end)
    uint32_t fbytes;  // byte b != 0: block b's group is inside the guard band
    float g0r, g0i;   // block 0's first output: its two boundary numerators come last
};

// Guard band in two steps (bound: rd_mfma.h).  Per sample r = |num| - 2^-21 |b d| takes care of the part of
// the error that scales with the products; what is left, 4 E0 F + const, needs F = the largest |component|
// involved.  The common path compares a group's min r with that term at F = the largest |g| ANY input can
// produce (RD_MF_C0_MAX, 2.6e-4 byte units squared: ~1e-5 of the samples of a noise input are below it);
// only when some lane of the wave fails that test - a wave-uniform branch - is F taken from the values at
// hand (v_max3 over the block) and the test repeated with it.
__device__ __forceinline__ bool rd_mf_any(bool c) { return __ballot(c) != 0; }

// One 16-output block of the tile: 6 MFMAs, the digit combine, then this lane's group of 8 signs.
// xw + WOFF: LDS address this lane's (g6, g7) go to; xr: where its predecessors' are.
template <int B, int DBG, int WOFF>
__device__ __forceinline__ void rd_mf_block(const rd_h8 (&Ahi)[3], const rd_h8 (&Alo)[3], const rd_u2v (&D)[9],
                                            rd_h8 (&bf)[3], uint32_t dc_addr, uint32_t xw, uint32_t xr,
                                            rd_mf_state &st, float *dg, int dleft) {
    const rd_f16v zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // the hi accumulator starts at -D_hi in all sixteen positions (four broadcast reads, in flight under
    // the fragment preparation)
    rd_f4v c0, c1, c2, c3;
    if (DBG != 4)
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4\n\tds_read_b128 %2, %4\n\tds_read_b128 %3, %4"
                     : "=&v"(c0), "=&v"(c1), "=&v"(c2), "=&v"(c3) : "v"(dc_addr) : "memory");
    // fragments of k-steps 2B, 2B+1, 2B+2 of the window: the first one is the previous block's last
    if (B == 0) bf[0] = rd_mf_frag(D[0]); else bf[0] = bf[2];
    bf[1] = rd_mf_frag(D[2 * B + 1]);
    bf[2] = rd_mf_frag(D[2 * B + 2]);
    // The six MFMAs go out as one burst with no vector instruction between them: a wave then sits in the
    // matrix pipe's queue for ~192 cycles while the other waves of the SIMD issue their VALU work, instead
    // of every wave stalling at an MFMA every few instructions (in-order issue: measured 1450 -> ... cycles/tile)
    __builtin_amdgcn_sched_barrier(0);
    rd_f16v ah, al;
    if (DBG == 4) {  // ablation: no matrix pipe, the vector work on stand-in values
        const rd_u4v q0 = __builtin_bit_cast(rd_u4v, bf[0]), q1 = __builtin_bit_cast(rd_u4v, bf[1]),
                     q2 = __builtin_bit_cast(rd_u4v, bf[2]);
#pragma unroll
        for (int i = 0; i < 16; i++) {
            ah[i] = __builtin_bit_cast(float, (i & 8 ? q1 : q0)[i & 3] | 0x3f000000u) + (float)i;
            al[i] = __builtin_bit_cast(float, (i & 8 ? q2 : q1)[i & 3] | 0x3f000000u);
        }
    } else {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : : "memory");
        const rd_f16v dc = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w, c2.x, c2.y, c2.z, c2.w, c3.x, c3.y, c3.z, c3.w};
        ah = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ahi[0], bf[0], dc, 0, 0, 0);  // C = -D_hi: the -127.4 offset
        al = __builtin_amdgcn_mfma_f32_32x32x16_f16(Alo[0], bf[0], zero, 0, 0, 0);
        ah = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ahi[1], bf[1], ah, 0, 0, 0);
        al = __builtin_amdgcn_mfma_f32_32x32x16_f16(Alo[1], bf[1], al, 0, 0, 0);
        ah = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ahi[2], bf[2], ah, 0, 0, 0);
        al = __builtin_amdgcn_mfma_f32_32x32x16_f16(Alo[2], bf[2], al, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (DBG == 5) {  // ablation: the matrix pipe with next to no vector work behind it
        st.W ^= __builtin_bit_cast(uint32_t, ah[0] + al[15]);
        return;
    }
    float g[16];  // g[2r], g[2r+1] = re, im of output r of this lane's group
#pragma unroll
    for (int i = 0; i < 16; i++) g[i] = __builtin_fmaf(ah[i], 2048.0f, al[i]);
    if (DBG == 3) {
#pragma unroll
        for (int r = 0; r < 8; r++)  // the tile's last output (column 31, half 1, r = 7) belongs to the next tile
            if (16 * B + r < dleft) { dg[2 * (16 * B + r)] = g[2 * r]; dg[2 * (16 * B + r) + 1] = g[2 * r + 1]; }
    }
    {
        const rd_f4v x = {g[12], g[13], g[14], g[15]};
        rd_lds_write16<WOFF>(xw, x);
    }
    rd_f4v p = {0.0f, 0.0f, 0.0f, 0.0f};
    if (B > 0) p = rd_lds_read16<0>(xr);  // g[base-1], g[base]: in flight under the group's own work
    float nmin = 3.0e38f, tmax = 0.0f;
    uint32_t w6 = 0;
#pragma unroll
    for (int q = 2; q < 8; q += 2) {
        float ta, tb;
        const float na = rd_mf_num(g[2 * q - 4], g[2 * q - 3], g[2 * q - 2], g[2 * q - 1], ta);
        const float nb = rd_mf_num(g[2 * q - 2], g[2 * q - 1], g[2 * q], g[2 * q + 1], tb);
        nmin = rd_min3abs(nmin, na, nb);
        tmax = rd_max3abs(tmax, ta, tb);
        w6 = rd_shift_in_sign(w6, na);
        w6 = rd_shift_in_sign(w6, nb);
    }
    if (B == 0) {
        st.g0r = g[0]; st.g0i = g[1];  // its two boundary numerators follow at the end of the tile
        st.W = w6;
    } else {
        rd_lds_wait(p);
        float t0, t1;
        const float n0 = rd_mf_num(p.x, p.y, p.z, p.w, t0);
        const float n1 = rd_mf_num(p.z, p.w, g[0], g[1], t1);
        nmin = rd_min3abs(nmin, n0, n1);
        tmax = rd_max3abs(tmax, t0, t1);
        uint32_t w2 = rd_shift_in_sign(0u, n0);
        w2 = rd_shift_in_sign(w2, n1);
        st.W = (st.W << 8) | (w2 << 6) | w6;
    }
    if (DBG == 0 || DBG == 3) {
        const float nm = rd_mf_guard(nmin, tmax);
        if (rd_mf_any(!(nm > RD_MF_C0_MAX))) {  // rare; NaN counts as inside
            float F = 0.0f;
#pragma unroll
            for (int r = 0; r < 8; r++) F = rd_max3abs(F, g[2 * r], g[2 * r + 1]);
            if (B > 0) {
                F = rd_max3abs(F, p.x, p.y);
                F = rd_max3abs(F, p.z, p.w);
            }
            if (!(nm > rd_mf_c0(F))) st.fbytes |= 1u << (8 * B);
        }
    }
}

// cache policy of the tile loads (the builtin's aux operand): 0 default, 2 = nt.  The input is streamed
// once and never re-read: nt loads-only 0.330 ms (6.7 TB/s) against 0.358, loads + stores 0.427 against 0.462,
// whole kernel 0.496 against 0.509, and the search kernel behind it finds more of the bits in cache.
#ifndef RD_MF_LOAD_AUX
#define RD_MF_LOAD_AUX 2
#endif

__device__ __forceinline__ void rd_mf_issue(const rd_layout &lay, uint32_t s, uint32_t ti, uint8_t *img, int lane) {
    const uint8_t *src = lay.iq + (size_t)s * lay.stream_stride + (size_t)ti * RD_TILE_BYTES;
    const int perm = 8 * (lane & 7) + (lane >> 3);
    const __attribute__((address_space(1))) void *g0 = (const __attribute__((address_space(1))) void *)(src + perm * 16);
    // the instruction offset advances the global and the LDS address alike; the LDS base makes up
    // the difference between the 1024-byte source groups and the 1152-byte image groups
    __builtin_amdgcn_global_load_lds(g0, (__attribute__((address_space(3))) void *)(img + 16), 16, 0, RD_MF_LOAD_AUX);
    __builtin_amdgcn_global_load_lds(g0, (__attribute__((address_space(3))) void *)(img + 16 + 128), 16, 1024, RD_MF_LOAD_AUX);
    __builtin_amdgcn_global_load_lds(g0, (__attribute__((address_space(3))) void *)(img + 16 + 256), 16, 2048, RD_MF_LOAD_AUX);
    __builtin_amdgcn_global_load_lds(g0, (__attribute__((address_space(3))) void *)(img + 16 + 384), 16, 3072, RD_MF_LOAD_AUX);
    // the 16 bytes before the tile (previous tile, or the caller's history).  With zero history there
    // is nothing to read: the first run of the stream is re-evaluated exactly anyway.
    const bool has_halo = (ti > 0) || lay.hist_mode;
    if (lane == 0)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (has_halo ? -16 : 0)),
                                         (__attribute__((address_space(3))) void *)(img), 16, 0, 0);
}

__device__ __forceinline__ void rd_mf_flush(const uint32_t *pend, uint32_t count, uint32_t *fix_list, uint32_t fix_cap,
                                            uint32_t *counters, int lane) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&counters[RD_CNT_FIX], count);
    base = __builtin_amdgcn_readfirstlane(base);
    for (uint32_t i = lane; i < count; i += 64)
        if (base + i < fix_cap) fix_list[base + i] = pend[i];
}

// Store the staged words: four tiles as one 16-byte store per lane, fewer tile by tile.
__device__ __forceinline__ void rd_mf_store_staged(uint32_t stage_addr, uint32_t nst, uint32_t *base, int lane,
                                                   uint32_t stflags) {
    if (nst == RD_MF_STAGE_TILES) {
#pragma unroll
        for (int j = 0; j < RD_MF_STAGE_TILES / 4; j++) {
            const rd_u4v v = rd_lds_read16u(stage_addr + 16 * (lane + 64 * j));
            // non-temporal: the words are read next by another kernel, never again by this one (0.3-1.3 % faster;
            // RD_K1_STFLAGS & 1 switches to plain stores for A/B runs)
            if (!(stflags & 1)) __builtin_nontemporal_store(v, (rd_u4v *)(base + 4 * (lane + 64 * j)));
            else *(rd_u4v *)(base + 4 * (lane + 64 * j)) = v;
        }
    } else {
        for (uint32_t q = 0; q < nst; q++) base[64 * q + lane] = rd_lds_read4(stage_addr + 256 * q + 4 * lane);
    }
}

// Position of a wave in its sequence of tiles: chunks of `chunk` consecutive tiles.  A wave's first chunk is its
// own number; the following ones come from work queues (rd_internal.h: RD_NQUEUE): waves do not run equally fast - the
// workgroups a CU received first issue ahead of the later ones and get through a tile in 2.5 us where the last
// ones need 3.9 - so equal shares leave the fast ones idle for the last fifth of the launch.
struct rd_mf_pos {
    uint32_t tile, s, ti, inchunk;
};
// next_tile: first tile of the chunk that follows p's (>= total when there is none)
struct rd_mf_nextchunk {
    uint32_t tile, s, ti;  // first tile of the chunk that follows the current one (tile >= total: none), split
};
__device__ __forceinline__ rd_mf_nextchunk rd_mf_chunk_at(uint64_t chunk_id, uint32_t chunk, uint32_t tps, uint32_t total) {
    rd_mf_nextchunk c;
    const uint64_t nt = chunk_id * chunk;
    c.tile = nt < total ? (uint32_t)nt : 0xFFFFFFFFu;
    c.s = 0; c.ti = 0;
    if (nt < total) {  // (the one division per chunk: done at the loop top, where few registers are live)
        c.s = __builtin_amdgcn_readfirstlane(c.tile / tps);
        c.ti = c.tile - c.s * tps;
    }
    return c;
}
__device__ __forceinline__ rd_mf_pos rd_mf_next(rd_mf_pos p, uint32_t chunk, uint32_t tps, const rd_mf_nextchunk &nc) {
    rd_mf_pos q;
    if (p.inchunk + 1 < chunk) {
        q.tile = p.tile + 1; q.inchunk = p.inchunk + 1;
        q.s = p.s; q.ti = p.ti + 1;
        if (q.ti >= tps) { q.ti = 0; q.s++; }
    } else {
        q.tile = nc.tile; q.inchunk = 0;
        q.s = nc.s; q.ti = nc.ti;
    }
    return q;
}

// The nine 8-byte window pieces of a lane.  Inline asm for the same reason as the exchange buffer: a
// compiler-visible LDS read would wait for vmcnt(0), i.e. for the tiles that are being prefetched into
// the OTHER image buffer.
__device__ __forceinline__ void rd_mf_read_window(uint32_t a_prv, uint32_t a_own, rd_u2v (&D)[9]) {
    asm volatile("ds_read_b64 %0, %9\n\t"
                 "ds_read_b64 %1, %10\n\t"
                 "ds_read_b64 %2, %10 offset:128\n\t"
                 "ds_read_b64 %3, %10 offset:256\n\t"
                 "ds_read_b64 %4, %10 offset:384\n\t"
                 "ds_read_b64 %5, %10 offset:512\n\t"
                 "ds_read_b64 %6, %10 offset:640\n\t"
                 "ds_read_b64 %7, %10 offset:768\n\t"
                 "ds_read_b64 %8, %10 offset:896\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(D[0]), "=&v"(D[1]), "=&v"(D[2]), "=&v"(D[3]), "=&v"(D[4]), "=&v"(D[5]), "=&v"(D[6]),
                   "=&v"(D[7]), "=&v"(D[8])
                 : "v"(a_prv), "v"(a_own)
                 : "memory");
}

// DBG: 0 product; 1 no global loads; 2 loads + LDS reads only; 3 also dumps g (dbg_g[tile][2048][2],
// sample order); 4 = 1 without the MFMAs; 5 = 1 with the MFMAs and almost no vector work - 1, 2, 4, 5 are
// timing ablations with garbage results.
template <int DBG, int NBUF>
__global__ __launch_bounds__(RD_MF_WG, 2) void k_demod_mfma(rd_layout lay, uint32_t tiles_per_stream, uint32_t total_tiles,
                                                         uint32_t chunk, uint32_t *fix_list, uint32_t fix_cap,
                                                         uint32_t *counters, float *dbg_g, uint32_t stflags) {
    // NBUF = 1 (default): one image buffer, tile i+1 in flight while tile i is computed, 4 workgroups per CU.
    // NBUF = 2 (RD_K1_NBUF=2): two buffers, tiles i+1 and i+2 in flight, 3 workgroups per CU - measured slower
    // (0.54 vs 0.53 ms: the fourth wave per SIMD is worth more than the second tile in flight).
    __shared__ __attribute__((aligned(16))) uint8_t s_img[RD_MF_WAVES][NBUF][RD_MF_IMG_PAD];
    __shared__ __attribute__((aligned(32))) uint8_t s_xb[RD_MF_WAVES][RD_MF_XB_BYTES];
    __shared__ uint32_t s_pend[RD_MF_WAVES][RD_MF_PEND];
    // packed words of up to four consecutive tiles of a stream, stored together: one 16-byte store per lane
    // (1 KiB contiguous per wave) instead of four dword stores (round 1: a dword store per tile cost 20 % of
    // the read bandwidth, profiles/r01_ubench_read_bw.txt)
    __shared__ __attribute__((aligned(16))) uint32_t s_stage[RD_MF_WAVES][RD_MF_STAGE_TILES][64];
    constexpr bool LOADS = DBG != 1 && DBG != 4 && DBG != 5;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint8_t *img0 = s_img[wave][0];
    uint8_t *xb = s_xb[wave];
    uint32_t *mypend = s_pend[wave];
    uint32_t npend = 0;

    const int n = lane & 31, h = lane >> 5;
    // tap fragments: 6 x 4 registers for the whole kernel
    rd_h8 Ahi[3], Alo[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        Ahi[d] = *(const rd_h8 *)g_mf_taps.v[0][d][lane];
        Alo[d] = *(const rd_h8 *)g_mf_taps.v[1][d][lane];
    }
    // a use in front of the loop: the wait for these six loads must not end up inside it, where it would
    // be a vmcnt(0) that also drains the prefetched tiles in every iteration
    asm volatile("" : "+v"(Ahi[0]), "+v"(Ahi[1]), "+v"(Ahi[2]), "+v"(Alo[0]), "+v"(Alo[1]), "+v"(Alo[2]));
    // window addresses in the image (buffer 0; buffer 1 is RD_MF_IMG_PAD further)
    const uint32_t img_addr = rd_lds_addr(img0);
    const uint32_t own = 16 + RD_MF_GROUP_BYTES * (n >> 3) + 16 * (n & 7) + 8 * h;
    const uint32_t prv = n == 0 ? 8 * h
                       : (n & 7) ? own + 16 * 55
                                 : 16 + RD_MF_GROUP_BYTES * ((n >> 3) - 1) + 16 * 63 + 8 * h;
    // Predecessor exchange.  Step b writes (g6, g7) of block b; the lane that follows in time is
    // (n, 1, b) after (n, 0, b), (n, 0, b) after (n, 1, b-1), (n, 0, 0) after (n-1, 1, 3), and (0, 0, 0) after
    // lane 63 of the previous tile.  Buffers: X0 at 0, X1 and X3 at 1024, X2 at 2048 (X1 is dead when X3 is
    // written), two carry slots at 3072 (lane 63's X3, alternating with the tile parity).
    const uint32_t xb_addr = rd_lds_addr(xb), pend_addr = rd_lds_addr(mypend);
    const uint32_t stage_addr = rd_lds_addr(s_stage[wave]);
    const uint32_t xw = xb_addr + 16 * lane;                       // + 0 / 1024 / 2048 for b = 0 / 1 / 2
    uint32_t xw3 = xb_addr + (lane == 63 ? 3072 : 1024 + 16 * lane);
    const uint32_t xr1 = xb_addr + (h ? 1024 + 16 * (lane - 32) : 16 * (lane + 32));          // X1[l-32] | X0[l+32]
    const uint32_t xr2 = xb_addr + (h ? 2048 + 16 * (lane - 32) : 1024 + 16 * (lane + 32));   // X2[l-32] | X1[l+32]
    const uint32_t xr3 = xb_addr + (h ? 1024 + 16 * (lane - 32) : 2048 + 16 * (lane + 32));   // X3[l-32] | X2[l+32]
    uint32_t xr0 = xb_addr + (h ? 16 * (lane - 32) : lane ? 1024 + 16 * (lane + 31) : 3072 + 16);  // X0[l-32] | X3[l+31]
    const uint32_t rtoggle = lane == 0 ? 16u : 0u, wtoggle = lane == 63 ? 16u : 0u;
    if (lane < 8) ((uint32_t *)(xb + 3072))[lane] = 0;  // carry slots: finite values from the start
    // -D_hi * 2^-24 (the -127.4 offset): read into all sixteen positions of the hi accumulator in front of
    // every block's MFMAs - from LDS rather than from a 16-register tuple held for the whole kernel
    if (lane < 4) ((float *)(xb + 3104))[lane] = -(float)RD_MF_DHI / 16777216.0f;
    const uint32_t psel = h ? 0x07030602u : 0x05010400u;

    const uint32_t nwaves = gridDim.x * RD_MF_WAVES;
    const uint32_t wave_id = blockIdx.x * RD_MF_WAVES + wave;
    // RD_K1_STFLAGS & 4096: every wave takes chunks wave_id, wave_id + nwaves, ... (equal shares: A/B); the queue
    // needs chunks of at least four tiles (the id of the following chunk is asked for in a chunk's first
    // iteration, read in its second and first used in its last but one)
    const bool dynamic = !(stflags & 4096) && chunk >= 4 && NBUF == 1;
    uint32_t cur_chunk = wave_id;      // the chunk `cur` is in
    rd_mf_nextchunk nextc = {0xFFFFFFFFu, 0, 0};
    if (!dynamic) nextc = rd_mf_chunk_at((uint64_t)cur_chunk + nwaves, chunk, tiles_per_stream, total_tiles);
    uint32_t grab = 0;                 // lane 0: what the atomic returned (valid one loop-top wait after its issue)
    bool grab_pending = false;
    const uint32_t my_queue = wave_id % RD_NQUEUE;
    uint32_t *queue = &counters[RD_CNT_QUEUE0 + RD_QUEUE_STRIDE * my_queue];
    rd_mf_pos cur;
    cur.tile = wave_id * chunk;
    cur.s = cur.tile / tiles_per_stream;
    cur.ti = cur.tile % tiles_per_stream;
    cur.inchunk = 0;
    rd_mf_pos nx1 = rd_mf_next(cur, chunk, tiles_per_stream, nextc);
    uint32_t buf = 0;  // image buffer of the current tile (wave-uniform)

    uint32_t nst = 0;             // tiles staged (wave-uniform)
    uint32_t *st_base = nullptr;  // word 0 of the first staged tile
    bool st_flush = false;        // the staged group ends here (next tile is not the next 64 words)
    uint32_t rg_word = 0;         // a ragged last tile is stored word by word, predicated
    uint32_t *rg_ptr = nullptr;
    if (LOADS && cur.tile < total_tiles) rd_mf_issue(lay, cur.s, cur.ti, img0, lane);
    if (NBUF == 2 && LOADS && nx1.tile < total_tiles) rd_mf_issue(lay, nx1.s, nx1.ti, img0 + RD_MF_IMG_PAD, lane);
    while (cur.tile < total_tiles) {
        const uint32_t tile = cur.tile, s = cur.s, ti = cur.ti, inchunk = cur.inchunk;
        // this tile has landed when at most the next tile's five loads are outstanding (vmcnt counts in
        // issue order; the previous iteration's word store and list flush are older or harmless)
        if (NBUF == 2 && nx1.tile < total_tiles) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" : "+v"(grab) : : "memory");
        if (dynamic) {
            if (grab_pending) {  // asked for one iteration ago: the wait above covers it
                nextc = rd_mf_chunk_at((uint64_t)nwaves + my_queue + (uint64_t)RD_NQUEUE * __builtin_amdgcn_readfirstlane(grab),
                                       chunk, tiles_per_stream, total_tiles);
                grab_pending = false;
            }
            if (inchunk == 0) {  // a new chunk: ask for the one after it (inline asm: the compiler would wait for the result at once)
                // lane 0 only, by way of the exec mask inside the statement (a branch around it costs 14 registers)
                const uint32_t one = 1, zoff = 0;
                uint64_t saved_exec;
                asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, 1\n\tglobal_atomic_add %0, %2, %3, %4 sc0\n\ts_mov_b64 exec, %1"
                             : "+v"(grab), "=&s"(saved_exec) : "v"(zoff), "v"(one), "s"(queue) : "memory");
                grab_pending = true;
            }
        } else if (inchunk == 0 && tile != wave_id * chunk) {  // equal shares: a new chunk was entered
            cur_chunk += nwaves;
            nextc = rd_mf_chunk_at((uint64_t)cur_chunk + nwaves, chunk, tiles_per_stream, total_tiles);
        }
        // the short stretch from "data has landed" to "next loads issued" runs at raised priority: a wave whose
        // tile arrived should not queue behind three other waves' vector work before it can ask for the next
        if (!(stflags & 4)) __builtin_amdgcn_s_setprio(3);  // (RD_K1_STFLAGS & 4 switches it off: A/B)
        rd_u2v D[9];
        const uint32_t boff = buf * RD_MF_IMG_PAD;
        rd_mf_read_window(img_addr + boff + prv, img_addr + boff + own, D);
        // the window is in registers: this buffer takes the tile after next
        // stores of finished tiles go out here, before the loads (they share vmcnt, in issue order)
        if (nst == RD_MF_STAGE_TILES || (nst && st_flush)) {
            rd_mf_store_staged(stage_addr, nst, st_base, lane, stflags);
            nst = 0;
        }
        if (rg_ptr) *rg_ptr = rg_word;
        rg_ptr = nullptr;
        const rd_mf_pos nx2 = rd_mf_next(nx1, chunk, tiles_per_stream, nextc);
        const rd_mf_pos fetch = NBUF == 2 ? nx2 : nx1;
        if (LOADS && fetch.tile < total_tiles) rd_mf_issue(lay, fetch.s, fetch.ti, img0 + boff, lane);
        if (!(stflags & 4)) __builtin_amdgcn_s_setprio(0);

        uint32_t word = 0, fbytes = 0;
        if (DBG == 2 || DBG == 6) {
#pragma unroll
            for (int j = 0; j < 9; j++) word ^= D[j].x ^ D[j].y;
        } else {
            rd_h8 bf[3];
            rd_mf_state stt;
            stt.W = 0; stt.fbytes = 0; stt.g0r = 0.0f; stt.g0i = 0.0f;
            float *dg = DBG == 3 ? dbg_g + ((size_t)tile * RD_TILE_SAMPLES + 64 * n + 8 * h + 1) * 2 : nullptr;
            const int dleft = RD_TILE_SAMPLES - (64 * n + 8 * h + 1);  // outputs of this lane inside the tile
            rd_mf_block<0, DBG, 0>(Ahi, Alo, D, bf, xb_addr + 3104, xw, 0, stt, dg, dleft);
            rd_mf_block<1, DBG, 1024>(Ahi, Alo, D, bf, xb_addr + 3104, xw, xr1, stt, dg, dleft);
            rd_mf_block<2, DBG, 2048>(Ahi, Alo, D, bf, xb_addr + 3104, xw, xr2, stt, dg, dleft);
            rd_mf_block<3, DBG, 0>(Ahi, Alo, D, bf, xb_addr + 3104, xw3, xr3, stt, dg, dleft);
            {   // block 0's first two numerators: W holds 30 bits, its bits 31, 30 are theirs
                rd_f4v p = rd_lds_read16<0>(xr0);
                rd_lds_wait(p);
                float t0, t1;
                const float n0 = rd_mf_num(p.x, p.y, p.z, p.w, t0);
                const float n1 = rd_mf_num(p.z, p.w, stt.g0r, stt.g0i, t1);
                stt.W |= __builtin_bit_cast(uint32_t, n0) & 0x80000000u;
                stt.W |= (__builtin_bit_cast(uint32_t, n1) >> 1) & 0x40000000u;
                if (DBG == 0 || DBG == 3) {
                    const float nm = rd_mf_guard(rd_min3abs(3.0e38f, n0, n1), rd_max3abs(0.0f, t0, t1));
                    if (rd_mf_any(!(nm > RD_MF_C0_MAX))) {
                        float F = rd_max3abs(0.0f, p.x, p.y);
                        F = rd_max3abs(F, p.z, p.w);
                        F = rd_max3abs(F, stt.g0r, stt.g0i);
                        if (!(nm > rd_mf_c0(F))) stt.fbytes |= 1u;
                    }
                }
            }
            xr0 ^= rtoggle;
            xw3 ^= wtoggle;
            word = __builtin_bitreverse32(stt.W);  // byte b = the signs of block b's group
            fbytes = stt.fbytes;
        }
        // The lane holds bytes (groups) 2b + h of its column's two words: gather word h of the column
        // (lanes n and n + 32 exchange halves), the flags likewise.
        uint32_t gmask = 0;
        {
            const auto w2 = __builtin_amdgcn_permlane32_swap(word, word, false, false);  // [0]: half 0's, [1]: half 1's
            word = __builtin_amdgcn_perm(w2[1], w2[0], psel);
        }
        const bool carry = inchunk > 0 && ti > 0;  // previous iteration = previous tile of this stream
        const uint32_t run = ti * 64 + 2 * n + h;  // word index in the stream
        const uint32_t t0 = run * RD_RUN;
        const bool any_flag = rd_mf_any(fbytes != 0);
        if (any_flag) {  // wave-uniform, rare
            const auto f2 = __builtin_amdgcn_permlane32_swap(fbytes, fbytes, false, false);
            const uint32_t fb = __builtin_amdgcn_perm(f2[1], f2[0], psel);
            gmask = ((fb * 0x00204081u) >> 21) & 0xFu;  // bytes 0/1 -> bits
        }
        if (lane == 0) {
            if (ti == 0 && !lay.hist_mode) gmask = 0xFu;  // zero history: first run exact
            else if (!carry) gmask |= 1u;                 // no predecessors for the tile's first group
        }
        const bool ragged = (ti + 1 == tiles_per_stream) && (lay.n_samples % RD_TILE_SAMPLES) != 0;
        if (DBG == 6) {
            if (word == 0x12345678u) lay.bits[0] = word;  // keeps the loads alive, stores nothing
        } else if (!ragged) {
            if (nst == 0) {
                st_base = &lay.bits[(size_t)s * lay.bits_stride + ti * 64];
                // diagnostic (RD_K1_STFLAGS & 2): all stores land in the first MiB of the bits array
                if (stflags & 2) st_base = &lay.bits[(((size_t)s * lay.bits_stride + ti * 64) & 0x3FFFFu) & ~255u];
            }
            rd_lds_write4(stage_addr + 256 * nst + 4 * (2 * n + h), word);  // the lane holds word 2n + h of the tile
            nst++;
            // the group goes on only if the next tile of this wave is the next 64 words of the same stream
            const bool next_ragged = (nx1.ti + 1 == tiles_per_stream) && (lay.n_samples % RD_TILE_SAMPLES) != 0;
            st_flush = !(nx1.tile < total_tiles && nx1.s == s && nx1.ti == ti + 1 && !next_ragged);
        } else if (t0 < lay.n_samples) {
            const uint32_t left = lay.n_samples - t0;
            if (left < RD_RUN) {
                word &= (1u << left) - 1u;
                gmask &= (1u << ((left + RD_GROUP - 1) / RD_GROUP)) - 1u;
            }
            rg_word = word;
            rg_ptr = &lay.bits[(size_t)s * lay.bits_stride + run];
        } else {
            gmask = 0;
        }
        if (DBG == 1 || DBG == 2 || (DBG >= 4 && DBG != 7)) gmask = 0;  // (incl. 6)
        const uint64_t fm = (any_flag || !carry) ? __ballot(gmask != 0) : 0;
        if (fm) {
            const uint32_t nf = (uint32_t)__popcll(fm);
            if (npend + nf > RD_MF_PEND) {
                rd_mf_flush(mypend, npend, fix_list, fix_cap, counters, lane);
                npend = 0;
            }
            if (gmask)
                rd_lds_write4(pend_addr + 4 * (npend + __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32),
                                                                                  __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0))),
                              ((uint32_t)((size_t)s * lay.bits_stride + run) << 4) | gmask);
            npend += nf;
        }
        cur = nx1;
        nx1 = nx2;
        if (NBUF == 2) buf ^= 1;
    }
    if (nst) rd_mf_store_staged(stage_addr, nst, st_base, lane, stflags);
    if (rg_ptr) *rg_ptr = rg_word;
    if (npend) rd_mf_flush(mypend, npend, fix_list, fix_cap, counters, lane);
}

static int rd_mf_env(const char *name, int dflt) {
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}

void rd_launch_demod_mfma(const rd_layout &lay, uint32_t *fix_list, uint32_t fix_cap, uint32_t *counters, hipStream_t st,
                          hipEvent_t ev_start, hipEvent_t ev_stop, float *dbg_g) {
    const uint32_t tps = (lay.n_samples + RD_TILE_SAMPLES - 1) / RD_TILE_SAMPLES;
    const uint64_t total64 = (uint64_t)lay.n_streams * tps;
    if (total64 == 0) return;
    const uint32_t total = (uint32_t)total64;
    static uint32_t stflags = 0;
    static int dbg = -1, chunk_env = 0, per_cu_env = 0, n_cu = 0, nbuf = 1, per_cu_occ[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (dbg < 0) {
        dbg = rd_mf_env("RD_K1_DEBUG", 0);
        chunk_env = rd_mf_env("RD_K1_CHUNK", 0);
        per_cu_env = rd_mf_env("RD_K1_WGS_PER_CU", 0);
        nbuf = rd_mf_env("RD_K1_NBUF", 1) == 2 ? 2 : 1;
        stflags = (uint32_t)rd_mf_env("RD_K1_STFLAGS", 0);
        int dev = 0;
        hipGetDevice(&dev);
        hipDeviceProp_t prop;
        n_cu = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                   ? prop.multiProcessorCount : 256;
    }
    const int variant = dbg_g ? 3 : (dbg == 1 || dbg == 2 || (dbg >= 4 && dbg <= 7)) ? dbg : 0;
    // persistent grid sized from the occupancy API (registers and LDS of the variant actually launched)
    if (!per_cu_occ[variant]) {  // (nbuf is fixed per process)
        int occ = 0;
        hipError_t e = hipErrorUnknown;
        switch (variant) {
            case 0: e = (nbuf == 1 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_demod_mfma<0, 1>, RD_MF_WG, 0) : hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_demod_mfma<0, 2>, RD_MF_WG, 0)); break;
            case 1: e = (nbuf == 1 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_demod_mfma<1, 1>, RD_MF_WG, 0) : hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_demod_mfma<1, 2>, RD_MF_WG, 0)); break;
            case 2: e = (nbuf == 1 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_demod_mfma<2, 1>, RD_MF_WG, 0) : hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_demod_mfma<2, 2>, RD_MF_WG, 0)); break;
            case 4: e = (nbuf == 1 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_demod_mfma<4, 1>, RD_MF_WG, 0) : hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_demod_mfma<4, 2>, RD_MF_WG, 0)); break;
            case 5: e = (nbuf == 1 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_demod_mfma<5, 1>, RD_MF_WG, 0) : hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_demod_mfma<5, 2>, RD_MF_WG, 0)); break;
            case 6: e = (nbuf == 1 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_demod_mfma<6, 1>, RD_MF_WG, 0) : hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_demod_mfma<6, 2>, RD_MF_WG, 0)); break;
            case 7: e = (nbuf == 1 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_demod_mfma<7, 1>, RD_MF_WG, 0) : hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_demod_mfma<7, 2>, RD_MF_WG, 0)); break;
            default: e = (nbuf == 1 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_demod_mfma<3, 1>, RD_MF_WG, 0) : hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_demod_mfma<3, 2>, RD_MF_WG, 0)); break;
        }
        per_cu_occ[variant] = (e == hipSuccess && occ >= 1) ? (occ > 8 ? 8 : occ) : 2;
    }
    // (three workgroups per CU instead of the four that fit: within the run-to-run noise, +-2 %)
    const int per_cu = (per_cu_env >= 1 && per_cu_env <= 8) ? per_cu_env : per_cu_occ[variant];
    // Tiles per chunk: a multiple of the 4-tile store groups.  With the work queues (default) 16: the chunks past a
    // wave's first are handed out on demand, the waves that run faster take more of them, and shorter chunks
    // even the finish out (0.459 ms; 28: 0.471, 12: 0.466, 8: 0.465 but a longer fix-up list).  With equal shares
    // (RD_K1_STFLAGS & 4096) 28, the best of a sweep from 8 to 66 on 4096 x 132 tiles (profiles/r02_chunk_sweep.txt).
    // A small workload gets shorter chunks, down to one store group, until every resident wave has one.
    uint32_t chunk = chunk_env > 0 ? (uint32_t)chunk_env : ((stflags & 4096) ? 28 : 16);
    if (chunk_env <= 0) {
        const uint64_t waves = (uint64_t)n_cu * per_cu * RD_MF_WAVES;
        while (chunk > RD_MF_STAGE_TILES && total64 / chunk < waves) chunk -= RD_MF_STAGE_TILES;
    }
    if (chunk > total) chunk = total;
    const uint64_t chunks = (total64 + chunk - 1) / chunk;
    uint64_t wgs = (chunks + RD_MF_WAVES - 1) / RD_MF_WAVES;
    const uint64_t max_wgs = (uint64_t)n_cu * per_cu;
    if (wgs > max_wgs) wgs = max_wgs;
#define RD_LAUNCH_MF2(D, NB)                                                                                         \
    do {                                                                                                             \
        if (ev_start || ev_stop)                                                                                     \
            hipExtLaunchKernelGGL((k_demod_mfma<D, NB>), dim3((unsigned)wgs), dim3(RD_MF_WG), 0, st, ev_start,       \
                                  ev_stop, 0, lay, tps, total, chunk, fix_list, fix_cap, counters, dbg_g, stflags);  \
        else                                                                                                         \
            hipLaunchKernelGGL((k_demod_mfma<D, NB>), dim3((unsigned)wgs), dim3(RD_MF_WG), 0, st, lay, tps, total,   \
                               chunk, fix_list, fix_cap, counters, dbg_g, stflags);                                  \
    } while (0)
#define RD_LAUNCH_MF(D) do { if (nbuf == 1) RD_LAUNCH_MF2(D, 1); else RD_LAUNCH_MF2(D, 2); } while (0)
    if (variant == 3) RD_LAUNCH_MF(3);
    else if (variant == 1) RD_LAUNCH_MF(1);
    else if (variant == 2) RD_LAUNCH_MF(2);
    else if (variant == 4) RD_LAUNCH_MF(4);
    else if (variant == 5) RD_LAUNCH_MF(5);
    else if (variant == 6) RD_LAUNCH_MF(6);
    else if (variant == 7) RD_LAUNCH_MF(7);
    else RD_LAUNCH_MF(0);
#undef RD_LAUNCH_MF
#undef RD_LAUNCH_MF2
}

// Test hook (tests/test_gpu_mfma.py): run the kernel on host data, return the raw filter outputs g
// (tile-major, [tiles][2048][2] floats in the kernel's units), the packed bits BEFORE any fix-up and
// the fix-up list.  iq_host holds n_streams x n_samples x 2 bytes; hist_bytes >= 0 bytes of history
// precede every stream when hist_mode is set (stream stride = 2 n_samples + hist_bytes).
extern "C" int rd_debug_demod_mfma(const uint8_t *iq_host, int n_streams, uint32_t n_samples, int hist_mode,
                                   uint32_t hist_bytes, float *g_out, uint32_t *bits_out, uint32_t *fix_out,
                                   uint32_t fix_cap, uint32_t *n_fix) {
    int rc = rd_ensure_device_public();
    if (rc) return rc;
    const size_t stride = (size_t)n_samples * 2 + (hist_mode ? hist_bytes : 0);
    if (stride % 16 || (hist_mode && hist_bytes % 16)) return RD_ERR_ARG;
    const uint32_t tps = (n_samples + RD_TILE_SAMPLES - 1) / RD_TILE_SAMPLES;
    const size_t words = (n_samples + 31) / 32;
    const size_t iq_bytes = stride * n_streams;
    uint8_t *d_iq = nullptr;
    uint32_t *d_bits = nullptr, *d_fix = nullptr, *d_cnt = nullptr;
    float *d_g = nullptr;
    const size_t g_floats = (size_t)n_streams * tps * RD_TILE_SAMPLES * 2;
#define RD_DBG_CHK(x) do { if ((x) != hipSuccess) { rc = RD_ERR_DEVICE; goto out; } } while (0)
    RD_DBG_CHK(hipMalloc(&d_iq, iq_bytes + RD_INPUT_PAD));
    RD_DBG_CHK(hipMemset(d_iq + iq_bytes, 127, RD_INPUT_PAD));
    RD_DBG_CHK(hipMemcpy(d_iq, iq_host, iq_bytes, hipMemcpyHostToDevice));
    RD_DBG_CHK(hipMalloc(&d_bits, words * n_streams * 4));
    RD_DBG_CHK(hipMemset(d_bits, 0, words * n_streams * 4));
    RD_DBG_CHK(hipMalloc(&d_fix, (size_t)fix_cap * 4 + 4));
    RD_DBG_CHK(hipMalloc(&d_cnt, RD_CNT_TOTAL * 4));
    RD_DBG_CHK(hipMemset(d_cnt, 0, RD_CNT_TOTAL * 4));
    RD_DBG_CHK(hipMalloc(&d_g, g_floats * 4));
    RD_DBG_CHK(hipMemset(d_g, 0, g_floats * 4));
    {
        rd_layout lay;
        lay.iq = d_iq + (hist_mode ? hist_bytes : 0);
        lay.stream_stride = stride;
        lay.n_streams = n_streams;
        lay.n_samples = n_samples;
        lay.hist_mode = hist_mode;
        lay.valid_from = hist_mode ? -(long)(hist_bytes / 2) : 0;
        lay.bits = d_bits;
        lay.bits_stride = words;
        rd_launch_demod_mfma(lay, d_fix, fix_cap, d_cnt, nullptr, nullptr, nullptr, d_g);
    }
    RD_DBG_CHK(hipDeviceSynchronize());
    RD_DBG_CHK(hipMemcpy(g_out, d_g, g_floats * 4, hipMemcpyDeviceToHost));
    RD_DBG_CHK(hipMemcpy(bits_out, d_bits, words * n_streams * 4, hipMemcpyDeviceToHost));
    {
        uint32_t cnt[RD_CNT_SLOTS];
        RD_DBG_CHK(hipMemcpy(cnt, d_cnt, sizeof cnt, hipMemcpyDeviceToHost));
        *n_fix = cnt[RD_CNT_FIX];
        const uint32_t have = cnt[RD_CNT_FIX] < fix_cap ? cnt[RD_CNT_FIX] : fix_cap;
        if (have) RD_DBG_CHK(hipMemcpy(fix_out, d_fix, (size_t)have * 4, hipMemcpyDeviceToHost));
    }
out:
#undef RD_DBG_CHK
    hipFree(d_iq); hipFree(d_bits); hipFree(d_fix); hipFree(d_cnt); hipFree(d_g);
    return rc;
}
