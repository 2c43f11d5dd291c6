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
#define RD_MF_IMG_PAD 4640
// predecessor exchange: 4 step buffers of 64 x 16 B, then two carry slots (tile parity)
#define RD_MF_XB_BYTES (1024 + 4 * 1024 + 32)  // 1 KiB in front keeps every base address non-negative
#define RD_MF_PEND 64
#define RD_MF_LDS_WAVE (RD_MF_IMG_PAD + RD_MF_XB_BYTES + RD_MF_PEND * 4)

typedef _Float16 rd_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 rd_h2 __attribute__((ext_vector_type(2)));
typedef float rd_f16v __attribute__((ext_vector_type(16)));
typedef uint32_t rd_u4v __attribute__((ext_vector_type(4)));
typedef float rd_f4v __attribute__((ext_vector_type(4)));

__device__ const rd_mf_taps g_mf_taps = rd_mf_make_taps();
static const rd_mf_taps h_mf_taps = rd_mf_make_taps();

extern "C" void rd_debug_mfma_taps(uint16_t *out) { memcpy(out, &h_mf_taps, sizeof h_mf_taps); }

// two bytes (already isolated in the low byte of each half: an f16 subnormal b * 2^-24 each) ->
// (5 b - 637) * 2^-12 as packed f16, exact: one v_pk_fma_f16
__device__ __forceinline__ uint32_t rd_mf_center(uint32_t two) {
    const rd_h2 x = __builtin_bit_cast(rd_h2, two);
    const rd_h2 a = {(_Float16)20480.0f, (_Float16)20480.0f};                    // 5 * 2^12
    const rd_h2 c = {(_Float16)(-637.0f / 4096.0f), (_Float16)(-637.0f / 4096.0f)};
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_fma(x, a, c));
}

// the lane's 8 window bytes of one k-step -> B fragment (element order RD_MF_ELEM)
__device__ __forceinline__ rd_h8 rd_mf_frag(uint2 d) {
    rd_u4v v;
    // bytes 1 and 3 of a dword into the low bytes of the two halves: one v_perm_b32 (selector 0x0c = 0x00)
    v.x = rd_mf_center(d.x & 0x00FF00FFu);
    v.y = rd_mf_center(__builtin_amdgcn_perm(0u, d.x, 0x0c030c01u));
    v.z = rd_mf_center(d.y & 0x00FF00FFu);
    v.w = rd_mf_center(__builtin_amdgcn_perm(0u, d.y, 0x0c030c01u));
    return __builtin_bit_cast(rd_h8, v);
}

// -(ar cr + ai ci): numerator of py:89 for n = (ar, ai), n+ = (cr, ci) in the g frame
__device__ __forceinline__ float rd_mf_num(float ar, float ai, float cr, float ci) {
    return __builtin_fmaf(-ar, cr, -(ai * ci));
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
__device__ __forceinline__ uint32_t rd_lds_addr(const void *p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)p;
}

struct rd_mf_state {
    float F;          // max |component| seen
    uint32_t W;       // sign bits, first sample at the top (reversed at the end)
    float nmin[4];    // min |numerator| of each block's group
    float g0r, g0i;   // block 0's first output: its two boundary numerators come last
};

// One 16-output block of the tile: 6 MFMAs, the digit combine, then this lane's group of 8 signs.
// xw: LDS address this lane's (g6, g7) go to (+ 1024 B); xr: where its predecessors' are (+ 1024 B).
template <int B, int DBG>
__device__ __forceinline__ void rd_mf_block(const rd_h8 (&Ahi)[3], const rd_h8 (&Alo)[3], const rd_h8 (&bf)[9],
                                            uint32_t xw, uint32_t xr, rd_mf_state &st, float *dg, int dleft) {
    const rd_f16v zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    rd_f16v ah = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ahi[0], bf[2 * B], zero, 0, 0, 0);
    rd_f16v al = __builtin_amdgcn_mfma_f32_32x32x16_f16(Alo[0], bf[2 * B], zero, 0, 0, 0);
    ah = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ahi[1], bf[2 * B + 1], ah, 0, 0, 0);
    al = __builtin_amdgcn_mfma_f32_32x32x16_f16(Alo[1], bf[2 * B + 1], al, 0, 0, 0);
    ah = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ahi[2], bf[2 * B + 2], ah, 0, 0, 0);
    al = __builtin_amdgcn_mfma_f32_32x32x16_f16(Alo[2], bf[2 * B + 2], al, 0, 0, 0);
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
        rd_lds_write16<1024 * B>(xw, x);
    }
    rd_f4v p = {0.0f, 0.0f, 0.0f, 0.0f};
    if (B > 0) p = rd_lds_read16<1024 * B>(xr);  // g[base-1], g[base]: in flight under the group's own work
#pragma unroll
    for (int r = 0; r < 8; r++) st.F = rd_max3abs(st.F, g[2 * r], g[2 * r + 1]);
    float nm = 3.0e38f;
    uint32_t w6 = 0;
#pragma unroll
    for (int q = 2; q < 8; q += 2) {
        const float na = rd_mf_num(g[2 * q - 4], g[2 * q - 3], g[2 * q - 2], g[2 * q - 1]);
        const float nb = rd_mf_num(g[2 * q - 2], g[2 * q - 1], g[2 * q], g[2 * q + 1]);
        nm = rd_min3abs(nm, na, nb);
        w6 = rd_shift_in_sign(w6, na);
        w6 = rd_shift_in_sign(w6, nb);
    }
    if (B == 0) {
        st.g0r = g[0]; st.g0i = g[1];  // its two boundary numerators follow at the end of the tile
        st.W = w6;
    } else {
        rd_lds_wait(p);
        st.F = rd_max3abs(st.F, p.x, p.y);
        st.F = rd_max3abs(st.F, p.z, p.w);
        const float n0 = rd_mf_num(p.x, p.y, p.z, p.w);
        const float n1 = rd_mf_num(p.z, p.w, g[0], g[1]);
        nm = rd_min3abs(nm, n0, n1);
        uint32_t w2 = rd_shift_in_sign(0u, n0);
        w2 = rd_shift_in_sign(w2, n1);
        st.W = (st.W << 8) | (w2 << 6) | w6;
    }
    st.nmin[B] = nm;
}

__device__ __forceinline__ void rd_mf_issue(const rd_layout &lay, uint32_t s, uint32_t ti, uint8_t *img, int lane) {
    const uint8_t *src = lay.iq + (size_t)s * lay.stream_stride + (size_t)ti * RD_TILE_BYTES;
    const int perm = 8 * (lane & 7) + (lane >> 3);
    const __attribute__((address_space(1))) void *g0 = (const __attribute__((address_space(1))) void *)(src + perm * 16);
    // the instruction offset advances the global and the LDS address alike; the LDS base makes up
    // the difference between the 1024-byte source groups and the 1152-byte image groups
    __builtin_amdgcn_global_load_lds(g0, (__attribute__((address_space(3))) void *)(img + 16), 16, 0, 0);
    __builtin_amdgcn_global_load_lds(g0, (__attribute__((address_space(3))) void *)(img + 16 + 128), 16, 1024, 0);
    __builtin_amdgcn_global_load_lds(g0, (__attribute__((address_space(3))) void *)(img + 16 + 256), 16, 2048, 0);
    __builtin_amdgcn_global_load_lds(g0, (__attribute__((address_space(3))) void *)(img + 16 + 384), 16, 3072, 0);
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

// DBG: 0 product; 1 no global loads; 2 loads + LDS reads only; 3 also dumps g (dbg_g[tile][2048][2],
// sample order) - 1 and 2 are timing ablations with garbage results.
template <int DBG>
__global__ __launch_bounds__(RD_MF_WG, 2) void k_demod_mfma(rd_layout lay, uint32_t tiles_per_stream, uint32_t total_tiles,
                                                         uint32_t chunk, uint32_t *fix_list, uint32_t fix_cap,
                                                         uint32_t *counters, float *dbg_g) {
    // three separate arrays: the compiler then knows that the exchange buffer and the pending list do not
    // alias the image the LDS-DMA writes, and does not drain vmcnt (the NEXT tile's loads) before them
    __shared__ __attribute__((aligned(16))) uint8_t s_img[RD_MF_WAVES][RD_MF_IMG_PAD];
    __shared__ __attribute__((aligned(16))) uint8_t s_xb[RD_MF_WAVES][RD_MF_XB_BYTES];
    __shared__ uint32_t s_pend[RD_MF_WAVES][RD_MF_PEND];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint8_t *img = s_img[wave];
    uint8_t *xb = s_xb[wave] + 1024;
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
    // window addresses in the image
    const uint32_t own = 16 + RD_MF_GROUP_BYTES * (n >> 3) + 16 * (n & 7) + 8 * h;
    const uint32_t prv = n == 0 ? 8 * h
                       : (n & 7) ? own + 16 * 55
                                 : 16 + RD_MF_GROUP_BYTES * ((n >> 3) - 1) + 16 * 63 + 8 * h;
    // predecessor exchange: step b writes (g6, g7) of block b to xb[b][lane]; the lane that follows in
    // time is (n, 1, b) after (n, 0, b), (n, 0, b) after (n, 1, b-1), (n, 0, 0) after (n-1, 1, 3), and
    // (0, 0, 0) after lane 63 of the previous tile (carry slot, alternating with the tile parity).
    const int xrd = h ? 16 * (lane - 32) : 16 * (lane + 32) - 1024;  // + 1024 b, b >= 1
    uint32_t xrd0 = h ? 16 * (lane - 32) : lane ? 3072 + 16 * (lane + 31) : 4096 + 16;  // block 0 (end of tile)
    uint32_t xwr3 = lane == 63 ? 4096 : 3072 + 16 * lane;                                  // step 3
    const uint32_t rtoggle = lane == 0 ? 16u : 0u, wtoggle = lane == 63 ? 16u : 0u;
    if (lane < 8) ((uint32_t *)(xb + 4096))[lane] = 0;  // carry slots: finite values from the start
    const uint32_t xb_addr = rd_lds_addr(xb), pend_addr = rd_lds_addr(mypend);
    const uint32_t psel = h ? 0x07030602u : 0x05010400u;

    const uint32_t nwaves = gridDim.x * RD_MF_WAVES;
    // a wave's chunks: chunk index wg, wg + nwaves, ...; (s, ti) advances by one inside a chunk and by
    // (nwaves - 1) * chunk + 1 between chunks
    const uint32_t jump = (nwaves - 1) * chunk + 1;
    const uint32_t jq = jump / tiles_per_stream, jr = jump % tiles_per_stream;
    uint32_t tile = (blockIdx.x * RD_MF_WAVES + wave) * chunk;
    uint32_t s = tile / tiles_per_stream, ti = tile % tiles_per_stream;
    uint32_t inchunk = 0;

    uint32_t st_word = 0;
    uint32_t *st_ptr = nullptr;
    if (DBG != 1 && tile < total_tiles) rd_mf_issue(lay, s, ti, img, lane);
    while (tile < total_tiles) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this tile has landed in LDS
        uint2 D[9];
        D[0] = *(const uint2 *)(img + prv);
#pragma unroll
        for (int e = 0; e < 8; e++) D[e + 1] = *(const uint2 *)(img + own + 128 * e);
        // window in registers: the image can take the next tile while this one is computed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (st_ptr) *st_ptr = st_word;  // previous tile's words, before the loads (they share vmcnt)
        st_ptr = nullptr;
        uint32_t ntile, ns, nti, ninchunk;
        if (inchunk + 1 < chunk) {
            ntile = tile + 1; ninchunk = inchunk + 1;
            ns = s; nti = ti + 1;
            if (nti >= tiles_per_stream) { nti = 0; ns++; }
        } else {
            ntile = tile + jump; ninchunk = 0;
            ns = s + jq; nti = ti + jr;
            if (nti >= tiles_per_stream) { nti -= tiles_per_stream; ns++; }
        }
        if (DBG != 1 && ntile < total_tiles) rd_mf_issue(lay, ns, nti, img, lane);

        uint32_t word = 0, fbytes = 0;
        if (DBG == 2) {
#pragma unroll
            for (int j = 0; j < 9; j++) word ^= D[j].x ^ D[j].y;
        } else {
            rd_h8 bf[9];
#pragma unroll
            for (int j = 0; j < 9; j++) bf[j] = rd_mf_frag(D[j]);
            rd_mf_state stt;
            stt.F = 0.0f; stt.W = 0; stt.g0r = 0.0f; stt.g0i = 0.0f;
            const uint32_t xw = xb_addr + 16 * lane, xr = xb_addr + xrd;
            float *dg = DBG == 3 ? dbg_g + ((size_t)tile * RD_TILE_SAMPLES + 64 * n + 8 * h + 1) * 2 : nullptr;
            const int dleft = RD_TILE_SAMPLES - (64 * n + 8 * h + 1);  // outputs of this lane inside the tile
            rd_mf_block<0, DBG>(Ahi, Alo, bf, xw, xr, stt, dg, dleft);
            rd_mf_block<1, DBG>(Ahi, Alo, bf, xw, xr, stt, dg, dleft);
            rd_mf_block<2, DBG>(Ahi, Alo, bf, xw, xr, stt, dg, dleft);
            rd_mf_block<3, DBG>(Ahi, Alo, bf, xb_addr + xwr3 - 3072, xr, stt, dg, dleft);
            {   // block 0's first two numerators: W holds 30 bits, its bits 31, 30 are theirs
                rd_f4v p = rd_lds_read16<0>(xb_addr + xrd0);
                rd_lds_wait(p);
                stt.F = rd_max3abs(stt.F, p.x, p.y);
                stt.F = rd_max3abs(stt.F, p.z, p.w);
                const float n0 = rd_mf_num(p.x, p.y, p.z, p.w);
                const float n1 = rd_mf_num(p.z, p.w, stt.g0r, stt.g0i);
                stt.nmin[0] = rd_min3abs(stt.nmin[0], n0, n1);
                stt.W |= __builtin_bit_cast(uint32_t, n0) & 0x80000000u;
                stt.W |= (__builtin_bit_cast(uint32_t, n1) >> 1) & 0x40000000u;
            }
            xrd0 ^= rtoggle;
            xwr3 ^= wtoggle;
            word = __builtin_bitreverse32(stt.W);  // byte b = the signs of block b's group
            const float thr = rd_mf_threshold(stt.F);
#pragma unroll
            for (int b = 0; b < 4; b++) fbytes |= (stt.nmin[b] > thr) ? 0u : (1u << (8 * b));  // NaN -> flagged
        }
        // The lane holds bytes (groups) 2b + h of its column's two words: gather word h of the column
        // (lanes n and n + 32 exchange halves), the flags likewise.
        uint32_t gmask;
        {
            const auto w2 = __builtin_amdgcn_permlane32_swap(word, word, false, false);  // [0]: half 0's, [1]: half 1's
            word = __builtin_amdgcn_perm(w2[1], w2[0], psel);
            const auto f2 = __builtin_amdgcn_permlane32_swap(fbytes, fbytes, false, false);
            const uint32_t fb = __builtin_amdgcn_perm(f2[1], f2[0], psel);
            gmask = ((fb * 0x00204081u) >> 21) & 0xFu;  // bytes 0/1 -> bits
        }
        const bool carry = inchunk > 0 && ti > 0;  // previous iteration = previous tile of this stream
        const uint32_t run = ti * 64 + 2 * n + h;  // word index in the stream
        const uint32_t t0 = run * RD_RUN;
        if (lane == 0) {
            if (ti == 0 && !lay.hist_mode) gmask = 0xFu;  // zero history: first run exact
            else if (!carry) gmask |= 1u;                 // no predecessors for the tile's first group
        }
        const bool ragged = (ti + 1 == tiles_per_stream) && (lay.n_samples % RD_TILE_SAMPLES) != 0;
        if (!ragged) {
            st_word = word;
            st_ptr = &lay.bits[(size_t)s * lay.bits_stride + run];
        } else if (t0 < lay.n_samples) {
            const uint32_t left = lay.n_samples - t0;
            if (left < RD_RUN) {
                word &= (1u << left) - 1u;
                gmask &= (1u << ((left + RD_GROUP - 1) / RD_GROUP)) - 1u;
            }
            st_word = word;
            st_ptr = &lay.bits[(size_t)s * lay.bits_stride + run];
        } else {
            gmask = 0;
        }
        if (DBG == 1 || DBG == 2) gmask = 0;
        const uint64_t fm = __ballot(gmask != 0);
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
        tile = ntile; s = ns; ti = nti; inchunk = ninchunk;
    }
    if (st_ptr) *st_ptr = st_word;
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
    static int dbg = -1, chunk_env = 0, per_cu_env = 0, n_cu = 0, per_cu_occ[4] = {0, 0, 0, 0};
    if (dbg < 0) {
        dbg = rd_mf_env("RD_K1_DEBUG", 0);
        chunk_env = rd_mf_env("RD_K1_CHUNK", 0);
        per_cu_env = rd_mf_env("RD_K1_WGS_PER_CU", 0);
        int dev = 0;
        hipGetDevice(&dev);
        hipDeviceProp_t prop;
        n_cu = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                   ? prop.multiProcessorCount : 256;
    }
    const int variant = dbg_g ? 3 : (dbg == 1 || dbg == 2) ? dbg : 0;
    // persistent grid sized from the occupancy API (registers and LDS of the variant actually launched)
    if (!per_cu_occ[variant]) {
        int occ = 0;
        hipError_t e = hipErrorUnknown;
        switch (variant) {
            case 0: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_demod_mfma<0>, RD_MF_WG, 0); break;
            case 1: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_demod_mfma<1>, RD_MF_WG, 0); break;
            case 2: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_demod_mfma<2>, RD_MF_WG, 0); break;
            default: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_demod_mfma<3>, RD_MF_WG, 0); break;
        }
        per_cu_occ[variant] = (e == hipSuccess && occ >= 1) ? (occ > 8 ? 8 : occ) : 2;
    }
    const int per_cu = (per_cu_env >= 1 && per_cu_env <= 8) ? per_cu_env : per_cu_occ[variant];
    uint32_t chunk = chunk_env > 0 ? (uint32_t)chunk_env : 6;
    if (chunk > total) chunk = total;
    const uint64_t chunks = (total64 + chunk - 1) / chunk;
    uint64_t wgs = (chunks + RD_MF_WAVES - 1) / RD_MF_WAVES;
    const uint64_t max_wgs = (uint64_t)n_cu * per_cu;
    if (wgs > max_wgs) wgs = max_wgs;
#define RD_LAUNCH_MF(D)                                                                                              \
    do {                                                                                                             \
        if (ev_start || ev_stop)                                                                                     \
            hipExtLaunchKernelGGL((k_demod_mfma<D>), dim3((unsigned)wgs), dim3(RD_MF_WG), 0, st, ev_start, ev_stop,  \
                                  0, lay, tps, total, chunk, fix_list, fix_cap, counters, dbg_g);                    \
        else                                                                                                         \
            hipLaunchKernelGGL((k_demod_mfma<D>), dim3((unsigned)wgs), dim3(RD_MF_WG), 0, st, lay, tps, total,       \
                               chunk, fix_list, fix_cap, counters, dbg_g);                                           \
    } while (0)
    if (variant == 3) RD_LAUNCH_MF(3);
    else if (variant == 1) RD_LAUNCH_MF(1);
    else if (variant == 2) RD_LAUNCH_MF(2);
    else RD_LAUNCH_MF(0);
#undef RD_LAUNCH_MF
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
    RD_DBG_CHK(hipMalloc(&d_cnt, RD_CNT_SLOTS * 4));
    RD_DBG_CHK(hipMemset(d_cnt, 0, RD_CNT_SLOTS * 4));
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
