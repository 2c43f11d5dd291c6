// rd_demod_mfma.hip - k_demod_mfma: the fused IQ -> sign-bit kernel with the FIR on the matrix pipe.
//
// Reads lay.iq once, writes the packed sign bits and lists the 8-sample groups whose sign is not certain (the tail
// re-evaluates those exactly: k_tail / k_fixup in rd_kernels.hip).  Reference stages: LUT py:38-39 + rotate_fs4
// py:46-49 + fir9 py:71-73 + discriminate numerator py:89 + quantize py:98 (py = /root/reference/src/rtldavis/dsp.py).
// Arithmetic and its error bound: rd_mfma.h.
//
// One wave = one 2048-sample tile per iteration.  Column n = lane & 31 owns samples a0 .. a0+63 (a0 = tile + 64 n), its
// window is the 144 bytes from 16 bytes before them; lane half h = lane >> 5 supplies bytes 8h..8h+7 of every 16-byte
// k-step of that window.  The 8-output formulation of rd_mfma.h: per block b = 0..7 a lane receives re / im (both tap
// digits) of g[a0 + 8 b + 4 h + 1 + r'], r' = 0..3, and decides the four signs of t = base .. base + 3,
// base = a0 + 8 b + 4 h: two numerators from its own outputs, two with g[base - 1], g[base] of the lane that precedes
// it in time, exchanged through a per-wave LDS buffer.  A wave works through chunks of consecutive tiles so that a
// tile's first group finds its predecessors in the previous iteration; at the start of a chunk (and of a stream)
// that one group is listed for the exact pass instead.
// The forms this file used to carry beside this one - the 16-output formulation (24 MFMAs per tile), two image buffers,
// software-pipelined blocks, the in-kernel preamble search and the self-fix variant - lost their A/B runs
// (docs/history/r03.md, profiles/r03_*) and live in the git history (b326bb2 and before).
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <hip/hip_ext.h>

#include "rd_internal.h"
#include "rd_math.h"
#include "rd_mfma.h"

#define RD_MF_WG 256
#define RD_MF_WAVES (RD_MF_WG / 64)
#ifndef RD_MF_MINWAVES
#define RD_MF_MINWAVES 4  // waves per SIMD the register allocator must leave room for (128 VGPRs)
#endif
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
#define RD_MF_PEND 32        // entries a wave keeps before it appends them to the list (or to the groups' buckets)
// tiles whose words are stored together (a multiple of 4 that divides the default chunk)
#ifndef RD_MF_STAGE_TILES
#define RD_MF_STAGE_TILES 4
#endif

typedef _Float16 rd_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 rd_h16 __attribute__((ext_vector_type(16)));
typedef _Float16 rd_h2 __attribute__((ext_vector_type(2)));
typedef float rd_f16v __attribute__((ext_vector_type(16)));
typedef uint32_t rd_u4v __attribute__((ext_vector_type(4)));
typedef float rd_f4v __attribute__((ext_vector_type(4)));
typedef uint32_t rd_u2v __attribute__((ext_vector_type(2)));

static const rd_mf_taps h_mf_taps = rd_mf_make_taps();   // (the 16-output tap matrix: the RSSI windows of rd_kernels.hip use it)
__device__ const rd_mf_taps8 g_mf_taps8 = rd_mf_make_taps8();
static const rd_mf_taps8 h_mf_taps8 = rd_mf_make_taps8();
extern "C" void rd_debug_mfma_taps8(uint16_t *out) { memcpy(out, &h_mf_taps8, sizeof h_mf_taps8); }
// the same tap matrix compressed for the 2:4-sparse instruction (rd_mfma.h): what the kernel issues; -DRD_MF_SPARSE=0
// (`make dense`) builds the pair of dense instructions of round 3 instead - same results, 1.5 % slower at the power cap
#ifndef RD_MF_SPARSE
#define RD_MF_SPARSE 1
#endif
constexpr bool rd_mf_taps8s_ok() { bool ok = true; (void)rd_mf_make_taps8s(&ok); return ok; }
static_assert(rd_mf_taps8s_ok(), "a row of the tap matrix has more than two non-zero elements in a group of four");
constexpr rd_mf_taps8s rd_mf_taps8s_value() { bool ok = true; return rd_mf_make_taps8s(&ok); }
__device__ const rd_mf_taps8s g_mf_taps8s = rd_mf_taps8s_value();
static const rd_mf_taps8s h_mf_taps8s = rd_mf_taps8s_value();
extern "C" void rd_debug_mfma_taps8s(uint16_t *vals, uint32_t *idx) {
    memcpy(vals, h_mf_taps8s.v, sizeof h_mf_taps8s.v);
    memcpy(idx, h_mf_taps8s.idx, sizeof h_mf_taps8s.idx);
}

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

// Guard band in two steps (bound: rd_mfma.h).  Per sample r = |num| - 2^-21 |b d| takes care of the part of
// the error that scales with the products; what is left, 4 E0 F + const, needs F = the largest |component|
// involved.  The common path compares a group's min r with that term at F = the largest |g| ANY input can
// produce (RD_MF_C0_MAX, 2.6e-4 byte units squared: ~1e-5 of the samples of a noise input are below it);
// only when some lane of the wave fails that test - a wave-uniform branch - is F taken from the values at
// hand (v_max3 over the block) and the test repeated with it.
__device__ __forceinline__ bool rd_mf_any(bool c) { return __ballot(c) != 0; }


// ------------------------------------------------------------------------------------------------------------------
// The 8-output formulation of rd_mfma.h: 16 MFMAs per tile (a matrix instruction in flight slows the vector issue of the
// whole SIMD).  Lane (n, h), block b = 0..7: re / im of g[a0 + 8 b + 4 h + 1 + r'], r' = 0..3 - it decides the four signs of
// t = base .. base + 3 (base = a0 + 8 b + 4 h): two from its own outputs, two with g[base - 1], g[base] of the lane
// that precedes it in time ((n, 0, b) for h = 1, (n, 1, b - 1) for h = 0, (n - 1, 1, 7) for block 0 of half 0, the
// previous tile's lane 63 for lane 0).  Those two are FINISHED one block later, under the next block's MFMAs: the
// exchange's LDS round trip then costs nothing.  Block 0's are finished at the end of the tile.
// Exchange buffers (per wave): Q (block 0) at 0, P1 (odd blocks) at 1024, P0 (even blocks >= 2) at 2048, the two
// carry slots at 3072.
struct rd_mf8_addr {
    uint32_t xw;             // this lane's slot: + 0 / 1024 / 2048 for block 0 / odd / even
    uint32_t xw7;            // block 7: the slot in P1, lane 63: the carry slot (toggles with the tile parity)
    uint32_t rd1, rdE, rdO;  // predecessors of block 1 / the even blocks / the odd blocks >= 3
    uint32_t rd0;            // block 0's, read at the end of the tile (lane 0: the carry slot, toggles)
};
struct rd_mf8_kept {  // what a block leaves for its deferred finish
    float n2, n3;     // its two own numerators
    float g0r, g0i;   // its first output (second boundary numerator)
    float g1r, g1i, g2r, g2i;  // for the second-level guard (F over the group)
    float nmin, tmax;
};

template <int B>
__device__ __forceinline__ constexpr int rd_mf8_woff() { return B == 0 ? 0 : (B & 1) ? 1024 : 2048; }

// finish block B >= 1: boundary numerators, guard band, the four signs into W (8 bits per block: four signs, then
// four bits of no meaning that the last shift drags in - masked at the end of the tile)
template <int B, int DBG>
__device__ __forceinline__ void rd_mf8_finish(const rd_mf8_kept &k, rd_f4v &p, uint32_t &W, uint32_t &fb, bool &slow) {
    rd_lds_wait(p);
    float t0, t1;
    const float n0 = rd_mf_num(p.x, p.y, p.z, p.w, t0);
    const float n1 = rd_mf_num(p.z, p.w, k.g0r, k.g0i, t1);
    if (DBG == 0 || DBG == 3) {
        const float nm = rd_mf_guard(rd_min3abs(k.nmin, n0, n1), rd_max3abs(k.tmax, t0, t1));
        if (rd_mf_any(!(nm > RD_MF_C0_MAX))) {  // rare; NaN counts as inside
            slow = true;
            float F = rd_max3abs(0.0f, p.x, p.y);
            F = rd_max3abs(F, p.z, p.w);
            F = rd_max3abs(F, k.g0r, k.g0i);
            F = rd_max3abs(F, k.g1r, k.g1i);
            F = rd_max3abs(F, k.g2r, k.g2i);
            if (!(nm > rd_mf_c0(F))) fb |= 1u << (8 * (B & 3));
        }
    }
    uint32_t w = rd_shift_in_sign(W, n0);
    w = rd_shift_in_sign(w, n1);
    w = rd_shift_in_sign(w, k.n2);
    W = __builtin_amdgcn_alignbit(w, __builtin_bit_cast(uint32_t, k.n3), 27);  // (w << 5) | sign and four more bits
}

template <int B, int DBG>
__device__ __forceinline__ void rd_mf8_block(const rd_h8 (&A)[2], const rd_f16v &dcC, const int sidx, const int sidx2, const rd_u2v (&D)[9],
                                             rd_h8 &bfA, rd_h16 &bb, const rd_mf8_addr &ad, rd_mf8_kept &k0, rd_mf8_kept &kp, rd_f4v &p, uint32_t &Wlo,
                                             uint32_t &Whi, uint32_t &fblo, uint32_t &fbhi, bool &slow, float *dg, int dleft) {
    const rd_h8 bfB = rd_mf_frag(D[B + 1]);
    if (RD_MF_SPARSE && DBG != 8 && DBG != 9) {
        // the 16-element B operand lives in eight registers for the whole tile; the new fragment goes into the half the
        // fragment of two blocks ago occupies - even blocks read (F_b, F_b+1), odd blocks (F_b+1, F_b) with the taps of
        // the two k-steps swapped (A[1], sidx2): no register-to-register move
        const rd_h16 wB = __builtin_shufflevector(bfB, bfB, 0, 1, 2, 3, 4, 5, 6, 7, 0, 1, 2, 3, 4, 5, 6, 7);
        if (B & 1) bb = __builtin_shufflevector(wB, bb, 0, 1, 2, 3, 4, 5, 6, 7, 24, 25, 26, 27, 28, 29, 30, 31);
        else bb = __builtin_shufflevector(bb, wB, 0, 1, 2, 3, 4, 5, 6, 7, 16, 17, 18, 19, 20, 21, 22, 23);
    }
    __builtin_amdgcn_sched_barrier(0);
    rd_f16v acc;
    if (DBG == 8 || DBG == 9) {
        // WRONG results, timing only: ONE matrix instruction per block - what a 2:4-sparse v_smfmac_f32_32x32x32_f16 would
        // issue (the tap rows are 2:4 sparse).  9: with the copy of the DC term into the accumulators that instruction
        // needs (it has no C operand)
        rd_f16v c0 = dcC;
        if (DBG == 9) asm volatile("" : "+v"(c0));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0], bfA, c0, 0, 0, 0);
        asm volatile("" : : "v"(bfB));
    } else if (RD_MF_SPARSE) {
        // ONE 2:4-sparse instruction for the block's two k-steps (A[0] = the compressed taps, sidx their positions).  It
        // accumulates in place: the accumulators start as a copy of the constant C tuple (-D_hi in the hi rows)
        // (-D_hi in the hi rows, 0 in the lo rows, written per block from a scalar pair - eight 64-bit moves: a constant
        // 16-register tuple kept for the whole kernel, what the dense pair's C operand is, does not fit beside the rest)
        const uint32_t dbits = __builtin_bit_cast(uint32_t, -(float)RD_MF_DHI / 16777216.0f);
        uint64_t dh2 = ((uint64_t)dbits << 32) | dbits;
        asm volatile("" : "+s"(dh2));
        // (sixteen v_mov_b32 and eight v_pk_mov_b32 measured 0.3 and 1.7 % slower: gpurun r4 ab i32 / i2)
        struct { uint64_t q[8]; } ci;
#pragma unroll
        for (int i = 0; i < 4; i++) asm volatile("v_mov_b64 %0, %1" : "=v"(ci.q[i]) : "s"(dh2));
#pragma unroll
        for (int i = 4; i < 8; i++) asm volatile("v_mov_b64 %0, 0" : "=v"(ci.q[i]));
        // (a vector write of a register the matrix instruction reads as its accumulator needs two wait states in between;
        // the compiler's hazard recognizer does not look into the asm statements above)
        asm volatile("s_nop 1" ::: "memory");
        const rd_f16v c0 = __builtin_bit_cast(rd_f16v, ci);
        acc = (B & 1) ? __builtin_amdgcn_smfmac_f32_32x32x32_f16(A[1], bb, c0, sidx2, 0, 0)
                      : __builtin_amdgcn_smfmac_f32_32x32x32_f16(A[0], bb, c0, sidx, 0, 0);
    } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0], bfA, dcC, 0, 0, 0);  // C: -D_hi in the hi rows
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[1], bfB, acc, 0, 0, 0);
    }
    // the previous block's finish runs under these two (block 0's waits for the end of the tile)
    if (B >= 2) rd_mf8_finish<B - 1, DBG>(kp, p, B - 1 <= 3 ? Wlo : Whi, B - 1 <= 3 ? fblo : fbhi, slow);
    __builtin_amdgcn_sched_barrier(0);
    float g[8];  // g[2 r'], g[2 r' + 1] = re, im of output r'
#pragma unroll
    for (int i = 0; i < 8; i++) g[i] = __builtin_fmaf(acc[i], 2048.0f, acc[8 + i]);
    if (DBG == 3) {
#pragma unroll
        for (int r = 0; r < 4; r++)  // the tile's last output (column 31, half 1, block 7, r' = 3) belongs to the next tile
            if (8 * B + r < dleft) { dg[2 * (8 * B + r)] = g[2 * r]; dg[2 * (8 * B + r) + 1] = g[2 * r + 1]; }
    }
    if (DBG != 10 && DBG != 11) {   // (10, 11 - WRONG results, timing only: without the predecessor exchange's LDS traffic)
        const rd_f4v x = {g[4], g[5], g[6], g[7]};
        if (B == 7) rd_lds_write16<0>(ad.xw7, x);
        else rd_lds_write16<rd_mf8_woff<B>()>(ad.xw, x);
    } else {
        asm volatile("" : : "v"(g[4]), "v"(g[5]), "v"(g[6]), "v"(g[7]));
    }
    if (B >= 1 && DBG != 10 && DBG != 11) p = rd_lds_read16<0>(B == 1 ? ad.rd1 : (B & 1) ? ad.rdO : ad.rdE);  // consumed one block later
    if (DBG == 11) asm volatile("" : "+v"(p));   // (11: the predecessor's values unknown to the compiler - the finish's arithmetic stays, 10 folds it away)
    rd_mf8_kept &k = B == 0 ? k0 : kp;
    float t2, t3;
    k.n2 = rd_mf_num(g[0], g[1], g[2], g[3], t2);
    k.n3 = rd_mf_num(g[2], g[3], g[4], g[5], t3);
    k.nmin = rd_min3abs(3.0e38f, k.n2, k.n3);
    k.tmax = rd_max3abs(0.0f, t2, t3);
    k.g0r = g[0]; k.g0i = g[1];
    k.g1r = g[2]; k.g1i = g[3]; k.g2r = g[4]; k.g2i = g[5];
    bfA = bfB;
}

// One tile: `word` = word h of column n (final), `fbw` = its flag bytes (byte k != 0: group k inside the guard band).
template <int DBG>
__device__ __forceinline__ void rd_mf8_tile(const rd_h8 (&A)[2], const rd_f16v &dcC, const int sidx, const int sidx2, const rd_u2v (&D)[9],
                                            const rd_mf8_addr &ad,
                                            uint32_t &word, uint32_t &fbw, bool &slow, float *dg, int dleft) {
    rd_h8 bfA = rd_mf_frag(D[0]);
    rd_h16 bb = __builtin_shufflevector(bfA, bfA, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);   // (the low half counts)
    rd_mf8_kept k0, kp;
    rd_f4v p = {0.0f, 0.0f, 0.0f, 0.0f};
    uint32_t Wlo = 0, Whi = 0, fblo = 0, fbhi = 0;
    slow = false;
    rd_mf8_block<0, DBG>(A, dcC, sidx, sidx2, D, bfA, bb, ad, k0, kp, p, Wlo, Whi, fblo, fbhi, slow, dg, dleft);
    rd_mf8_block<1, DBG>(A, dcC, sidx, sidx2, D, bfA, bb, ad, k0, kp, p, Wlo, Whi, fblo, fbhi, slow, dg, dleft);
    rd_mf8_block<2, DBG>(A, dcC, sidx, sidx2, D, bfA, bb, ad, k0, kp, p, Wlo, Whi, fblo, fbhi, slow, dg, dleft);
    rd_mf8_block<3, DBG>(A, dcC, sidx, sidx2, D, bfA, bb, ad, k0, kp, p, Wlo, Whi, fblo, fbhi, slow, dg, dleft);
    rd_mf8_block<4, DBG>(A, dcC, sidx, sidx2, D, bfA, bb, ad, k0, kp, p, Wlo, Whi, fblo, fbhi, slow, dg, dleft);
    rd_mf8_block<5, DBG>(A, dcC, sidx, sidx2, D, bfA, bb, ad, k0, kp, p, Wlo, Whi, fblo, fbhi, slow, dg, dleft);
    rd_mf8_block<6, DBG>(A, dcC, sidx, sidx2, D, bfA, bb, ad, k0, kp, p, Wlo, Whi, fblo, fbhi, slow, dg, dleft);
    rd_mf8_block<7, DBG>(A, dcC, sidx, sidx2, D, bfA, bb, ad, k0, kp, p, Wlo, Whi, fblo, fbhi, slow, dg, dleft);
    // block 0's predecessors (the previous column's block 7, this column's half 0, or the carry) together with block 7's
    rd_f4v p0 = rd_lds_read16<0>(ad.rd0);
    rd_mf8_finish<7, DBG>(kp, p, Whi, fbhi, slow);  // (its wait covers both reads)
    {
        rd_lds_wait(p0);
        float t0, t1;
        const float n0 = rd_mf_num(p0.x, p0.y, p0.z, p0.w, t0);
        const float n1 = rd_mf_num(p0.z, p0.w, k0.g0r, k0.g0i, t1);
        if (DBG == 0 || DBG == 3) {
            // (block 0 keeps its outputs until here for the second-level test: listing its groups on the constant
            // threshold alone made the fix-up list 76 % longer and k_fixup 8 us slower)
            const float nm = rd_mf_guard(rd_min3abs(k0.nmin, n0, n1), rd_max3abs(k0.tmax, t0, t1));
            if (rd_mf_any(!(nm > RD_MF_C0_MAX))) {
                slow = true;
                float F = rd_max3abs(0.0f, p0.x, p0.y);
                F = rd_max3abs(F, p0.z, p0.w);
                F = rd_max3abs(F, k0.g0r, k0.g0i);
                F = rd_max3abs(F, k0.g1r, k0.g1i);
                F = rd_max3abs(F, k0.g2r, k0.g2i);
                if (!(nm > rd_mf_c0(F))) fblo |= 1u;
            }
        }
        uint32_t b0 = rd_shift_in_sign(0u, n0);
        b0 = rd_shift_in_sign(b0, n1);
        b0 = rd_shift_in_sign(b0, k0.n2);
        b0 = __builtin_amdgcn_alignbit(b0, __builtin_bit_cast(uint32_t, k0.n3), 27);
        Wlo = (b0 << 24) | Wlo;  // blocks 1-3 left 24 bits there
    }
    // a byte of W = [s0 s1 s2 s3 x x x x]: after the bit reversal block b's four signs are the low nibble of byte b & 3,
    // sample order.  Word h of the column = half 0's nibbles | half 1's << 4, of blocks 0-3 (h = 0) or 4-7 (h = 1):
    // the swap hands lane (n, 0) the other half's Vlo and lane (n, 1) the other half's Vhi.
    const uint32_t Vlo = __builtin_bitreverse32(Wlo & 0xF0F0F0F0u), Vhi = __builtin_bitreverse32(Whi & 0xF0F0F0F0u);
    const auto r = __builtin_amdgcn_permlane32_swap(Vlo, Vhi, false, false);
    word = (r[1] << 4) | r[0];
    fbw = 0;
    if (slow && rd_mf_any((fblo | fbhi) != 0)) {
        const auto f = __builtin_amdgcn_permlane32_swap(fblo, fbhi, false, false);
        fbw = f[0] | f[1];
    }
}

// cache policy of the tile loads (the builtin's aux operand): 0 default, 2 = nt.  The input is streamed
// once and never re-read: nt loads-only 0.330 ms (6.7 TB/s) against 0.358, loads + stores 0.427 against 0.462,
// whole kernel 0.496 against 0.509, and the search kernel behind it finds more of the bits in cache.
#ifndef RD_MF_LOAD_AUX
#define RD_MF_LOAD_AUX 2
#endif

// halo_dma = false: the caller has put the 16 bytes before the tile into the image itself (RD_OPT_HALO)
__device__ __forceinline__ void rd_mf_issue(const rd_layout &lay, uint32_t s, uint32_t ti, uint8_t *img, int lane,
                                            bool halo_dma = true) {
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
    if (lane == 0 && halo_dma)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (has_halo ? -16 : 0)),
                                         (__attribute__((address_space(3))) void *)(img), 16, 0, 0);
}

// the lane number from the exec-mask counters: inside the rarely taken paths below, so that no register holds a
// lane-derived address for them across the whole tile loop (the allocator spilled those - and a reload is a scratch
// load with a vmcnt(0) wait in the stretch between a tile's arrival and the next tile's loads)
__device__ __forceinline__ int rd_lane_now() {
    return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// bucket_cnt != null (RD_DEMOD_FIX_BUCKETS, the one-launch tail): entry e of stream s goes to the bucket of the workgroup
// of k_tail that owns s - fix_list[(s >> RD_FT_GSH) * fix_cap + slot], slot drawn from bucket_cnt[s >> RD_FT_GSH] - so
// that each of those workgroups finds the words of its own streams without reading the whole list.  A wave flushes
// ~11 entries once, at its end: 46 k returning atomics per launch spread over a thousand counters.
__device__ __forceinline__ void rd_mf_flush(const uint32_t *pend, uint32_t count, uint32_t *fix_list, uint32_t fix_cap,
                                            uint32_t *counters, uint32_t bits_stride, uint32_t *bucket_cnt) {
    const int lane = rd_lane_now();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (bucket_cnt) {
        for (uint32_t i = lane; i < count; i += 64) {
            const uint32_t e = pend[i];
            const uint32_t grp = ((e >> 4) / bits_stride) >> RD_FT_GSH;
            const uint32_t slot = atomicAdd(&bucket_cnt[grp], 1u);  // (past fix_cap: k_tail sees the count and raises the overflow flag)
            if (slot < fix_cap) fix_list[(size_t)grp * fix_cap + slot] = e;
        }
        return;
    }
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&counters[RD_CNT_FIX], count);
    base = __builtin_amdgcn_readfirstlane(base);
    for (uint32_t i = lane; i < count; i += 64)
        if (base + i < fix_cap) fix_list[base + i] = pend[i];
}

// Store the staged words: four tiles as one 16-byte store per lane, fewer tile by tile.
// Self-fix (RD_DEMOD_SELF_FIX, the batch path): a wave that has run out of tiles re-evaluates the groups it flagged
// itself - the arithmetic of k_fixup (rd_exact_group_dw: float64 on integer data, exact), one lane per listed word -
// and patches its own words.  A wave lists ~11 words in its life (the first group of each of its chunks, the first run
// of a stream, the occasional group inside the guard band): one pass, ~1 % of the wave's instructions, instead of a
// kernel launch whose 18 us are a launch and two dependent memory latencies.  The wave's word stores are waited for
// first; the entries name words of the wave's own tiles only.  `noinline`: the float64 path's registers are needed

// Store the staged words: four tiles as one 16-byte store per lane, fewer tile by tile.
__device__ __forceinline__ void rd_mf_store_staged(uint32_t stage_addr, uint32_t nst, uint32_t *base, uint32_t stflags) {
    const int lane = rd_lane_now();
    if (nst == RD_MF_STAGE_TILES) {
#pragma unroll
        for (int j = 0; j < RD_MF_STAGE_TILES / 4; j++) {
            const rd_u4v v = rd_lds_read16u(stage_addr + 16 * (lane + 64 * j));
            // non-temporal: the words are read next by another kernel, never again by this one (0.3-1.3 % faster;
            // RD_K1_STFLAGS & 1 of the diagnostic library switches to plain stores for A/B runs)
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
    // 32-bit scalar arithmetic (a 64-bit compare of two scalars compiles to a VECTOR compare): a chunk id at or
    // past the number of chunks means "none"; below it, chunk_id * chunk < total + chunk fits (total < 2^32 - 2^16)
    const uint32_t n_chunks = (total + chunk - 1) / chunk;
    const bool have = chunk_id < n_chunks;
    const uint32_t nt = have ? (uint32_t)chunk_id * chunk : 0u;
    c.tile = have ? nt : 0xFFFFFFFFu;
    c.s = 0; c.ti = 0;
    if (have) {  // (the one division per chunk: done at the loop top, where few registers are live)
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

// In-kernel stamps (RD_OPT_STAMP, diagnostic library only; cdna_hip_programming.md section 7): one statement with
// its own lgkmcnt(0), fenced against the scheduler on both sides.
__device__ __forceinline__ uint64_t rd_stamp() {
    uint64_t t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
__device__ __forceinline__ uint64_t rd_stamp_real() {  // 100 MHz constant clock
    uint64_t t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}


// DBG: 0 product; 3 also dumps g (dbg_g[tile][2048][2], sample order; the test hook).  Diagnostic library only
// (-DRD_DIAG, librtldavis_hip_diag.so; garbage results): 1 no global loads; 2 loads + LDS reads + stores only; 6 loads
// only; 7 no guard band.  STAMP (diagnostic library): s_memtime stamps, per-wave sums in dbg_g.
// stflags: 1 plain instead of non-temporal word stores, 2 all stores into one MiB (wrong results), 4 no s_setprio,
// 4096 equal shares of chunks per wave instead of the work queues (1 - 4096: diagnostic library, RD_K1_STFLAGS);
// RD_STF_BUCKETS: the fix-up entries go to per-group buckets (RD_DEMOD_FIX_BUCKETS, set by the host).
#define RD_STAMP_WORDS 12
#define RD_STF_BUCKETS 8192u
template <int DBG, bool STAMP>
__global__ __launch_bounds__(RD_MF_WG, RD_MF_MINWAVES) void k_demod_mfma(rd_layout lay, uint32_t tiles_per_stream, uint32_t total_tiles,
                                                         uint32_t chunk, uint32_t *fix_list, uint32_t fix_cap,
                                                         uint32_t *counters, float *dbg_g, uint32_t stflags) {
    // one image buffer per wave: tile i+1 in flight while tile i is computed, 4 workgroups per CU
    __shared__ __attribute__((aligned(16))) uint8_t s_img[RD_MF_WAVES][RD_MF_IMG_PAD];
    __shared__ __attribute__((aligned(32))) uint8_t s_xb[RD_MF_WAVES][RD_MF_XB_BYTES];
    __shared__ uint32_t s_pend[RD_MF_WAVES][RD_MF_PEND];
    // packed words of up to four consecutive tiles of a stream, stored together: one 16-byte store per lane
    // (1 KiB contiguous per wave) instead of four dword stores (round 1: a dword store per tile cost 20 % of
    // the read bandwidth, profiles/r01_ubench_read_bw.txt)
    __shared__ __attribute__((aligned(16))) uint32_t s_stage[RD_MF_WAVES][RD_MF_STAGE_TILES][64];
    constexpr bool LOADS = DBG != 1;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint8_t *img0 = s_img[wave];
    uint8_t *xb = s_xb[wave];
    uint32_t *mypend = s_pend[wave];
    uint32_t npend = 0;
    // RD_STF_BUCKETS: the fix-up entries go to per-group buckets whose counters the last pointer argument names
    constexpr bool CAN_BUCKET = DBG == 0 && !STAMP;
    uint32_t *bucket_cnt = (CAN_BUCKET && (stflags & RD_STF_BUCKETS)) ? (uint32_t *)dbg_g : nullptr;

    const int n = lane & 31, h = lane >> 5;
    // tap fragments (two, 8 registers) and the C operand of a block's first MFMA (-D_hi in the hi-digit rows, 0 in the
    // lo-digit rows: 16 registers) for the whole kernel
    rd_h8 A8[2];
    rd_f16v dcC;
    int sidx = 0, sidx2 = 0;
    if (RD_MF_SPARSE && DBG != 8 && DBG != 9) {
        // (the compressed taps of this lane's k-step, and - for the odd blocks, whose B operand holds the two k-steps the
        // other way round - of the other one: the other lane half's)
        A8[0] = *(const rd_h8 *)g_mf_taps8s.v[lane];
        A8[1] = *(const rd_h8 *)g_mf_taps8s.v[lane ^ 32];
        sidx = (int)g_mf_taps8s.idx[lane];
        sidx2 = (int)g_mf_taps8s.idx[lane ^ 32];
    } else {
        A8[0] = *(const rd_h8 *)g_mf_taps8.v[0][lane];
        A8[1] = *(const rd_h8 *)g_mf_taps8.v[1][lane];
    }
    const float dchi = -(float)RD_MF_DHI / 16777216.0f;
#pragma unroll
    for (int i = 0; i < 16; i++) dcC[i] = i < 8 ? dchi : 0.0f;
    // opaque: sixteen registers for the whole kernel, not sixteen v_mov in front of every block - and a use in front
    // of the loop: the wait for the tap loads must not end up inside it, where it would be a vmcnt(0) that also drains
    // the prefetched tile in every iteration
    if (RD_MF_SPARSE && DBG != 8 && DBG != 9) asm volatile("" : "+v"(A8[0]), "+v"(A8[1]), "+v"(sidx), "+v"(sidx2));
    else asm volatile("" : "+v"(A8[0]), "+v"(A8[1]), "+v"(dcC));
    // window addresses in the image
    const uint32_t img_addr = rd_lds_addr(img0);
    const uint32_t own = 16 + RD_MF_GROUP_BYTES * (n >> 3) + 16 * (n & 7) + 8 * h;
    const uint32_t prv = n == 0 ? 8 * h
                       : (n & 7) ? own + 16 * 55
                                 : 16 + RD_MF_GROUP_BYTES * ((n >> 3) - 1) + 16 * 63 + 8 * h;
    // Predecessor exchange (rd_mf8_addr): Q (block 0) at 0, P1 (odd blocks) at 1024, P0 (even blocks >= 2) at 2048, two
    // carry slots at 3072 (lane 63's block 7, alternating with the tile parity).
    const uint32_t xb_addr = rd_lds_addr(xb), pend_addr = rd_lds_addr(mypend);
    const uint32_t stage_addr = rd_lds_addr(s_stage[wave]);
    const uint32_t rtoggle = lane == 0 ? 16u : 0u, wtoggle = lane == 63 ? 16u : 0u;
    rd_mf8_addr ad8;
    ad8.xw = xb_addr + 16 * lane;
    ad8.xw7 = xb_addr + (lane == 63 ? 3072 : 1024 + 16 * lane);
    ad8.rd1 = xb_addr + (h ? 1024 + 16 * (lane - 32) : 16 * (lane + 32));
    ad8.rdE = xb_addr + (h ? 2048 + 16 * (lane - 32) : 1024 + 16 * (lane + 32));
    ad8.rdO = xb_addr + (h ? 1024 + 16 * (lane - 32) : 2048 + 16 * (lane + 32));
    ad8.rd0 = xb_addr + (h ? 16 * (lane - 32) : lane ? 1024 + 16 * (lane + 31) : 3072 + 16);
    if (lane < 8) ((uint32_t *)(xb + 3072))[lane] = 0;  // carry slots: finite values from the start

    const uint32_t nwaves = gridDim.x * RD_MF_WAVES;
    const uint32_t wave_id = blockIdx.x * RD_MF_WAVES + wave;
    // RD_K1_STFLAGS & 4096: every wave takes chunks wave_id, wave_id + nwaves, ... (equal shares: A/B); the queue
    // needs chunks of at least four tiles (the id of the following chunk is asked for in a chunk's first
    // iteration, read in its second and first used in its last but one)
    const bool dynamic = !(stflags & 4096) && chunk >= 4;
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

    uint32_t nst = 0;             // tiles staged (wave-uniform)
    uint32_t *st_base = nullptr;  // word 0 of the first staged tile
    bool st_flush = false;        // the staged group ends here (next tile is not the next 64 words)
    // stamp sums (wave-uniform, scalar registers): cycles at the loop-top wait, from there to the last load issued,
    // and in the arithmetic; the waits that follow an iteration with a word store apart
    uint64_t sm_wait = 0, sm_gap = 0, sm_comp = 0, sm_wait_st = 0, sm_t0 = 0, sm_r0 = 0, sm_mark = 0, sm_wmax = 0;
    uint32_t sm_iters = 0, sm_iters_st = 0;
    bool sm_stored = false;
    if (STAMP) { sm_t0 = rd_stamp(); sm_r0 = rd_stamp_real(); }
    if (LOADS && cur.tile < total_tiles) rd_mf_issue(lay, cur.s, cur.ti, img0, lane);
    if (STAMP) sm_mark = rd_stamp();
    while (cur.tile < total_tiles) {
        const uint32_t tile = cur.tile, s = cur.s, ti = cur.ti, inchunk = cur.inchunk;
        uint64_t sm_a = 0;
        if (STAMP) { sm_a = rd_stamp(); sm_comp += sm_a - sm_mark; }
        // this tile has landed (vmcnt counts in issue order; the previous iteration's word store and list flush are
        // older or harmless)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(grab) : : "memory");
        uint64_t sm_b = 0;
        if (STAMP) {
            sm_b = rd_stamp();
            const uint64_t w = sm_b - sm_a;
            sm_wait += w;
            if (w > sm_wmax) sm_wmax = w;
            if (sm_stored) { sm_wait_st += w; sm_iters_st++; }
            sm_iters++;
            sm_stored = false;
        }
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
        rd_mf_read_window(img_addr + prv, img_addr + own, D);
        // the window is in registers: the image takes the next tile
        // stores of finished tiles go out here, before the loads (they share vmcnt, in issue order)
        if (nst == RD_MF_STAGE_TILES || (nst && st_flush)) {
            rd_mf_store_staged(stage_addr, nst, st_base, stflags);
            nst = 0;
            if (STAMP) sm_stored = true;
        }
        const rd_mf_pos nx2 = rd_mf_next(nx1, chunk, tiles_per_stream, nextc);
        // Halo carry: when the tile fetched next is the one that follows this tile in its stream, the 16 bytes in
        // front of it are this tile's last 16 - k-step 8 of column 31, in the registers of lanes 31 and 63 - and go
        // into the image's halo slot by one ds_write_b64 instead of a fifth LDS-DMA instruction (a 16-byte request of
        // its own through the whole memory pipe).  The window read above has completed (its lgkmcnt(0)), the next
        // window read follows in LDS order; the four group loads do not touch the slot.
        bool halo_dma = true;
        {
            const bool follows = nx1.tile < total_tiles && nx1.s == s && nx1.ti == ti + 1;  // wave-uniform
            if (follows) {
                halo_dma = false;
                if (n == 31) asm volatile("ds_write_b64 %0, %1" : : "v"(img_addr + 8 * h), "v"(D[8]) : "memory");
            }
        }
        if (LOADS && nx1.tile < total_tiles) rd_mf_issue(lay, nx1.s, nx1.ti, img0, lane, halo_dma);
        if (!(stflags & 4)) __builtin_amdgcn_s_setprio(0);
        if (STAMP) { sm_mark = rd_stamp(); sm_gap += sm_mark - sm_b; }

        uint32_t word = 0;
        bool slow_taken = false;
        uint32_t fbw8 = 0;  // the word's flag bytes, already gathered
        if (DBG == 2 || DBG == 6) {
#pragma unroll
            for (int j = 0; j < 9; j++) word ^= D[j].x ^ D[j].y;
        } else {
            float *dg = DBG == 3 ? dbg_g + ((size_t)tile * RD_TILE_SAMPLES + 64 * n + 4 * h + 1) * 2 : nullptr;
            const int dleft = RD_TILE_SAMPLES - (64 * n + 4 * h + 1);
            rd_mf8_tile<DBG>(A8, dcC, sidx, sidx2, D, ad8, word, fbw8, slow_taken, dg, dleft);
            ad8.rd0 ^= rtoggle;
            ad8.xw7 ^= wtoggle;
        }
        uint32_t gmask = 0;
        const bool carry = inchunk > 0 && ti > 0;  // previous iteration = previous tile of this stream
        const uint32_t run = ti * 64 + 2 * n + h;  // word index in the stream
        const uint32_t t0 = run * RD_RUN;
        // (the flag bytes are only ever set inside the wave-uniform second-level branches: no ballot in the common case)
        const bool any_flag = slow_taken && rd_mf_any(fbw8 != 0);
        if (any_flag) gmask = ((fbw8 * 0x00204081u) >> 21) & 0xFu;  // bytes 0/1 -> bits
        if (!carry && lane == 0) {  // (the scalar test first: one tile in sixteen)
            if (ti == 0 && !lay.hist_mode) gmask = 0xFu;  // zero history: first run exact
            else gmask |= 1u;                             // no predecessors for the tile's first group
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
            // a ragged last tile of a stream is stored word by word, at once (it sits between the next tile's loads in
            // vmcnt order: the next loop-top wait includes it - once per stream, and only for ragged streams)
            lay.bits[(size_t)s * lay.bits_stride + run] = word;
        } else {
            gmask = 0;
        }
        if (DBG == 1 || DBG == 2 || DBG == 6) gmask = 0;
        const uint64_t fm = (any_flag || !carry) ? __ballot(gmask != 0) : 0;
        if (fm) {
            const uint32_t nf = (uint32_t)__popcll(fm);
            if (npend + nf > RD_MF_PEND) {
                rd_mf_flush(mypend, npend, fix_list, fix_cap, counters, (uint32_t)lay.bits_stride, bucket_cnt);
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
    }
    if (nst) rd_mf_store_staged(stage_addr, nst, st_base, stflags);
    if (npend) rd_mf_flush(mypend, npend, fix_list, fix_cap, counters, (uint32_t)lay.bits_stride, bucket_cnt);
    if (STAMP && dbg_g) {  // a buffer of its own: nothing else in the kernel reads it
        const uint64_t t1 = rd_stamp(), r1 = rd_stamp_real();
        sm_comp += t1 - sm_mark;
        if (lane == 0) {
            uint64_t *o = (uint64_t *)dbg_g + (size_t)wave_id * RD_STAMP_WORDS;
            o[0] = sm_wait; o[1] = sm_gap; o[2] = sm_comp; o[3] = sm_wait_st;
            o[4] = sm_iters; o[5] = sm_iters_st; o[6] = t1 - sm_t0; o[7] = r1 - sm_r0;
            o[8] = sm_wmax; o[9] = sm_r0; o[10] = r1;
            uint32_t xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            o[11] = xcc;
        }
    }
}

static int rd_mf_env(const char *name, int dflt) {
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}

// Launch parameters read once per process.  A function-local static: initialised exactly once, also when several
// threads make their first launch together (handles may be used from several threads, rd_api.hip).
struct rd_mf_params {
    int dbg = 0, stamp = 0, chunk_env = 0, per_cu_env = 0, n_cu = 256;
    uint32_t stflags = 0;
    rd_mf_params() {
        chunk_env = rd_mf_env("RD_K1_CHUNK", 0);          // tuning knobs: results do not depend on them
        per_cu_env = rd_mf_env("RD_K1_WGS_PER_CU", 0);
#ifdef RD_DIAG
        // timing ablations and A/B switches, wrong results for most of them: the diagnostic library only
        dbg = rd_mf_env("RD_K1_DEBUG", 0);
        stamp = rd_mf_env("RD_K1_STAMPS", 0) ? 1 : 0;
        stflags = (uint32_t)rd_mf_env("RD_K1_STFLAGS", 0) & (1u | 2u | 4u | 4096u);
#endif
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            n_cu = prop.multiProcessorCount;
    }
};
static const rd_mf_params &rd_mf_get_params() {
    static const rd_mf_params p;
    return p;
}

#ifdef RD_DIAG
// stamps of the last stamped launch: RD_STAMP_WORDS uint64 per wave (diagnostic library only)
static uint64_t *g_stamp_buf = nullptr;
static uint32_t g_stamp_waves = 0;
extern "C" int rd_diag_read_stamps(uint64_t *out, uint32_t cap_waves, uint32_t *n_waves) {
    if (!g_stamp_buf || !n_waves) return RD_ERR_STATE;
    if (hipDeviceSynchronize() != hipSuccess) return RD_ERR_DEVICE;
    *n_waves = g_stamp_waves;
    const uint32_t n = g_stamp_waves < cap_waves ? g_stamp_waves : cap_waves;
    if (n && hipMemcpy(out, g_stamp_buf, (size_t)n * RD_STAMP_WORDS * 8, hipMemcpyDeviceToHost) != hipSuccess) return RD_ERR_DEVICE;
    return RD_OK;
}
extern "C" int rd_diag_variant(int *dbg, int *stamp, uint32_t *stflags) {
    const rd_mf_params &P = rd_mf_get_params();
    if (dbg) *dbg = P.dbg;
    if (stamp) *stamp = P.stamp;
    if (stflags) *stflags = P.stflags;
    return RD_OK;
}
#endif

struct rd_mf_launch_args {
    rd_layout lay;
    uint32_t tps, total;
    uint64_t total64;
    uint32_t *fix_list, fix_cap, *counters;
    hipStream_t st;
    hipEvent_t ev_start, ev_stop;
    uint32_t *bucket_cnt = nullptr;      // RD_DEMOD_FIX_BUCKETS
    uint32_t *chunk_out = nullptr;
    float *dbg_g;
};

template <int D, bool STAMP>
static void rd_mf_launch_variant(const rd_mf_launch_args &a) {
    const rd_mf_params &P = rd_mf_get_params();
    // persistent grid sized from the occupancy API (registers and LDS of the variant actually launched)
    static const int per_cu_occ = [] {
        int occ = 0;
        const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_demod_mfma<D, STAMP>, RD_MF_WG, 0);
        return (e == hipSuccess && occ >= 1) ? (occ > 8 ? 8 : occ) : 2;
    }();
    // (three workgroups per CU instead of the four that fit: 4 % slower)
    const int per_cu = (P.per_cu_env >= 1 && P.per_cu_env <= 8) ? P.per_cu_env : per_cu_occ;
    // Tiles per chunk: a multiple of the 4-tile store groups.  With the work queues (default) 8: the chunks past a
    // wave's first are handed out on demand, the waves that run faster take more of them, and shorter chunks even the
    // finish out.  With the sparse-instruction kernel, three interleaved rounds on two boxes (profiles/r04_chunk.txt):
    // 8: 0.434-0.445 ms, 10: 0.437, 6: 0.436, 16 (the default until then): 0.451-0.455, 12: 0.455, 4: 0.441; the
    // fix-up list grows (a chunk's first group is always listed: 0.23 % of the runs against 0.13 %) and k_tail with it by
    // 4 us - the step is 1 % shorter.  With equal shares (RD_K1_STFLAGS & 4096) 28 (profiles/r02_chunk_sweep.txt).
    // A small workload gets shorter chunks, down to one store group, until every resident wave has one.
    uint32_t chunk = P.chunk_env > 0 ? (uint32_t)P.chunk_env : ((P.stflags & 4096) ? 28 : 8);
    if (P.chunk_env <= 0) {
        const uint64_t waves = (uint64_t)P.n_cu * per_cu * RD_MF_WAVES;
        while (chunk > RD_MF_STAGE_TILES && a.total64 / chunk < waves) chunk -= RD_MF_STAGE_TILES;
    }
    if (chunk > a.total) chunk = a.total;
    const uint64_t chunks = (a.total64 + chunk - 1) / chunk;
    uint64_t wgs = (chunks + RD_MF_WAVES - 1) / RD_MF_WAVES;
    const uint64_t max_wgs = (uint64_t)P.n_cu * per_cu;
    if (wgs > max_wgs) wgs = max_wgs;
    float *dbg = a.dbg_g;
    uint32_t stf = P.stflags;
    if (a.bucket_cnt && D == 0 && !STAMP) { dbg = (float *)a.bucket_cnt; stf |= RD_STF_BUCKETS; }
    if (a.chunk_out) { a.chunk_out[0] = chunk; a.chunk_out[1] = (uint32_t)wgs * RD_MF_WAVES; }
#ifdef RD_DIAG
    if (STAMP) {
        const uint32_t nw = (uint32_t)wgs * RD_MF_WAVES;
        if (nw > g_stamp_waves || !g_stamp_buf) {
            if (g_stamp_buf) hipFree(g_stamp_buf);
            g_stamp_buf = nullptr;
            if (hipMalloc(&g_stamp_buf, (size_t)nw * RD_STAMP_WORDS * 8) != hipSuccess) g_stamp_buf = nullptr;
        }
        g_stamp_waves = g_stamp_buf ? nw : 0;
        dbg = (float *)g_stamp_buf;
    }
#endif
    if (a.ev_start || a.ev_stop)
        hipExtLaunchKernelGGL((k_demod_mfma<D, STAMP>), dim3((unsigned)wgs), dim3(RD_MF_WG), 0, a.st, a.ev_start,
                              a.ev_stop, 0, a.lay, a.tps, a.total, chunk, a.fix_list, a.fix_cap, a.counters, dbg, stf);
    else
        hipLaunchKernelGGL((k_demod_mfma<D, STAMP>), dim3((unsigned)wgs), dim3(RD_MF_WG), 0, a.st, a.lay, a.tps, a.total,
                           chunk, a.fix_list, a.fix_cap, a.counters, dbg, stf);
}

// Returns false when the fix-up entries went to the one global list although buckets were asked for (the stamped and
// ablated variants of the diagnostic library keep the list): the caller then launches k_fixup as before.
bool rd_launch_demod_mfma(const rd_layout &lay, uint32_t *fix_list, uint32_t fix_cap, uint32_t *counters, hipStream_t st,
                          hipEvent_t ev_start, hipEvent_t ev_stop, float *dbg_g, uint32_t flags, uint32_t *chunk_out,
                          uint32_t *bucket_cnt) {
    rd_mf_launch_args a;
    a.chunk_out = chunk_out;
    a.bucket_cnt = (flags & RD_DEMOD_FIX_BUCKETS) && !dbg_g ? bucket_cnt : nullptr;
    a.lay = lay;
    a.tps = (lay.n_samples + RD_TILE_SAMPLES - 1) / RD_TILE_SAMPLES;
    a.total64 = (uint64_t)lay.n_streams * a.tps;
    if (a.total64 == 0) return a.bucket_cnt != nullptr;
    a.total = (uint32_t)a.total64;
    a.fix_list = fix_list; a.fix_cap = fix_cap; a.counters = counters;
    a.st = st; a.ev_start = ev_start; a.ev_stop = ev_stop; a.dbg_g = dbg_g;
    if (dbg_g) {  // the test hook: raw filter outputs as well
        rd_mf_launch_variant<3, false>(a);
        return false;
    }
#ifdef RD_DIAG
    const rd_mf_params &P = rd_mf_get_params();
    if (P.stamp || P.dbg) {
        a.bucket_cnt = nullptr;   // (these variants keep the one fix-up list: the caller runs k_fixup on it, bounded by fix_cap)
        if (P.stamp && P.dbg == 0) { rd_mf_launch_variant<0, true>(a); return false; }
        switch (P.dbg) {
            case 1: rd_mf_launch_variant<1, false>(a); return false;
            case 2: rd_mf_launch_variant<2, false>(a); return false;
            case 6: rd_mf_launch_variant<6, false>(a); return false;
            case 7: rd_mf_launch_variant<7, false>(a); return false;
            case 8: rd_mf_launch_variant<8, false>(a); return false;
            case 9: rd_mf_launch_variant<9, false>(a); return false;
            case 10: rd_mf_launch_variant<10, false>(a); return false;
            case 11: rd_mf_launch_variant<11, false>(a); return false;
            default:
                fprintf(stderr, "[rd diag] no kernel variant RD_K1_DEBUG=%d RD_K1_STAMPS=%d is compiled in\n", P.dbg, P.stamp);
                abort();
        }
    }
#endif
    rd_mf_launch_variant<0, false>(a);
    return a.bucket_cnt != nullptr;
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
