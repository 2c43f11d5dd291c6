// rd_internal.h - launch interface between rd_api.hip (host logic, C ABI) and rd_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rtldavis_hip.h"
#include "rd_host.h"

// counters[] slots (device uint32)
// FIX: flagged runs; MATCH: preamble matches (= primary records); REC: second records of
// block-boundary positions; PARSED: CRC-valid messages
// OVF: the one-launch tail could not hold this input (1: a stream with more matches than its list, 2: more records
// than the output array has room for, 8: a group's fix-up bucket overflowed, 16: a workgroup gave up waiting for the
// totals of the groups in front of it)
enum { RD_CNT_FIX = 0, RD_CNT_MATCH = 1, RD_CNT_REC = 2, RD_CNT_TASKS = 3, RD_CNT_PARSED = 4, RD_CNT_OVF = 5, RD_CNT_SLOTS = 8 };
// RD_CNT_FIX: entries in the global fix-up list (k_fixup's input; written by k_tail as the sum over its groups' buckets)
// rd_launch_demod flags
#define RD_DEMOD_FIX_BUCKETS 4u    /* the fix-up entries go to per-group buckets (the one-launch tail, k_tail): fix_list =
                                      [groups][fix_cap] entries, bucket_cnt = [groups] counters, group = stream >> RD_FT_GSH */
// The one-launch tail (rd_kernels.hip: k_tail): a workgroup owns RD_FT_STREAMS consecutive streams
#define RD_FT_GSH 2
#define RD_FT_STREAMS (1 << RD_FT_GSH)
struct rd_ft_bufs {
    uint32_t *fixb = nullptr;    // [groups][fix_bcap] fix-up entries, filled by the demod kernel's waves
    uint32_t *fixcnt = nullptr;  // [groups] entries written (lives behind the counter set of the run: cleared with it)
    uint32_t fix_bcap = 0;
    uint32_t bcap = 0;           // matches a stream's list in LDS holds (a multiple of 32, rd_host.h: rd_ord_bucket_cap)
    uint32_t *gstate = nullptr;  // [3][groups]: (seq << 20 | records), matches, fix-up entries per group
};
// Behind the RD_CNT_SLOTS counters the host reads back: the demod kernel's work queues (chunks handed out beyond
// the first one of every wave).  One counter word sustains ~90 atomics per microsecond, so there are RD_NQUEUE of
// them, 256 bytes apart; wave w draws from queue w % RD_NQUEUE, which owns the chunks nwaves + q + RD_NQUEUE k.
#define RD_NQUEUE 32
#define RD_QUEUE_STRIDE 64                                     /* words */
#define RD_CNT_QUEUE0 RD_QUEUE_STRIDE                          /* first queue's word index */
#define RD_CNT_TOTAL (RD_CNT_QUEUE0 + RD_NQUEUE * RD_QUEUE_STRIDE)  /* words per counter set */

// Geometry of the fused demod kernel
#define RD_TILE_SAMPLES 2048  // 64 lanes x 32 samples: one wave iteration
#define RD_TILE_BYTES 4096
#define RD_INPUT_PAD 8192     // bytes readable past the last stream (ragged last tile)

// One set of streams laid out with a fixed byte stride.
struct rd_layout {
    const uint8_t *iq;        // sample 0 of stream 0 (16-byte aligned)
    size_t stream_stride;     // bytes between streams (multiple of 16)
    int n_streams;
    uint32_t n_samples;       // samples per stream handled by this launch
    int hist_mode;            // 0: zero history before sample 0 (after reset);
                              // 1: >= 16 samples of raw history precede each stream's sample 0
    long valid_from;          // first readable sample (0 for hist_mode 0, negative otherwise)
    uint32_t *bits;           // packed sign bits, word w of stream s at bits[s*bits_stride + w]
    size_t bits_stride;       // words
};

// rd_devcfg: rd_host.h (the pure-host translation unit shares it)

struct rd_match {
    int32_t stream;
    int32_t pos;  // bit-array coordinate of the first preamble sample
};

// lazy, fork-aware HIP initialisation (rd_api.hip); RD_OK or RD_ERR_DEVICE
int rd_ensure_device_public(void);

// --- launches (all asynchronous on `st`) ---
// ev_start / ev_stop (optional): events that receive the kernel's own begin / end timestamps.
// flags: RD_DEMOD_FIX_BUCKETS with bucket_cnt (the groups' entry counters; fix_list = [groups][fix_cap] then).  Returns the
// flags the launched kernel honoured (the ablated kernels of the diagnostic library keep the one list: none).
// chunk_out: receives the tiles per chunk the launch used and the number of waves
uint32_t rd_launch_demod(const rd_layout &lay, uint32_t *fix_list, uint32_t fix_cap, uint32_t *counters, hipStream_t st,
                         hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr, uint32_t flags = 0,
                         uint32_t *chunk_out = nullptr, uint32_t *bucket_cnt = nullptr);
// The kernel itself (rd_demod_mfma.hip).  dbg_g (test hook): when given, the kernel also dumps g[tile][2048][2].
// Returns true when the fix-up entries went to the buckets.
bool rd_launch_demod_mfma(const rd_layout &lay, uint32_t *fix_list, uint32_t fix_cap, uint32_t *counters, hipStream_t st,
                          hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr, float *dbg_g = nullptr,
                          uint32_t flags = 0, uint32_t *chunk_out = nullptr, uint32_t *bucket_cnt = nullptr);
// all != 0: re-evaluate every run exactly (used when the guard list overflowed or the
// layout does not meet the fast kernel's alignment requirements).
// zero_next (may be null): RD_CNT_TOTAL words (counters and work queues) to clear for the handle's next run.
// zero_words: how many words at zero_next (the counter set, plus k_tail's per-group bucket counters when the handle
// has them behind it)
// expect (0: unknown): the list length of the handle's previous run - the grid is sized for it (+25 %) instead of
// for the list's capacity (every lane loops with the grid's stride, so a short grid is only slower, never wrong)
void rd_launch_fixup(const rd_layout &lay, const uint32_t *fix_list, uint32_t fix_cap, uint32_t *counters, int all,
                     uint32_t *zero_next, hipStream_t st, uint32_t zero_words = RD_CNT_TOTAL, uint64_t expect = 0);
// Search positions p in [p_lo, p_hi] of every stream's bit array (bits outside [0, n_bits) are 0).
void rd_launch_search(const uint32_t *bits, size_t bits_stride, int n_streams, long n_bits, long p_lo, long p_hi,
                      const rd_devcfg &cfg, rd_match *matches, uint32_t match_cap, uint32_t *counters,
                      hipStream_t st);
// The whole tail in one launch (k_tail): exact bits for the listed groups, search, slice with order and dedupe, RSSI /
// SNR, final records.  seq: 1..4095, different from the previous launch on fb.gstate.  skip_fix: the bits are final.
// Returns 1 when launched, 0 when the shape is not the one the kernel is built for.
int rd_launch_tail_fused(const rd_layout &lay, const rd_devcfg &cfg, int n_calls, long p_lo, long p_hi, const rd_ft_bufs &fb,
                         uint32_t bucket_limit, uint32_t seq, int skip_fix, rd_packet *recs, uint32_t rec_cap, uint32_t *counters,
                         hipStream_t st, hipEvent_t ev_stop = nullptr, uint32_t *zero_next = nullptr, uint32_t zero_words = 0);
// Slice + RSSI/SNR, one wave per match.  batch_mode = 1: position = absolute sample, calls derived
// from it (n_calls blocks from reset).  batch_mode = 0: position = window index q of call `call`,
// lay.iq points at the newest block's first sample.
// recs holds 2 * match_cap entries: recs[i] is match i's record (stream = -1 when no call reports
// it); the second record of a position on a block boundary (q = B in call b and q = 0 in call
// b+1, py:194) goes to recs[match_cap + k], k < RD_CNT_REC.  Per-call duplicates (py:203-205) are
// left to the host, which orders the records anyway.  recs (device memory) and recs_host (the
// device address of pinned, mapped host memory) are both optional: the kernel stores to each that
// is given; with recs_host the records need no device-to-host copy.
// returns 1 when the records were written densely (one per task, RD_CNT_TASKS of them from index 0), else 0
int rd_launch_slice(const rd_layout &lay, const uint32_t *bits, size_t bits_stride, long n_bits, const rd_devcfg &cfg,
                     const rd_match *matches, uint32_t match_cap, int batch_mode, int n_calls, int call,
                     rd_packet *recs, rd_packet *recs_host, uint32_t *counters, hipStream_t st,
                     hipEvent_t ev_stop = nullptr, void *tasks = nullptr);
// tasks: 2 * match_cap entries of RD_TASK_BYTES (the two-kernel form of the batch path; null: one kernel)
#define RD_TASK_BYTES 16
// Parser.parse front half (protocol.py:290-311) over the records of a batch run (layout as
// above): CRC-valid ones are written to `parsed` (RD_CNT_PARSED) with their frequency error.
void rd_launch_parse(const rd_layout &lay, const rd_devcfg &cfg, const rd_packet *recs, uint32_t match_cap,
                     rd_parsed *parsed, uint32_t *counters, hipStream_t st, int dense = 0);
// d[t0 .. t0+n) of stream `stream` in float64
void rd_launch_disc(const rd_layout &lay, int stream, long t0, long n, double *out, hipStream_t st);
// f[t0 .. t0+n) (interleaved re,im) of stream `stream` in float64
void rd_launch_filtered(const rd_layout &lay, int stream, long t0, long n, double *out, hipStream_t st);
// window <- (window >> shift_bits) with `block` (n_block_bits) appended at the top; window has n_win_bits
void rd_launch_window_update(uint32_t *win_out, const uint32_t *win_in, long n_win_bits, const uint32_t *block,
                             long n_block_bits, int n_streams, size_t win_stride, size_t block_stride,
                             hipStream_t st);

// The streaming handle's one-launch block (k_stream_block, rd_kernels.hip): everything of one demodulate() call - ring
// roll, exact bits, window, search, slice + RSSI, results into mapped host memory - for NS streams in lock step.
#define RD_SB_THREADS 1024
#ifndef RD_SBC_THREADS
#define RD_SBC_THREADS 256   // k_stream_block_cplx: a workgroup takes 8 x this many samples of the block
#endif
struct rd_sb_args {
    rd_devcfg cfg;
    uint8_t *ring;            // the streams' raw rings: [hdr 32 B][previous block][newest block], ring_stride apart
    size_t ring_stride;
    const uint8_t *in;        // device address of the pinned input: NS x 2B bytes
    const uint32_t *win_in;   // quantized window before this block (2B bits per stream)
    uint32_t *win_out;        // ... and after it
    rd_packet *recs_host;     // mapped host memory: B + 1 records per stream (record i of a stream = its match i)
    uint32_t *cnt_host;       // mapped: matches per stream
    uint32_t *flag_host;      // mapped: seq per stream, stored last
    uint32_t seq;
    long seen_before;         // blocks since reset
    uint64_t *stamps;         // diagnostic library: phase stamps (else null)
};
// 1 when the kernel was launched, 0 when the configuration is not one it is built for
int rd_launch_stream_block(const rd_sb_args &a, int n_streams, hipStream_t st);

// The same for the complex-input branch (k_stream_block_cplx): one stream, complex128 ring, input from mapped host memory
// as complex128 or (in_is_u8) as bytes through the LUT
struct rd_sbc_args {
    rd_devcfg cfg;
    double *ring;             // [hdr 32 doubles][previous block][newest block]
    const void *in;           // device address of the pinned input: B complex128 or 2B bytes
    int in_is_u8;
    const uint32_t *win_in;
    uint32_t *win_out;
    rd_packet *recs_host;     // mapped host memory: B + 1 records
    uint32_t *cnt_host, *flag_host;
    uint32_t seq;
    long seen_before;
    uint32_t *sync;           // device word, zero between launches: the workgroups count themselves in
    uint64_t *stamps;         // diagnostic library: phase stamps (else null)
};
int rd_launch_stream_block_cplx(const rd_sbc_args &a, hipStream_t st);

// complex128 input path (py:144-150): raw ring of interleaved doubles
struct rd_cplx_layout {
    const double *x;   // sample 0 (interleaved re,im)
    long valid_from;   // first readable sample
    long n;            // samples
};
void rd_launch_cplx_bits(const rd_cplx_layout &lay, uint32_t *bits, hipStream_t st);
void rd_launch_cplx_disc(const rd_cplx_layout &lay, long t0, long n, double *out, hipStream_t st);
void rd_launch_cplx_filtered(const rd_cplx_layout &lay, long t0, long n, double *out, hipStream_t st);
void rd_launch_cplx_slice(const rd_cplx_layout &lay, const uint32_t *bits, long n_bits, const rd_devcfg &cfg,
                          const rd_match *matches, uint32_t match_cap, int call, rd_packet *recs, rd_packet *recs_host,
                          uint32_t *counters, hipStream_t st);
void rd_launch_lut(const uint8_t *in, double *out, size_t n_cplx, hipStream_t st);

// stage kernels on device arrays (float64)
void rd_launch_rotate(const double *in, double *out, size_t n, hipStream_t st);
void rd_launch_fir9(const double *in, double *out, size_t n_out, hipStream_t st);
void rd_launch_discriminate(const double *in, double *out, size_t n_out, hipStream_t st);
void rd_launch_quantize(const double *in, uint8_t *out, size_t n, hipStream_t st);
void rd_launch_pack_bytes(const uint8_t *in01, uint32_t *words, size_t n, hipStream_t st);
