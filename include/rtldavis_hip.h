/*
 * rtldavis_hip.h - C ABI of librtldavis_hip.so: the MI355X (gfx950) implementation of
 * rtldavis's IQ -> bits -> packets path.
 *
 * The reference has no FFI for this path (it is pure Python/NumPy, and the legacy Go
 * twin is pure Go), so each entry point below names the reference function whose work
 * it takes over.  "py" = /root/reference/src/rtldavis/dsp.py, "go" = /root/reference/dsp/dsp.go.
 * The ctypes stub a maintainer of the reference would add is in INTEGRATION.md.
 *
 * Conventions: plain pointers and sizes only; every function returns RD_OK (0) or a
 * negative rd_status; outputs are caller-allocated; no callbacks; no global state other
 * than the per-process HIP context, which is created lazily by the first call that
 * needs the device (never by rd_create / rd_create_multi / rd_batch_create), so a handle
 * may be created before fork() (py worker model: __main__.py:277, worker.py:29); a process
 * forked AFTER the device was used gets RD_ERR_DEVICE instead of undefined behaviour.
 * A handle is used by one thread at a time (like a reference Demodulator); different handles
 * are independent.  There is no CPU fallback: without a usable HIP device every entry point
 * that computes returns RD_ERR_DEVICE.
 * rd_last_error() returns a thread-local, human-readable message for the last failure.
 */
#ifndef RTLDAVIS_HIP_H
#define RTLDAVIS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RD_MAX_PREAMBLE 64   /* symbols */
#define RD_MAX_PKT_BYTES 32  /* (packet_symbols + 7) / 8 */

typedef enum rd_status {
    RD_OK = 0,
    RD_ERR_ARG = -1,      /* bad argument / "Incompatible array sizes" (py:32-36,145-149) */
    RD_ERR_DEVICE = -2,   /* HIP failure: no device, launch or copy error */
    RD_ERR_CAPACITY = -3, /* caller's output array too small; *n holds the needed count */
    RD_ERR_STATE = -4     /* call order violated (e.g. results before run) */
} rd_status;

/* py:101-125 PacketConfig / go:172-218 NewPacketConfig.  Derived constants are computed inside. */
typedef struct rd_config {
    int32_t bit_rate;
    int32_t symbol_length;    /* samples per symbol */
    int32_t preamble_symbols; /* <= RD_MAX_PREAMBLE */
    int32_t packet_symbols;   /* <= 8 * RD_MAX_PKT_BYTES */
    int32_t block_size;       /* samples per demodulate() call; multiple of 4 */
    uint8_t preamble[RD_MAX_PREAMBLE]; /* 0/1 per symbol */
} rd_config;

/* py:12-17 Packet, plus where it came from.  index is the window-relative q_idx of py:190-246. */
typedef struct rd_packet {
    int32_t stream; /* stream number in a batch (0 for rd_handle) */
    int32_t call;   /* block number b of the demodulate() call that returns it */
    int32_t index;
    int32_t nbytes;
    uint8_t data[RD_MAX_PKT_BYTES];
    double rssi;
    double snr;
} rd_packet;

/* One CRC-valid message of protocol.Parser.parse's front half (/root/reference/src/rtldavis/
 * protocol.py:282-318): bytes bit-reversed (:290), CRC-16-CCITT over data[2:] == 0 (:297),
 * frequency error from the mean discriminator output over the preamble window (:304-311). */
typedef struct rd_parsed {
    int32_t stream;
    int32_t call;
    int32_t index;     /* Packet.index of the packet it came from */
    int32_t freq_err;  /* Hz, -int(mean * sample_rate / 2pi) */
    int32_t id;        /* transmitter id: data[2] & 7 after the bit swap (:318) */
    int32_t nbytes;    /* length of data[] = packet bytes - 2 */
    uint8_t data[RD_MAX_PKT_BYTES]; /* bit-swapped message bytes, sync word removed (msg_data, :317) */
    double rssi;
    double snr;
} rd_parsed;

/* Mean per-launch duration of the kernels of the rd_batch_run calls made since timing was
 * enabled or last read (HIP events recorded on each run's stream). */
typedef struct rd_timing {
    float demod_ms;  /* fused LUT+rotate+FIR+discriminator-sign+pack kernel */
    float fixup_ms;  /* exact re-evaluation of guard-band samples */
    float search_ms; /* preamble search */
    float slice_ms;  /* slice + RSSI/SNR */
    float total_ms;
    int32_t runs;    /* runs averaged */
} rd_timing;

const char *rd_last_error(void);
/* Number of visible HIP devices (<0 on error).  Does not create a context. */
int rd_device_count(void);
/* Bind this process/thread's later calls to a device (hipSetDevice). */
int rd_set_device(int device);
/* Deadline of every host-side wait of this library (the polling waits for a run's results, a block's kernels, a
 * device-to-host copy), in milliseconds; returns the previous value.  Default: RD_WAIT_TIMEOUT_MS from the
 * environment, else 10000; ms < 0 restores that default.  A wait that passes its deadline returns RD_ERR_DEVICE
 * ("timed out after ... waiting for ...") instead of spinning for ever, so that the caller's per-block try/except
 * (/root/reference/src/rtldavis/worker.py:56-58: log, drop the block, continue) fires when a kernel never completes.
 * The handle stays usable: a streaming handle drops the blocks in flight (waits for them again, discards their
 * packets) at its next rd_demod_submit / rd_demod_block / rd_reset; a batch handle's next rd_batch_results waits again.
 * ms = 0 makes every wait that is not satisfied at its first poll time out (test hook). */
int rd_set_wait_timeout_ms(int ms);

/* ---------------------------------------------------------------------------------------------
 * Streaming demodulator: one stream, one block per call, state carried across calls.
 * Replaces py:128-253 Demodulator / go:220-310.
 * ------------------------------------------------------------------------------------------- */
typedef struct rd_demod rd_demod;

/* py:129-137 __init__ (no device work; safe before fork). */
int rd_create(const rd_config *cfg, rd_demod **out);
void rd_destroy(rd_demod *h);
/* py:248-253 reset. */
int rd_reset(rd_demod *h);
/*
 * py:139-169 demodulate.  `samples` is either uint8 interleaved I,Q (is_complex = 0,
 * count = 2 * block_size bytes; py:151-152) or complex128 (is_complex = 1, count =
 * block_size elements, interleaved re,im doubles; py:144-150).  Any other count returns
 * RD_ERR_ARG ("Incompatible array sizes").  Packets are written in the reference's order
 * (phase-major search order, per-call dedupe, py:171-205).
 */
int rd_demod_block(rd_demod *h, const void *samples, size_t count, int is_complex, rd_packet *out, int cap, int *n);
/*
 * Several receivers in lock step: n_streams independent Demodulators (one per SDR / hop channel)
 * fed one block each per call, all streams in one set of launches.  State is carried per stream
 * exactly as for rd_create.  iq: uint8 [n_streams][2 * block_size], stream-major; packets carry
 * their stream number and come sorted by (stream, reference order).  This is the shape
 * worker.py:34-54 would take with more than one dongle.
 */
int rd_create_multi(const rd_config *cfg, int n_streams, rd_demod **out);
int rd_demod_blocks(rd_demod *h, const uint8_t *iq, size_t nbytes, rd_packet *out, int cap, int *n);
/*
 * The same call split in two, for a receiver loop that must not wait (replaces the hop
 * runners/rtlsdr.py:100-103 `data_queue.put(samples)` -> worker.py:34-50 `demodulate(samples)`):
 * rd_demod_submit copies the block(s) to where the device takes them from (see rd_set_input_push below; the multi-launch
 * form: a pinned slot and a host-to-device copy on a copy stream) and queues the kernels, then returns; rd_demod_fetch
 * waits (polling) for the OLDEST submitted block and returns its packets exactly as rd_demod_block / rd_demod_blocks would.
 * Two blocks may be in flight (RD_ERR_STATE on a third submit): block i+1's copy overlaps block i's
 * kernels.  count: 2 * block_size * n_streams bytes, or block_size complex128 samples (single stream).
 * The state mirrors below need a quiet handle (everything fetched).
 */
int rd_demod_submit(rd_demod *h, const void *samples, size_t count, int is_complex);
/*
 * How rd_demod_submit hands a block to the device.  Where the whole of device memory is host-visible (PCIe large BAR:
 * hipDeviceAttributeIsLargeBar) and the runtime publishes the device's HDP flush register, the one-launch forms take the
 * block from an uncached DEVICE buffer the host has written it into (memcpy, sfence, HDP flush, launch: 128 KB in ~3 us
 * at ~45 GB/s) instead of reading a pinned host slot across the bus from inside the kernel (~13 GB/s).  Same packets
 * either way.  rd_set_input_push: -1 = use the push where available (default; RD_PUSH_INPUT=0 in the environment turns
 * it off), 0 = never, 1 = as -1; applies to handles whose device state is created afterwards; returns 1 when the push
 * is available on the current device (0 otherwise, < 0 on a device error).  rd_demod_input_mode: 1 when this handle's
 * copied blocks are pushed, 0 when they go through the pinned slot.
 */
int rd_set_input_push(int mode);
int rd_demod_input_mode(rd_demod *h);
/*
 * The same without ANY copy on the way in - SURVEY section 8f-4's "pinned-memory ring replacing the pickled-ndarray
 * queue hop" (/root/reference/src/rtldavis/runners/rtlsdr.py:100-103 `data_queue.put(samples)` ->
 * /root/reference/src/rtldavis/worker.py:37 `data_queue.get()`): rd_demod_register_input pins a buffer the PRODUCER owns
 * (a multiprocessing.shared_memory ring an SDR process fills, rtldavis_amd/ring.py) and maps it into the device;
 * rd_demod_submit_from launches on the block that lies at `offset` bytes into it (16-byte aligned; count as for
 * rd_demod_submit) - the one-launch block reads it across the bus where it is.  The block must stay untouched until its
 * rd_demod_fetch has returned.  One buffer per handle; registering again replaces it, host = NULL unregisters
 * (rd_destroy does too); needs a quiet handle.
 */
int rd_demod_register_input(rd_demod *h, void *host, size_t nbytes);
int rd_demod_submit_from(rd_demod *h, size_t offset, size_t count, int is_complex);
int rd_demod_fetch(rd_demod *h, rd_packet *out, int cap, int *n);
int rd_demod_inflight(rd_demod *h);
/*
 * The packets of the block returned last, again: a call that ended in RD_ERR_CAPACITY has consumed its
 * block (the reference's list is unbounded, py:190-246: up to block_size + 1 positions per call are legal),
 * *n told how many there are - grow the array and fetch them here.  Nothing is lost.
 */
int rd_demod_refetch(rd_demod *h, rd_packet *out, int cap, int *n);
/* discriminated (py:134) of one stream of a multi-stream handle */
int rd_copy_discriminated_stream(rd_demod *h, int stream, double *out, size_t n);
/* Lazily materialised mirrors of the reference's state arrays after the last call:
 * discriminated f64[2*block_size] (py:134), filtered complex128[block_size+1] as
 * interleaved doubles (py:133), quantized uint8[buffer_length] 0/1 (py:135). */
int rd_copy_discriminated(rd_demod *h, double *out, size_t n);
int rd_copy_filtered(rd_demod *h, double *out_interleaved, size_t n_complex);
int rd_copy_quantized(rd_demod *h, uint8_t *out, size_t n);

/* ---------------------------------------------------------------------------------------------
 * Batch demodulator: n_streams independent streams of n_blocks blocks each, all demodulated
 * from reset in one pass.  Output is, per stream and per call, exactly what n_blocks
 * successive Demodulator.demodulate() calls return (py:139-169).
 * ------------------------------------------------------------------------------------------- */
typedef struct rd_batch rd_batch;

int rd_batch_create(const rd_config *cfg, int n_streams, int n_blocks, rd_batch **out);
void rd_batch_destroy(rd_batch *b);
/* Device address of the resident input buffer, uint8 [n_streams][n_blocks*block_size][2]
 * (dense, stream-major), so a producer on the GPU can fill it in place. */
int rd_batch_input_ptr(rd_batch *b, void **dev_ptr, size_t *nbytes);
/* Host -> device copy of the whole input (PCIe; not part of the timed region of bench.py). */
int rd_batch_upload(rd_batch *b, const uint8_t *iq_host, size_t nbytes);
/* The same copy issued on `hip_stream` (hipStream_t, a copy stream of the caller's) without waiting: the handle's next
 * rd_batch_run waits for it on the device, so the upload of one resident batch overlaps the kernels of another
 * (host-fed pipelines: SURVEY section 7 "PCIe vs HBM").  iq_host: pinned memory, untouched until that run's results
 * have been fetched.  The copy queues behind the handle's previous run (which still reads the input). */
int rd_batch_upload_async(rd_batch *b, const uint8_t *iq_host, size_t nbytes, void *hip_stream);
/* Run the whole path on the resident input.  hip_stream: a hipStream_t (NULL = default
 * stream).  Asynchronous; rd_batch_results synchronises. */
int rd_batch_run(rd_batch *b, void *hip_stream);
/* Packets of the last run, sorted by (stream, call, reference order).  *n = count. */
int rd_batch_results(rd_batch *b, rd_packet *out, int cap, int *n);
/* Packed bitstream of one stream: sample t -> byte t/8, bit t%8 (LSB first);
 * nbytes >= (n_blocks*block_size + 7) / 8.  (go:105-113 Pack is the nearest reference stage;
 * the Python reference keeps one byte per bit, py:135.) */
int rd_batch_copy_bits(rd_batch *b, int stream, uint8_t *out, size_t nbytes);
/* Full-precision discriminator output d[t0 .. t0+n) of one stream (py:76-90), float64. */
int rd_batch_copy_discriminated(rd_batch *b, int stream, size_t t0, double *out, size_t n);
/* Parser.parse front half on the device for the whole batch (opt-in, before rd_batch_run):
 * after a run, rd_batch_parsed returns the CRC-valid messages sorted like rd_batch_results. */
int rd_batch_set_parse(rd_batch *b, int enabled);
int rd_batch_parsed(rd_batch *b, rd_parsed *out, int cap, int *n);
/* HIP-event timing on the run's stream.  enabled = 1: the demod kernel and the whole run (its
 * start/stop events and the end-of-run event are attached to the kernel dispatches themselves, so
 * they cost no idle time); 2: every stage (two more events recorded between kernels, ~6 us of GPU
 * idle time each); 0: off.  get_timing synchronises, returns the mean over the runs recorded so
 * far and starts a new window (stages not timed read 0). */
int rd_batch_set_timing(rd_batch *b, int enabled);
int rd_batch_get_timing(rd_batch *b, rd_timing *out);
/* Pipelined completion (opt-in; no counterpart in the reference, whose demodulate() py:128-253 returns its packets
 * synchronously).  enabled = 1: rd_batch_run ends without an event of its own; the run's readback is hung on the
 * stop event of the NEXT demod kernel launched on the same HIP stream, by this handle or any other - an event on a
 * run's last kernel idles the stream for 6-11 us, one on the demod kernel does not.  For callers that keep several
 * runs queued on one stream (bench.py: resident batches demodulated round-robin): a run's results are ready one
 * demod kernel later, the stream never idles.  If nothing is launched behind a run, the first call that needs its
 * results (rd_batch_results, rd_batch_get_timing, ...) records the event then.  A timed pipelined run has no
 * end-of-run event: rd_timing.total_ms covers the runs that have one (0 if none).  0 (default): off. */
int rd_batch_set_pipelined(rd_batch *b, int enabled);
/* Which forms of the kernels the last run took (it is waited for first; no counterpart in the reference - for tests and
 * tools that must know that an opt-in form really ran and did not fall back): a mask of RD_FORM_*. */
#define RD_FORM_ORDERED_TAIL 1u   /* records ordered and deduped on the device */
#define RD_FORM_SECOND_PASS 8u    /* a list or bucket overflowed: search and slice ran a second time, in full */
#define RD_FORM_ONE_LAUNCH_TAIL 16u /* everything behind the demod kernel ran as ONE launch (k_tail; implies ORDERED_TAIL) */
int rd_batch_last_run_forms(rd_batch *b, uint32_t *forms);
/* Counters of the last run: 32-sample runs with at least one 8-sample group re-evaluated
 * exactly (guard band), raw preamble matches. */
int rd_batch_get_counters(rd_batch *b, uint64_t *fixup_runs, uint64_t *matches);

/* ---------------------------------------------------------------------------------------------
 * Stage functions on host arrays (caller-allocated out-params, like the reference's).
 * Each runs its own float64 kernel on the device.
 * ------------------------------------------------------------------------------------------- */
/* py:20-39 ByteToCmplxLUT.execute / go:26-44: n_bytes must equal 2 * n_cplx else RD_ERR_ARG. */
int rd_lut_execute(const uint8_t *in_bytes, size_t n_bytes, double *out_cplx, size_t n_cplx);
/* py:42-49 rotate_fs4 / go:46-63 RotateFs4 (in == out allowed). */
int rd_rotate_fs4(const double *in_cplx, double *out_cplx, size_t n_cplx);
/* py:52-73 fir9 / go:65-83 FIR9: out[i] = sum_m c[m] * in[i+m], n_out <= n_in - 8. */
int rd_fir9(const double *in_cplx, size_t n_in, double *out_cplx, size_t n_out);
/* py:76-90 discriminate / go:85-95 Discriminate: n_out = n_in - 1. */
int rd_discriminate(const double *in_cplx, size_t n_in, double *out, size_t n_out);
/* py:93-98 quantize / go:97-103 Quantize: sign bit. */
int rd_quantize(const double *in, uint8_t *out, size_t n);
/* go:105-113 Pack + go:115-131 Search / py:171-188 _search on a 0/1-per-byte buffer:
 * indices in the reference's order; *n = count (RD_ERR_CAPACITY if > cap). */
int rd_search(const rd_config *cfg, const uint8_t *quantized, size_t n, int32_t *indices, int cap, int *count);

/* ---------------------------------------------------------------------------------------------
 * Wideband front end: one uint8 IQ capture at decim * out_rate samples/s ->
 * one out_rate uint8 IQ stream per channel, e.g. the 51 US hop channels (protocol.py:119-171) out
 * of one 26.88 MS/s capture, written straight into a batch demodulator's input buffer.
 * The reference has no channelizer (it retunes one dongle per hop, runners/rtlsdr.py:51,72):
 * parity is unpinned; the definition is in rtldavis_amd/csrc/rd_channelizer.hip and restated in
 * float64 by oracle/channelizer_oracle.py.  It runs on the matrix cores (f16 MFMA with the taps split
 * into two f16 digits, 2^-22 relative; the 8-bit samples are exact in f16, accumulation is fp32): output
 * bytes may differ from the float64 model by one LSB where the sum lands on a rounding boundary.
 * ------------------------------------------------------------------------------------------- */
typedef struct rd_chan_config {
    int32_t out_rate;   /* Hz per channel (19200 * symbol_length = 268800, protocol.py:309) */
    int32_t decim;      /* wideband rate = decim * out_rate; a multiple of 4 */
    int32_t n_taps;     /* length of the real low-pass prototype */
    int32_t n_channels;
    double gain;        /* applied before the 8-bit quantiser clip(rint(z * 127.6 + 127.4), 0, 255) */
} rd_chan_config;
typedef struct rd_chan rd_chan;

/* taps: n_taps doubles; shift_hz[c]: the wideband frequency (Hz, relative to the capture's centre)
 * that channel c moves to 0 Hz of its output - for rtldavis the channel centre plus out_rate / 4,
 * because the demodulator's Fs/4 rotation (dsp.py:42-49) expects the carrier at -Fs/4.  No device
 * work (safe before fork). */
int rd_chan_create(const rd_chan_config *cfg, const double *taps, const int64_t *shift_hz, rd_chan **out);
void rd_chan_destroy(rd_chan *h);
/* Host -> device copy of a capture (uint8 I,Q interleaved), or the device address of the resident
 * capture buffer (at least n_wide_samples) for a producer on the GPU. */
int rd_chan_upload(rd_chan *h, const uint8_t *wide_iq, size_t nbytes);
int rd_chan_input_ptr(rd_chan *h, size_t n_wide_samples, void **dev_ptr);
/* Channelize output samples 0 .. n_out-1 (n_out <= capture length / decim; zero history before
 * the capture) of every channel into device memory: channel c at dst_dev + c * dst_stream_stride,
 * 2 bytes per sample - rd_batch_input_ptr's layout with stride 2 * n_blocks * block_size.
 * Asynchronous on hip_stream (NULL = default stream). */
int rd_chan_run(rd_chan *h, size_t n_out, void *dst_dev, size_t dst_stream_stride, void *hip_stream);
/* Same, into a host array uint8 [n_channels][n_out][2] (synchronous). */
int rd_chan_run_host(rd_chan *h, size_t n_out, uint8_t *out_host, size_t nbytes);

/* ---------------------------------------------------------------------------------------------
 * Test hooks of the fused demod kernel (rtldavis_amd/csrc/rd_demod_mfma.hip).  Not part of the
 * drop-in surface: tests/test_mfma_model.py and tests/test_gpu_mfma.py use them to check the tap
 * matrix and the raw matrix-pipe outputs against an integer model.
 * ------------------------------------------------------------------------------------------- */
/* The constant A operand: uint16 [2 digits][3 k-steps][64 lanes][8 elements] f16 bit patterns. */
void rd_debug_mfma_taps(uint16_t *out);
/* The A operand of the 8-output formulation (rd_mfma.h, RD_OPT_B8): uint16 [2 k-steps][64 lanes][8 elements]; a row of
 * the 32-row tile is one digit of one component of one of 8 outputs. */
void rd_debug_mfma_taps8(uint16_t *out);
/* the same matrix compressed for the 2:4-sparse matrix instruction: vals[64][8] f16 bit patterns, idx[64] (rd_mfma.h) */
void rd_debug_mfma_taps8s(uint16_t *vals, uint32_t *idx);
/* Run k_demod_mfma alone on host data: g_out float [n_streams * tiles][2048][2] (kernel units),
 * bits_out the packed signs BEFORE the exact fix-up (words per stream = ceil(n_samples / 32)),
 * fix_out / n_fix the fix-up list it produced ((word index << 4) | group mask).  hist_mode: every
 * stream is preceded by hist_bytes of history (stride = hist_bytes + 2 n_samples, multiples of 16). */
int rd_debug_demod_mfma(const uint8_t *iq_host, int n_streams, uint32_t n_samples, int hist_mode,
                        uint32_t hist_bytes, float *g_out, uint32_t *bits_out, uint32_t *fix_out,
                        uint32_t fix_cap, uint32_t *n_fix);

#ifdef __cplusplus
}
#endif
#endif /* RTLDAVIS_HIP_H */
