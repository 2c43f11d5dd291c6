// rd_api.hip - C ABI of librtldavis_hip.so (see include/rtldavis_hip.h).
//
// Host-side orchestration only: buffer management, kernel sequencing, and the per-call
// ordering/dedupe of Demodulator._slice (py:190-205).  All arithmetic of the path runs in
// the kernels of rd_kernels.hip; there is no CPU fallback - without a usable HIP device
// every compute entry point returns RD_ERR_DEVICE.
#include <unistd.h>
#include <immintrin.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <mutex>
#include <map>

#include <chrono>

#include "rd_host.h"
#include "rd_internal.h"
#include "rd_math.h"

// RD_DEBUG_HOST=1: print host-side phase times of rd_batch_results to stderr (diagnostic)
static bool dbg_host() {
    static int v = -1;
    if (v < 0) v = getenv("RD_DEBUG_HOST") ? 1 : 0;
    return v == 1;
}
static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ------------------------------------------------------------------------------------------
// errors / device context
// ------------------------------------------------------------------------------------------
static thread_local std::string g_err;

static int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(RD_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

extern "C" const char *rd_last_error(void) { return g_err.c_str(); }

static pid_t g_hip_pid = 0;  // pid that first touched the device through this library

// HIP state does not survive fork(): a child of a process that already used the device
// must not reuse it.  Handles created before fork hold no device state (lazy init).
// for rd_channelizer.hip (same error slot, same fork rule)
int rd_fail_msg(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
static int ensure_device();
int rd_ensure_device_public(void) { return ensure_device(); }

static int ensure_device() {
    const pid_t me = getpid();
    if (g_hip_pid != 0 && g_hip_pid != me)
        return fail(RD_ERR_DEVICE, "HIP was initialised in parent process %d before fork(); "
                                   "create the device state in the child instead", (int)g_hip_pid);
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(RD_ERR_DEVICE, "no HIP device available (%s); librtldavis_hip has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    g_hip_pid = me;
    return RD_OK;
}

extern "C" int rd_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(RD_ERR_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}

extern "C" int rd_set_device(int device) {
    int rc = ensure_device();
    if (rc) return rc;
    HIPCHK(hipSetDevice(device));
    return RD_OK;
}

// Wait for an event / a stream by polling.  The runtime's blocking waits (hipStreamSynchronize, synchronous
// hipMemcpy) were measured to add 10-20 ms of wake-up latency per call on this platform, an order of magnitude more
// than a whole batch takes on the GPU.  Every wait has a deadline (rd_host.h: rd_waiter; RD_WAIT_TIMEOUT_MS, default
// 10 s): a kernel that never completes must surface as RD_ERR_DEVICE so that the caller's "log, drop the block, go on"
// (/root/reference/src/rtldavis/worker.py:56-58) can fire, not pin a core for ever.
static int wait_event(hipEvent_t ev, const char *what) {
    rd_waiter w(rd_wait_timeout_ms());
    for (;;) {
        const hipError_t e = hipEventQuery(ev);
        if (e == hipSuccess) return RD_OK;
        if (e != hipErrorNotReady) return fail(RD_ERR_DEVICE, "hipEventQuery: %s", hipGetErrorString(e));
        if (!w.relax()) return fail(RD_ERR_DEVICE, "timed out after %.0f ms waiting for %s", w.waited_ms(), what);
    }
}

static int wait_stream(hipStream_t st, const char *what) {
    rd_waiter w(rd_wait_timeout_ms());
    for (;;) {
        const hipError_t e = hipStreamQuery(st);
        if (e == hipSuccess) return RD_OK;
        if (e != hipErrorNotReady) return fail(RD_ERR_DEVICE, "hipStreamQuery: %s", hipGetErrorString(e));
        if (!w.relax()) return fail(RD_ERR_DEVICE, "timed out after %.0f ms waiting for %s", w.waited_ms(), what);
    }
}

extern "C" int rd_set_wait_timeout_ms(int ms) { return (int)rd_wait_timeout_set((double)ms); }

// device -> host copy of n bytes on `st`, completed on return (polling wait)
static int copy_d2h(void *dst, const void *src, size_t n, hipStream_t st) {
    if (n == 0) return RD_OK;
    HIPCHK(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, st));
    return wait_stream(st, "a device-to-host copy");
}

// configuration (py:101-125) and the per-call order / dedupe of _slice (py:171-205): rd_host.cpp
static int make_devcfg(const rd_config *c, rd_devcfg *d) {
    const char *why = "";
    const int rc = rd_make_devcfg(c, d, &why);
    return rc ? fail(rc, "%s", why) : RD_OK;
}
static void order_and_dedupe(const rd_packet *recs, size_t n, int S, rd_order_scratch &sc) { rd_order_and_dedupe(recs, n, S, sc); }

// ------------------------------------------------------------------------------------------
// batch demodulator
// ------------------------------------------------------------------------------------------
struct rd_batch {
    rd_config cfg;
    rd_devcfg dc;
    int n_streams, n_blocks;
    long n_samples;      // per stream
    size_t bits_stride;  // words per stream
    bool dev_ready = false, fast_ok = false, ran = false, timing = false, timing_detail = false;
    bool run_timing = false, run_detail = false;  // the timing mode latched by the run in flight
    int device = -1;      // the device the buffers live on (set by the first device call)
    bool fetched = false;  // batch_finish has seen the last run complete
    uint8_t *d_iq = nullptr;
    size_t iq_bytes = 0;
    uint32_t *d_bits = nullptr, *d_fix = nullptr, *d_cnt = nullptr;
    rd_match *d_matches = nullptr;
    void *d_tasks = nullptr;         // rec_cap entries of RD_TASK_BYTES (two-kernel slice)
    int dense = 0;                   // the last run's records are dense (one per task from index 0)
    rd_order_scratch order;          // host ordering scratch
    rd_packet *d_recs = nullptr;     // rec_cap = 2 * match_cap entries (layout: rd_launch_slice)
    int cnt_set = 0;                 // d_cnt holds two counter sets; a run's fixup kernel clears the other one
    bool parse = false;              // Parser.parse front half on the device (rd_batch_set_parse)
    rd_parsed *d_parsed = nullptr;   // rec_cap entries
    uint32_t fix_cap = 0, match_cap = 0, rec_cap = 0;
    hipStream_t stream = nullptr;
    std::vector<hipEvent_t> evs;  // 5 events per timed run, read back in rd_batch_get_timing
    size_t ev_runs = 0;           // timed runs recorded since the last rd_batch_get_timing
    hipEvent_t *ev = nullptr;     // the current run's five events
    hipEvent_t done = nullptr;    // recorded after the run's readback copies: results wait on it,
                                  // not on the stream, so another batch may already be queued behind
    hipEvent_t kdone = nullptr;   // recorded after the run's last kernel
    hipEvent_t uploaded = nullptr;  // recorded behind an rd_batch_upload_async copy: the next run's kernels wait for it
    bool upload_pending = false;
    hipStream_t copy_stream = nullptr;  // readback runs here, beside the next batch's kernels
    uint32_t h_cnt[RD_CNT_SLOTS] = {};
    uint32_t *h_cnt_pin = nullptr;   // pinned: counters of the run in flight
    rd_packet *h_recs_pin = nullptr; // pinned: records of the run in flight (rec_cap entries)
    uint32_t rec_pin_cap = 0;
    uint32_t spec_recs = 1024;       // records copied back speculatively with the counters
    uint64_t last_fix = 0, last_match = 0;
    rd_timing last_timing = {};
    // dev_order: the run in flight left its records in the reference's order with the per-call duplicates dropped
    // (k_tail), and rd_batch_results only copies them.  tail_off: an input overflowed k_tail's lists - the separate
    // kernels + host ordering until the next upload.
    bool dev_order = false, tail_off = false;
    size_t cnt_stride = RD_CNT_TOTAL;  // words per counter set (k_tail's per-group fix-up counters live behind the counters)
    // The one-launch tail (round 4, rd_launch_tail_fused: k_tail): fix-up, search, slice with order and dedupe, RSSI and
    // the final records in ONE kernel, a workgroup per RD_FT_STREAMS streams; the demod kernel's waves put their fix-up
    // entries into per-group buckets (RD_DEMOD_FIX_BUCKETS) and no k_fixup is launched.  ft_ok: the buffers exist
    // (Davis shape, RD_TAIL_IMPL unset); ft_run: the run in flight used it; its overflows (a stream's match list, a
    // group's fix-up bucket) send this input through the separate kernels (tail_off) and the next upload gets lists
    // twice as long.
    rd_ft_bufs ft;
    bool ft_ok = false, ft_run = false;
    uint32_t ft_seq = 0, ft_limit = 0;   // ft_limit: test hook RD_TEST_BUCKET_CAP (0: none)
    bool ft_sticky = false;              // an input overflowed the longest match lists: uploads no longer re-enable k_tail
    size_t ft_cnt_off = 0;               // the groups' fix-up counters inside a counter set (words)
    // Pipelined completion (rd_batch_set_pipelined): the run's last kernel carries no event; the readback is hung on
    // the stop event of the NEXT demod kernel launched on the same stream (any handle's), see batch_adopt below.
    bool second_pass = false;   // the run in flight needed a second search / slice pass (a list or bucket overflowed)
    uint32_t last_launch[2] = {0, 0};  // tiles per chunk and waves of the last demod launch
    bool pipelined = false;
    bool deferred = false;          // the run in flight has no completion event yet (guarded by g_tail_mx)
    hipEvent_t kfirst = nullptr;    // stop event of this handle's demod launch when it adopts another run untimed
    std::vector<uint8_t> ev_end;    // per timed run: its end-of-run event (ev[4]) was recorded
};

// pipelined completion: launch stream -> the handle whose last run waits there for an adopter (see batch_adopt)
static std::mutex g_tail_mx;
// (keyed by device AND stream: the null stream of one device is not the null stream of another)
typedef std::pair<int, hipStream_t> rd_tail_key;
static std::map<rd_tail_key, rd_batch *> g_tail;

static rd_layout batch_layout(const rd_batch *b) {
    rd_layout l;
    l.iq = b->d_iq;
    l.stream_stride = (size_t)b->n_samples * 2;
    l.n_streams = b->n_streams;
    l.n_samples = (uint32_t)b->n_samples;
    l.hist_mode = 0;
    l.valid_from = 0;
    l.bits = b->d_bits;
    l.bits_stride = b->bits_stride;
    return l;
}

extern "C" int rd_batch_create(const rd_config *cfg, int n_streams, int n_blocks, rd_batch **out) {
    if (!out) return fail(RD_ERR_ARG, "null out");
    rd_devcfg dc;
    int rc = make_devcfg(cfg, &dc);
    if (rc) return rc;
    if (n_streams < 1 || n_blocks < 1) return fail(RD_ERR_ARG, "n_streams and n_blocks must be >= 1");
    const long n = (long)n_blocks * cfg->block_size;
    const uint64_t runs = (uint64_t)n_streams * ((n + RD_RUN - 1) / RD_RUN);
    if (n > 0x7FFFFFF0L || runs > 0x0FFFFFFFull)
        return fail(RD_ERR_ARG, "batch too large: at most 2^28 32-sample runs (8.6e9 samples) per batch");
    rd_batch *b = new rd_batch();
    b->cfg = *cfg;
    b->dc = dc;
    b->n_streams = n_streams;
    b->n_blocks = n_blocks;
    b->n_samples = n;
    b->bits_stride = (size_t)((n + 31) / 32);
    b->fast_ok = ((size_t)n * 2) % 16 == 0;  // 16-byte aligned streams for the LDS-DMA loads
    *out = b;
    return RD_OK;
}

static uint32_t *batch_cnt(const rd_batch *b) { return b->d_cnt + (size_t)b->cnt_set * b->cnt_stride; }

// A handle is tied to the device that was current when its buffers were allocated; every entry point
// makes that device current for the calling thread first (handles may be used from several threads).
static int use_device(int device) {
    if (device >= 0) HIPCHK(hipSetDevice(device));
    return RD_OK;
}

static int batch_alloc(rd_batch *b) {
    if (b->dev_ready) return use_device(b->device);
    int rc = ensure_device();
    if (rc) return rc;
    HIPCHK(hipGetDevice(&b->device));
    const uint64_t runs = (uint64_t)b->n_streams * b->bits_stride;
    b->iq_bytes = (size_t)b->n_streams * b->n_samples * 2;
    // guard-band list: one entry per flagged 32-sample run (~1.5 % of the runs on noise)
    b->fix_cap = (uint32_t)std::min<uint64_t>(runs, std::max<uint64_t>(4096, runs / 8));
    // ~16 raw matches per 33-block stream on noise + one burst; the lists grow on overflow
    b->match_cap = (uint32_t)std::min<uint64_t>((uint64_t)b->n_streams * (8 + 1ull * b->n_blocks) + 1024, 1u << 26);
    // test hook: a deliberately small first list so that the overflow path (grow, search and slice again) runs on
    // ordinary inputs (tests/test_gpu_parity.py::test_match_list_overflow_is_transparent)
    if (const char *e = getenv("RD_TEST_MATCH_CAP")) {
        const long v = atol(e);
        if (v >= 1 && v < (long)b->match_cap) b->match_cap = (uint32_t)v;
    }
    b->rec_cap = 2 * b->match_cap;
    HIPCHK(hipMalloc(&b->d_iq, b->iq_bytes + RD_INPUT_PAD));
    HIPCHK(hipMemset(b->d_iq + b->iq_bytes, 127, RD_INPUT_PAD));
    HIPCHK(hipMalloc(&b->d_bits, runs * sizeof(uint32_t)));
    HIPCHK(hipMalloc(&b->d_fix, (size_t)b->fix_cap * sizeof(uint32_t)));
    {
        // RD_TAIL_IMPL=legacy: the separate kernels (k_fixup, k_search, k_classify + k_rssi_u8) + host ordering instead
        // of the one-launch tail (A/B; also what every shape other than the Davis one takes).  RD_TEST_BUCKET_CAP makes
        // the per-stream match lists short so that the fallback runs on ordinary inputs (test hook)
        const char *ti = getenv("RD_TAIL_IMPL");
        const bool legacy = (ti && ti[0] == 'l') || getenv("RD_SLICE_IMPL");  // (an explicit slice form means the separate kernels)
        b->ft_ok = !legacy && b->fast_ok && b->dc.S == 14 && b->dc.P == 16 && b->dc.K == 80 && b->dc.pre_mask == 0x91D3ull &&
                   b->n_samples < (1l << 30) && b->dc.B < (1 << 24) && b->bits_stride % 4 == 0 && b->dc.L >= b->dc.B &&
                   (long)(b->n_blocks + 1) * b->dc.B - b->dc.L >= 0 &&   // (a batch shorter than a packet has no position to report)
                   ((long)(b->n_blocks + 1) * b->dc.B - b->dc.L) / 32 + 16 <= (long)b->bits_stride;
        if (const char *e = getenv("RD_TEST_BUCKET_CAP")) {
            const long v = atol(e);
            if (v >= 1 && v < RD_BUCKET_MIN) b->ft_limit = (uint32_t)v;
        }
    }
    b->cnt_stride = RD_CNT_TOTAL;
    if (b->ft_ok) {  // the groups' fix-up counters live behind the per-stream match counters: cleared with the set
        const size_t groups = ((size_t)b->n_streams + RD_FT_STREAMS - 1) / RD_FT_STREAMS;
        b->ft_cnt_off = b->cnt_stride;
        b->cnt_stride += (groups + 3) & ~(size_t)3;
        // a group's bucket: the first group of every chunk of tiles (chunks of four tiles at the least) and the first
        // run of each of its streams are always listed; twice that, plus room for the groups inside the guard band
        const uint64_t tps = ((uint64_t)b->n_samples + RD_TILE_SAMPLES - 1) / RD_TILE_SAMPLES;
        b->ft.fix_bcap = (uint32_t)std::min<uint64_t>((2 * RD_FT_STREAMS * (tps / 4 + 8) + 63) & ~63ull, 1u << 20);
        if (const char *e = getenv("RD_TEST_FIX_BCAP")) {  // test hook: small buckets, so that their overflow path runs on ordinary inputs
            const long v = atol(e);
            if (v >= 1 && v < (long)b->ft.fix_bcap) b->ft.fix_bcap = (uint32_t)v;
        }
        b->ft.bcap = rd_ord_bucket_cap(b->n_samples);
        HIPCHK(hipMalloc(&b->ft.fixb, groups * b->ft.fix_bcap * sizeof(uint32_t)));
        HIPCHK(hipMalloc(&b->ft.gstate, 3 * groups * sizeof(uint32_t)));
        HIPCHK(hipMemset(b->ft.gstate, 0, 3 * groups * sizeof(uint32_t)));
    }
    HIPCHK(hipMalloc(&b->d_cnt, 2 * b->cnt_stride * sizeof(uint32_t)));
    HIPCHK(hipMemset(b->d_cnt, 0, 2 * b->cnt_stride * sizeof(uint32_t)));
    HIPCHK(hipMalloc(&b->d_matches, (size_t)b->match_cap * sizeof(rd_match)));
    HIPCHK(hipMalloc(&b->d_recs, (size_t)b->rec_cap * sizeof(rd_packet)));
    HIPCHK(hipMalloc(&b->d_tasks, (size_t)b->rec_cap * RD_TASK_BYTES));
    HIPCHK(hipHostMalloc((void **)&b->h_cnt_pin, RD_CNT_SLOTS * sizeof(uint32_t), hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void **)&b->h_recs_pin, (size_t)b->rec_cap * sizeof(rd_packet), hipHostMallocDefault));
    b->rec_pin_cap = b->rec_cap;
    HIPCHK(hipEventCreateWithFlags(&b->done, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&b->kdone, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&b->kfirst, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&b->uploaded, hipEventDisableTiming));
    HIPCHK(hipStreamCreateWithFlags(&b->copy_stream, hipStreamNonBlocking));
    b->dev_ready = true;
    return RD_OK;
}

extern "C" void rd_batch_destroy(rd_batch *b) {
    if (!b) return;
    {
        std::lock_guard<std::mutex> g(g_tail_mx);  // a run still waiting to be adopted dies with its handle
        auto it = g_tail.find(rd_tail_key(b->device, b->stream));
        if (it != g_tail.end() && it->second == b) g_tail.erase(it);
        b->deferred = false;
    }
    if (b->dev_ready && g_hip_pid == getpid()) {
        if (b->device >= 0) hipSetDevice(b->device);
        if (b->ran && !b->fetched) hipDeviceSynchronize();  // kernels of a run nobody fetched may still use the buffers
        hipFree(b->d_iq); hipFree(b->d_bits); hipFree(b->d_fix); hipFree(b->d_cnt);
        hipFree(b->d_matches); hipFree(b->d_recs); hipFree(b->d_tasks);
        hipFree(b->d_parsed);
        hipFree(b->ft.fixb); hipFree(b->ft.gstate);
        hipHostFree(b->h_cnt_pin); hipHostFree(b->h_recs_pin);
        if (b->done) hipEventDestroy(b->done);
        if (b->kdone) hipEventDestroy(b->kdone);
        if (b->kfirst) hipEventDestroy(b->kfirst);
        if (b->uploaded) hipEventDestroy(b->uploaded);
        if (b->copy_stream) hipStreamDestroy(b->copy_stream);
        for (auto &e : b->evs) if (e) hipEventDestroy(e);
    }
    delete b;
}

extern "C" int rd_batch_input_ptr(rd_batch *b, void **dev_ptr, size_t *nbytes) {
    if (!b || !dev_ptr) return fail(RD_ERR_ARG, "null argument");
    int rc = batch_alloc(b);
    if (rc) return rc;
    *dev_ptr = b->d_iq;
    if (nbytes) *nbytes = b->iq_bytes;
    return RD_OK;
}

static int batch_flush(rd_batch *b);

extern "C" int rd_batch_upload(rd_batch *b, const uint8_t *iq_host, size_t nbytes) {
    if (!b || !iq_host) return fail(RD_ERR_ARG, "null argument");
    int rc = batch_alloc(b);
    if (rc) return rc;
    if (nbytes != b->iq_bytes) {
        return fail(RD_ERR_ARG, "Incompatible array sizes: got %zu bytes, expected %zu", nbytes, b->iq_bytes);
    }
    if (b->ran && !b->fetched) {  // a run nobody fetched (e.g. its wait timed out) still reads the input
        rc = batch_flush(b);
        if (rc) return rc;
        rc = wait_event(b->done, "the previous run before an upload");
        if (rc) return rc;
    }
    HIPCHK(hipMemcpy(b->d_iq, iq_host, nbytes, hipMemcpyHostToDevice));
    if (!b->ft_sticky) b->tail_off = false;  // a new input: the one-launch tail gets its chance again
    return RD_OK;
}

// Host-fed batches without the serial upload (SURVEY section 7 "PCIe vs HBM"): the copy goes to `hip_stream` (a copy
// stream of the caller's) and returns at once; the handle's next rd_batch_run waits for it ON THE DEVICE, so the upload
// of one resident batch runs beside the kernels of another.  `iq_host` should be pinned (a pageable buffer makes the
// runtime stage the copy and the call block) and must stay untouched until that run has been launched and the copy
// has completed (rd_batch_results of that run is late enough).  The copy itself waits for the handle's previous run.
extern "C" int rd_batch_upload_async(rd_batch *b, const uint8_t *iq_host, size_t nbytes, void *hip_stream) {
    if (!b || !iq_host) return fail(RD_ERR_ARG, "null argument");
    int rc = batch_alloc(b);
    if (rc) return rc;
    if (nbytes != b->iq_bytes) return fail(RD_ERR_ARG, "Incompatible array sizes: got %zu bytes, expected %zu", nbytes, b->iq_bytes);
    hipStream_t cs = (hipStream_t)hip_stream;
    if (b->ran && !b->fetched) {  // the previous run still reads the input: the copy queues behind its completion
        rc = batch_flush(b);
        if (rc) return rc;
        HIPCHK(hipStreamWaitEvent(cs, b->done, 0));
    }
    HIPCHK(hipMemcpyAsync(b->d_iq, iq_host, nbytes, hipMemcpyHostToDevice, cs));
    HIPCHK(hipEventRecord(b->uploaded, cs));
    b->upload_pending = true;
    if (!b->ft_sticky) b->tail_off = false;
    return RD_OK;
}

// results come back with the run: counters plus as many records as the last run produced
// (+25 %); rd_batch_results fetches the remainder if this run produced more.
// The copies run on their own stream so that another batch's kernels queued on `st` need
// not wait for them.
static int batch_readback(rd_batch *b, hipEvent_t after) {
    HIPCHK(hipStreamWaitEvent(b->copy_stream, after, 0));
    HIPCHK(hipMemcpyAsync(b->h_cnt_pin, batch_cnt(b), RD_CNT_SLOTS * sizeof(uint32_t), hipMemcpyDeviceToHost,
                          b->copy_stream));
    const uint32_t spec = std::min(b->dev_order ? b->rec_cap : b->match_cap, b->spec_recs);
    if (spec)
        HIPCHK(hipMemcpyAsync(b->h_recs_pin, b->d_recs, (size_t)spec * sizeof(rd_packet), hipMemcpyDeviceToHost,
                              b->copy_stream));
    HIPCHK(hipEventRecord(b->done, b->copy_stream));
    return RD_OK;
}

// Pipelined completion.  An event on a run's last kernel leaves the GPU idle for 6-11 us before the next kernel
// starts (profiles/r02_event_gap.txt); an event on the demod kernel costs nothing measurable (the fix-up kernel
// starts as it ends).  A pipelined handle therefore launches its tail without an event and waits to be ADOPTED: the
// next run launched on the same stream - this handle's or another's - gives its demod kernel a stop event and the
// adopted run's readback waits for that one (kernels of a stream run in order: when the next demod kernel has ended,
// the adopted run's tail has).  The records arrive one demod kernel later; the stream never idles.  If nothing is
// launched behind it, the first call that needs the results records an event on the stream as before.
// g_tail: launch stream -> the handle waiting there; every transition of `deferred` happens under g_tail_mx (the
// adopter may be another thread's handle).

static int batch_flush_locked(rd_batch *p) {  // nobody adopted it: an event of its own
    if (!p->deferred) return RD_OK;
    auto it = g_tail.find(rd_tail_key(p->device, p->stream));
    if (it != g_tail.end() && it->second == p) g_tail.erase(it);
    p->deferred = false;
    int rc = use_device(p->device);
    if (rc) return rc;
    HIPCHK(hipEventRecord(p->kdone, p->stream));
    return batch_readback(p, p->kdone);
}
static int batch_flush(rd_batch *b) {
    std::lock_guard<std::mutex> g(g_tail_mx);
    return batch_flush_locked(b);
}
static bool batch_stream_has_tail(int device, hipStream_t st) {
    std::lock_guard<std::mutex> g(g_tail_mx);
    return g_tail.find(rd_tail_key(device, st)) != g_tail.end();
}
// `carrier`: recorded when a kernel launched on `st` after the waiting run's tail has ended
static int batch_adopt(int device, hipStream_t st, hipEvent_t carrier) {
    std::lock_guard<std::mutex> g(g_tail_mx);
    auto it = g_tail.find(rd_tail_key(device, st));
    if (it == g_tail.end()) return RD_OK;
    rd_batch *p = it->second;
    g_tail.erase(it);
    p->deferred = false;
    return batch_readback(p, carrier);
}

// search + slice part of a run (re-issued on list overflow: then with an event of its own, may_defer = false)
static int batch_search_slice(rd_batch *b, hipStream_t st, bool may_defer = false, uint32_t *zero_next = nullptr) {
    const rd_layout lay = batch_layout(b);
    const long B = b->dc.B, L = b->dc.L;
    const bool last_on_slice = !b->parse;
    const bool defer = may_defer && b->pipelined && last_on_slice && !(b->run_timing && b->run_detail);
    // the run's last kernel carries the end-of-run event itself when nothing follows it
    hipEvent_t last = defer ? nullptr : b->run_timing ? b->ev[4] : b->kdone;
    if (b->run_timing && may_defer) b->ev_end.push_back(defer ? 0 : 1);
    b->dev_order = false;
    if (b->ft_run && may_defer) {  // the run's first pass: everything behind the demod kernel in one launch
        rd_ft_bufs fb = b->ft;
        fb.fixcnt = batch_cnt(b) + b->ft_cnt_off;
        b->ft_seq = b->ft_seq % 4095u + 1u;
        const int ok = rd_launch_tail_fused(lay, b->dc, b->n_blocks, B - L, (long)(b->n_blocks + 1) * B - L, fb,
                                            b->ft_limit ? b->ft_limit : fb.bcap, b->ft_seq, 0, b->d_recs, b->rec_cap, batch_cnt(b), st,
                                            last_on_slice ? last : nullptr, zero_next, (uint32_t)b->cnt_stride);
        if (!ok) return fail(RD_ERR_STATE, "the one-launch tail refused a shape its buffers were allocated for");
        b->dev_order = true;   // (the records are final: order and dedupe happened on the device)
        b->dense = 1;
    }
    if (!b->dev_order) {
        rd_launch_search(b->d_bits, b->bits_stride, b->n_streams, b->n_samples, B - L, (long)(b->n_blocks + 1) * B - L,
                         b->dc, b->d_matches, b->match_cap, batch_cnt(b), st);
        if (b->run_timing && b->run_detail) HIPCHK(hipEventRecord(b->ev[3], st));
        b->dense = rd_launch_slice(lay, b->d_bits, b->bits_stride, b->n_samples, b->dc, b->d_matches, b->match_cap, 1,
                                   b->n_blocks, 0, b->d_recs, nullptr, batch_cnt(b), st, last_on_slice ? last : nullptr,
                                   b->d_tasks);
    }
    if (b->parse) {
        if (!b->d_parsed) HIPCHK(hipMalloc(&b->d_parsed, (size_t)b->rec_cap * sizeof(rd_parsed)));
        rd_launch_parse(lay, b->dc, b->d_recs, b->match_cap, b->d_parsed, batch_cnt(b), st, b->dense);
    }
    if (defer) {  // the readback is enqueued by whoever launches next on this stream (batch_adopt) or by batch_flush
        std::lock_guard<std::mutex> g(g_tail_mx);
        auto it = g_tail.find(rd_tail_key(b->device, st));
        if (it != g_tail.end() && it->second != b) {  // (cannot happen after rd_batch_run's adoption; kept safe)
            int rc = batch_flush_locked(it->second);
            if (rc) return rc;
        }
        g_tail[rd_tail_key(b->device, st)] = b;
        b->deferred = true;
        HIPCHK(hipGetLastError());
        return RD_OK;
    }
    // Every event recorded between two kernels idles the GPU for a few microseconds, so a timed run's end-of-run
    // event doubles as the copy stream's trigger.
    if (!last_on_slice) HIPCHK(hipEventRecord(last, st));
    int rc = batch_readback(b, last);
    if (rc) return rc;
    HIPCHK(hipGetLastError());
    return RD_OK;
}

extern "C" int rd_batch_run(rd_batch *b, void *hip_stream) {
    if (!b) return fail(RD_ERR_ARG, "null batch");
    int rc = batch_alloc(b);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)hip_stream;
    const rd_layout lay = batch_layout(b);
    // the previous run's readback of d_cnt / d_recs must be over before they are rewritten: the host
    // has already waited for it if the results were fetched (the normal order), else the stream waits
    if (b->ran && !b->fetched) {
        // (a pipelined run nobody adopted: its readback is not even enqueued yet.  b->stream is still the stream THAT
        // run was launched on - the key of its g_tail entry and where its completion event belongs - and `done`, recorded
        // behind its readback, orders this run's kernels on `st` after it whichever stream that was)
        rc = batch_flush(b);
        if (rc) return rc;
        HIPCHK(hipStreamWaitEvent(st, b->done, 0));
    }
    b->stream = st;
    if (b->upload_pending) {  // rd_batch_upload_async: the input is on its way
        HIPCHK(hipStreamWaitEvent(st, b->uploaded, 0));
        b->upload_pending = false;
    }
    b->fetched = false;
    // counters: this run uses the set the previous run's fixup kernel cleared (both start at zero)
    b->cnt_set ^= 1;
    uint32_t *cnt = batch_cnt(b), *cnt_next = b->d_cnt + (size_t)(b->cnt_set ^ 1) * b->cnt_stride;
    b->run_timing = b->timing;  // set_timing between run() and results() does not touch the run in flight
    b->run_detail = b->timing_detail;
    if (b->timing) {
        if (b->evs.size() < 5 * (b->ev_runs + 1)) {
            const size_t old = b->evs.size();
            b->evs.resize(old + 5, nullptr);
            for (size_t i = old; i < b->evs.size(); i++) HIPCHK(hipEventCreate(&b->evs[i]));
        }
        b->ev = &b->evs[5 * b->ev_runs];
        b->ev_runs++;
    }
    // a pipelined run waiting on this stream is adopted by this run's demod kernel (its stop event)
    const bool adopt = batch_stream_has_tail(b->device, st);
    // the one-launch tail: the fix-up entries go to its workgroups' buckets
    const bool want_ft = b->ft_ok && b->fast_ok && !b->tail_off && !(b->timing && b->timing_detail);
    const uint32_t dflags = want_ft ? RD_DEMOD_FIX_BUCKETS : 0u;
    uint32_t *fixl = want_ft ? b->ft.fixb : b->d_fix, *bcnt = want_ft ? cnt + b->ft_cnt_off : nullptr;
    const uint32_t fixc = want_ft ? b->ft.fix_bcap : b->fix_cap;
    uint32_t honoured = 0;
    if (b->timing && b->fast_ok) {
        // the demod kernel's dispatch carries its own start / stop events (no marker packets)
        honoured = rd_launch_demod(lay, fixl, fixc, cnt, st, b->ev[0], b->ev[1], dflags, b->last_launch, bcnt);
        if (adopt && (rc = batch_adopt(b->device, st, b->ev[1]))) return rc;
    } else if (b->fast_ok && adopt) {
        honoured = rd_launch_demod(lay, fixl, fixc, cnt, st, nullptr, b->kfirst, dflags, b->last_launch, bcnt);
        if ((rc = batch_adopt(b->device, st, b->kfirst))) return rc;
    } else {
        if (b->timing) HIPCHK(hipEventRecord(b->ev[0], st));
        if (b->fast_ok) honoured = rd_launch_demod(lay, fixl, fixc, cnt, st, nullptr, nullptr, dflags, b->last_launch, bcnt);
        if (b->timing) HIPCHK(hipEventRecord(b->ev[1], st));
        if (adopt) {  // (an event of THIS run: b->ev is null - or a finished run's - when the handle is untimed)
            hipEvent_t carrier = b->timing ? b->ev[1] : b->kfirst;
            if (!b->timing) HIPCHK(hipEventRecord(carrier, st));
            if ((rc = batch_adopt(b->device, st, carrier))) return rc;
        }
    }
    b->second_pass = false;
    b->ft_run = b->fast_ok && (honoured & RD_DEMOD_FIX_BUCKETS);   // (k_tail does the fix-up and clears the next counter set)
    if (!b->ft_run)   // (fixl / fixc: where this run's demod kernel put its list - the groups' bucket space when an ablated kernel
                      // of the diagnostic library declined the buckets; bounded by fixc either way)
        rd_launch_fixup(lay, fixl, fixc, cnt, b->fast_ok ? 0 : 1, cnt_next, st, (uint32_t)b->cnt_stride, b->last_fix);
    if (b->timing && b->timing_detail) HIPCHK(hipEventRecord(b->ev[2], st));
    rc = batch_search_slice(b, st, true, b->ft_run ? cnt_next : nullptr);
    if (rc) return rc;
    b->ran = true;
    return RD_OK;
}

// Synchronise, handle list overflows (re-running what is needed), fetch the counters.
static int batch_finish(rd_batch *b) {
    if (!b->ran) return fail(RD_ERR_STATE, "rd_batch_run has not been called");
    hipStream_t st = b->stream;
    {
        int frc = batch_flush(b);  // pipelined and not adopted: nothing was launched behind this run
        if (frc) return frc;
    }
    for (int attempt = 0; attempt < 8; attempt++) {
        const double ta = now_ms();
        int wrc = wait_event(b->done, "the batch run's results");
        if (wrc) return wrc;
        memcpy(b->h_cnt, b->h_cnt_pin, sizeof b->h_cnt);
        if (dbg_host()) fprintf(stderr, "[rd] finish: wait %.3f ms\n", now_ms() - ta);
        bool redo_search = false;
        if (b->ft_run && !b->second_pass && (b->h_cnt[RD_CNT_OVF] & (8u | 16u))) {
            // a group's fix-up bucket overflowed (quiet or degenerate input), or a workgroup gave up waiting for the
            // groups in front of it: every run is re-evaluated exactly, then the separate kernels
            const rd_layout lay = batch_layout(b);
            rd_launch_fixup(lay, b->d_fix, b->fix_cap, batch_cnt(b), 1, nullptr, st);
            b->last_fix = (uint64_t)b->n_streams * b->bits_stride;
            b->tail_off = true;
            b->h_cnt[RD_CNT_OVF] = 0;
            b->h_cnt[RD_CNT_FIX] = 0;
            redo_search = true;
        } else if (b->ft_run && !b->second_pass && (b->h_cnt[RD_CNT_OVF] & 1u)) {
            // a stream with more matches than its list in LDS holds: this input takes the separate kernels, the next
            // upload gets lists twice as long
            if (!b->ft_limit && b->ft.bcap < RD_BUCKET_MAX) b->ft.bcap = std::min<uint32_t>(2 * b->ft.bcap, RD_BUCKET_MAX);
            else if (!b->ft_limit) b->ft_sticky = true;  // (the longest lists there are: inputs like this one keep the separate kernels)
        }
        if (redo_search) {
        } else if (b->fast_ok && !b->ft_run && b->h_cnt[RD_CNT_FIX] > b->fix_cap) {
            // guard list overflowed (degenerate input): re-evaluate every run exactly
            const rd_layout lay = batch_layout(b);
            rd_launch_fixup(lay, b->d_fix, b->fix_cap, batch_cnt(b), 1, nullptr, st);
            // mark handled: this branch must not see the count again
            uint32_t cap = b->fix_cap;
            HIPCHK(hipMemcpyAsync(batch_cnt(b) + RD_CNT_FIX, &cap, sizeof cap, hipMemcpyHostToDevice, st));
            b->last_fix = (uint64_t)b->n_streams * b->bits_stride;
            redo_search = true;
        } else if (attempt == 0) {
            b->last_fix = (uint64_t)b->h_cnt[RD_CNT_FIX];
        }
        if (redo_search && b->tail_off) b->dev_order = false;  // (the bucket-overflow branch above: not the record-overflow one below)
        if (b->dev_order && b->h_cnt[RD_CNT_OVF]) {
            // a stream with more matches than its list holds (or more records than the output array): this input goes
            // through the separate kernels and the host-side ordering, now and until the next upload
            b->tail_off = true;
            if (dbg_host())
                fprintf(stderr, "[rd] one-launch tail overflow: flags %u (1 a stream's match list, 2 records), matches %u\n",
                        b->h_cnt[RD_CNT_OVF], b->h_cnt[RD_CNT_MATCH]);
            const uint32_t zero[5] = {0, 0, 0, 0, 0};  // matches, boundary records, tasks, parsed, overflow
            HIPCHK(hipMemcpyAsync(batch_cnt(b) + RD_CNT_MATCH, zero, sizeof zero, hipMemcpyHostToDevice, st));
            b->second_pass = true;
            int rc2 = batch_search_slice(b, st);
            if (rc2) return rc2;
            continue;
        }
        if (!b->dev_order && b->h_cnt[RD_CNT_MATCH] > b->match_cap) {
            // every pointer is cleared as it is freed: should one of the allocations below fail, rd_batch_destroy
            // must not free anything twice (and the handle reports the failure on every later call)
            hipFree(b->d_matches); b->d_matches = nullptr;
            hipFree(b->d_recs); b->d_recs = nullptr;
            hipHostFree(b->h_recs_pin); b->h_recs_pin = nullptr;
            b->rec_pin_cap = 0;
            hipFree(b->d_tasks); b->d_tasks = nullptr;
            hipFree(b->d_parsed); b->d_parsed = nullptr;
            const uint32_t want = b->h_cnt[RD_CNT_MATCH] + b->h_cnt[RD_CNT_MATCH] / 4 + 1024;
            b->match_cap = 0;  // until the new lists exist
            b->rec_cap = 0;
            b->ran = false;    // a failed re-allocation leaves a handle that must be run again
            const uint32_t new_rec_cap = 2 * want;
            HIPCHK(hipMalloc(&b->d_matches, (size_t)want * sizeof(rd_match)));
            HIPCHK(hipMalloc(&b->d_recs, (size_t)new_rec_cap * sizeof(rd_packet)));
            HIPCHK(hipMalloc(&b->d_tasks, (size_t)new_rec_cap * RD_TASK_BYTES));
            HIPCHK(hipHostMalloc((void **)&b->h_recs_pin, (size_t)new_rec_cap * sizeof(rd_packet), hipHostMallocDefault));
            b->match_cap = want;
            b->rec_cap = new_rec_cap;
            b->rec_pin_cap = new_rec_cap;
            b->ran = true;
            redo_search = true;
        }
        if (!redo_search) {
            b->last_match = b->h_cnt[RD_CNT_MATCH];
            b->fetched = true;
            return RD_OK;
        }
        const uint32_t zero[5] = {0, 0, 0, 0, 0};  // matches, boundary records, tasks, parsed, overflow flags
        HIPCHK(hipMemcpyAsync(batch_cnt(b) + RD_CNT_MATCH, zero, sizeof zero, hipMemcpyHostToDevice, st));
        b->second_pass = true;
        int rc2 = batch_search_slice(b, st);
        if (rc2) return rc2;
    }
    return fail(RD_ERR_DEVICE, "result lists kept overflowing");
}

extern "C" int rd_batch_results(rd_batch *b, rd_packet *out, int cap, int *n) {
    if (!b || !n) return fail(RD_ERR_ARG, "null argument");
    const double t0 = now_ms();
    int rc = batch_finish(b);
    if (rc) return rc;
    const double t1 = now_ms();
    // one record per match, plus the second records of block-boundary positions (kept apart on
    // the device, appended here)
    // sparse layout: one slot per match + the block-boundary twins behind match_cap; dense: one record per task
    const uint32_t nprim = b->dense ? std::min(b->h_cnt[RD_CNT_TASKS], b->rec_cap) : std::min(b->h_cnt[RD_CNT_MATCH], b->match_cap);
    const uint32_t nextra = b->dense ? 0u : std::min(b->h_cnt[RD_CNT_REC], b->match_cap);
    const uint32_t have = std::min(b->dev_order ? b->rec_cap : b->match_cap, b->spec_recs);
    // Late copies go to the copy stream: it has already waited for this run's last kernel, whereas the
    // launch stream may hold the NEXT batch's run by now (two resident batches alternate in bench.py) and a
    // copy queued there would wait for it.
    if (nprim > have) {  // more records than were copied back with the run: fetch the rest
        rc = copy_d2h(b->h_recs_pin + have, b->d_recs + have, (size_t)(nprim - have) * sizeof(rd_packet), b->copy_stream);
        if (rc) return rc;
    }
    if (nextra) {  // second records of block-boundary positions (q == B in one call, q == 0 in the next)
        rc = copy_d2h(b->h_recs_pin + nprim, b->d_recs + b->match_cap, (size_t)nextra * sizeof(rd_packet), b->copy_stream);
        if (rc) return rc;
    }
    const uint32_t nrec = nprim + nextra;
    b->spec_recs = std::max<uint32_t>(1024, nprim + nprim / 4);
    const double t2 = now_ms();
    if (b->dev_order) {  // the device wrote them in the reference's order, duplicates dropped: copy, nothing else
        *n = (int)nrec;
        if ((int)nrec > cap) return fail(RD_ERR_CAPACITY, "need room for %u packets", nrec);
        if (nrec) {
            if (!out) return fail(RD_ERR_ARG, "null out");
            memcpy(out, b->h_recs_pin, (size_t)nrec * sizeof(rd_packet));
        }
        if (dbg_host())
            fprintf(stderr, "[rd] results (k_tail): finish %.3f ms, D2H %u recs %.3f ms, copy %.3f ms\n", t1 - t0, nrec,
                    t2 - t1, now_ms() - t2);
        return RD_OK;
    }
    order_and_dedupe(b->h_recs_pin, nrec, b->dc.S, b->order);
    const std::vector<uint32_t> &kept = b->order.kept;
    *n = (int)kept.size();
    if ((int)kept.size() > cap) return fail(RD_ERR_CAPACITY, "need room for %zu packets", kept.size());
    if (!kept.empty()) {
        if (!out) return fail(RD_ERR_ARG, "null out");
        for (size_t i = 0; i < kept.size(); i++) out[i] = b->h_recs_pin[kept[i]];  // straight into the caller's array
    }
    if (dbg_host())
        fprintf(stderr, "[rd] results: finish %.3f ms, D2H %u recs %.3f ms, order+dedupe+copy %.3f ms\n", t1 - t0, nrec,
                t2 - t1, now_ms() - t2);
    return RD_OK;
}

extern "C" int rd_batch_copy_bits(rd_batch *b, int stream, uint8_t *out, size_t nbytes) {
    if (!b || !out) return fail(RD_ERR_ARG, "null argument");
    if (stream < 0 || stream >= b->n_streams) return fail(RD_ERR_ARG, "stream out of range");
    const size_t need = (size_t)((b->n_samples + 7) / 8);
    if (nbytes < need) return fail(RD_ERR_ARG, "bit buffer too small: %zu < %zu", nbytes, need);
    int rc = batch_finish(b);
    if (rc) return rc;
    return copy_d2h(out, (const uint8_t *)(b->d_bits + (size_t)stream * b->bits_stride), need, b->stream);
}

extern "C" int rd_batch_copy_discriminated(rd_batch *b, int stream, size_t t0, double *out, size_t n) {
    if (!b || !out) return fail(RD_ERR_ARG, "null argument");
    if (stream < 0 || stream >= b->n_streams || t0 + n > (size_t)b->n_samples)
        return fail(RD_ERR_ARG, "range out of bounds");
    int rc = batch_alloc(b);
    if (rc) return rc;
    if (n == 0) return RD_OK;
    double *d = nullptr;
    HIPCHK(hipMalloc(&d, n * sizeof(double)));
    rd_launch_disc(batch_layout(b), stream, (long)t0, (long)n, d, b->stream);
    rc = copy_d2h(out, d, n * sizeof(double), b->stream);
    hipFree(d);
    return rc;
}

extern "C" int rd_batch_set_parse(rd_batch *b, int enabled) {
    if (!b) return fail(RD_ERR_ARG, "null batch");
    b->parse = enabled != 0;
    return RD_OK;
}

extern "C" int rd_batch_parsed(rd_batch *b, rd_parsed *out, int cap, int *n) {
    if (!b || !n) return fail(RD_ERR_ARG, "null argument");
    if (!b->parse) return fail(RD_ERR_STATE, "rd_batch_set_parse(b, 1) must precede rd_batch_run");
    int rc = batch_finish(b);
    if (rc) return rc;
    const uint32_t np = std::min(b->h_cnt[RD_CNT_PARSED], b->rec_cap);
    std::vector<rd_parsed> recs(np);
    if (np) {
        rc = copy_d2h(recs.data(), b->d_parsed, (size_t)np * sizeof(rd_parsed), b->stream);
        if (rc) return rc;
    }
    const int S = b->dc.S;  // the order parse() sees: packets of a call in _slice's order
    std::sort(recs.begin(), recs.end(), [S](const rd_parsed &x, const rd_parsed &y) {
        if (x.stream != y.stream) return x.stream < y.stream;
        if (x.call != y.call) return x.call < y.call;
        const int px = x.index % S, py = y.index % S;
        if (px != py) return px < py;
        return x.index < y.index;
    });
    // a call reports a byte string once (py:203-205, first occurrence in _slice's order): messages
    // with equal bytes come from packets with equal bytes (the sync word is the matched preamble)
    size_t kept = 0, group = 0;
    for (size_t i = 0; i < recs.size(); i++) {
        const rd_parsed &r = recs[i];
        if (kept == 0 || r.stream != recs[kept - 1].stream || r.call != recs[kept - 1].call) group = kept;
        bool dup = false;
        for (size_t k = group; k < kept && !dup; k++)
            dup = recs[k].nbytes == r.nbytes && memcmp(recs[k].data, r.data, sizeof r.data) == 0;
        if (!dup) recs[kept++] = r;
    }
    *n = (int)kept;
    if ((int)kept > cap) return fail(RD_ERR_CAPACITY, "need room for %zu messages", kept);
    if (kept) {
        if (!out) return fail(RD_ERR_ARG, "null out");
        memcpy(out, recs.data(), kept * sizeof(rd_parsed));
    }
    return RD_OK;
}

// 1: pipelined completion (see batch_adopt): for callers that keep several runs queued on one stream and want the
// stream never to idle; a run's results then become available one demod kernel later.  0 (default): every run ends
// with an event of its own.
extern "C" int rd_batch_set_pipelined(rd_batch *b, int enabled) {
    if (!b) return fail(RD_ERR_ARG, "null batch");
    b->pipelined = enabled != 0;
    return RD_OK;
}

extern "C" int rd_batch_set_timing(rd_batch *b, int enabled) {
    if (!b) return fail(RD_ERR_ARG, "null batch");
    b->timing = enabled != 0;
    b->timing_detail = enabled >= 2;  // 1: demod kernel and total only (3 events per run); 2: every stage
    b->ev_runs = 0;
    b->ev_end.clear();
    // events for the first 64 timed runs exist before the first of them is launched (creating them one run at a
    // time showed as a stall of several milliseconds inside a 60 ms timed region)
    if (b->timing && b->dev_ready && use_device(b->device) == RD_OK) {
        const size_t want = 5 * 64;
        const size_t old = b->evs.size();
        if (old < want) {
            b->evs.resize(want, nullptr);
            for (size_t i = old; i < want; i++) HIPCHK(hipEventCreate(&b->evs[i]));
        }
    }
    return RD_OK;
}

extern "C" int rd_batch_get_timing(rd_batch *b, rd_timing *out) {
    if (!b || !out) return fail(RD_ERR_ARG, "null argument");
    if (!b->timing) return fail(RD_ERR_STATE, "timing not enabled");
    int rc = batch_finish(b);
    if (rc) return rc;
    // mean over the runs recorded since the last call (events are only read here, after the
    // timed region: hipEventElapsedTime costs milliseconds on ROCm 7.2)
    rd_timing t = {};
    size_t n_end = 0;
    for (size_t r = 0; r < b->ev_runs; r++) {
        hipEvent_t *e = &b->evs[5 * r];
        float v[5] = {0, 0, 0, 0, 0};
        HIPCHK(hipEventElapsedTime(&v[0], e[0], e[1]));
        if (b->timing_detail) {
            HIPCHK(hipEventElapsedTime(&v[1], e[1], e[2]));
            HIPCHK(hipEventElapsedTime(&v[2], e[2], e[3]));
            HIPCHK(hipEventElapsedTime(&v[3], e[3], e[4]));
        }
        // (a pipelined run has no end-of-run event: total_ms is the mean over the runs that have one, 0 if none)
        const bool has_end = r >= b->ev_end.size() || b->ev_end[r];
        if (has_end) { HIPCHK(hipEventElapsedTime(&v[4], e[0], e[4])); n_end++; }
        t.demod_ms += v[0]; t.fixup_ms += v[1]; t.search_ms += v[2]; t.slice_ms += v[3]; t.total_ms += v[4];
    }
    if (b->ev_runs) {
        const float k = 1.0f / (float)b->ev_runs;
        t.demod_ms *= k; t.fixup_ms *= k; t.search_ms *= k; t.slice_ms *= k;
        t.total_ms = n_end ? t.total_ms / (float)n_end : 0.0f;
    }
    t.runs = (int32_t)b->ev_runs;
    b->ev_runs = 0;
    b->ev_end.clear();
    b->last_timing = t;
    *out = t;
    return RD_OK;
}

// Which forms the last run took (after it has completed): RD_FORM_* bits.
extern "C" int rd_batch_last_run_forms(rd_batch *b, uint32_t *forms) {
    if (!b || !forms) return fail(RD_ERR_ARG, "null argument");
    int rc = batch_finish(b);
    if (rc) return rc;
    *forms = (b->dev_order ? RD_FORM_ORDERED_TAIL | RD_FORM_ONE_LAUNCH_TAIL : 0u) | (b->second_pass ? RD_FORM_SECOND_PASS : 0u);
    return RD_OK;
}

extern "C" int rd_batch_get_counters(rd_batch *b, uint64_t *fixup_runs, uint64_t *matches) {
    if (!b) return fail(RD_ERR_ARG, "null batch");
    int rc = batch_finish(b);
    if (rc) return rc;
    if (fixup_runs) *fixup_runs = b->last_fix;
    if (matches) *matches = b->last_match;
    return RD_OK;
}

// ------------------------------------------------------------------------------------------
// streaming demodulator (py:128-253)
// ------------------------------------------------------------------------------------------
// One block in flight: pinned input, its device staging copy, pinned counters and mapped records.
// ------------------------------------------------------------------------------------------
// Host push (streaming handle).  A kernel's own reads of pinned host memory run at ~13 GB/s however many CUs ask
// (profiles/r04_stream_stamps.txt: a complex128 block's 128 KB = 10 of the kernel's 15 us), the host's stores into device
// memory through the PCIe BAR at ~45 GB/s (tools/ubench/bar_write.hip: 128 KB in 2.9 us).  Where the whole of device memory
// is host-visible (hipDeviceAttributeIsLargeBar) rd_demod_submit therefore copies the block straight into an UNCACHED
// device buffer (no stale line in an XCD's L2 when the slot is written again), makes the stores leave the CPU's
// write-combining buffers (sfence), flushes the GPU's host data path (HDP: the register ROCr publishes for exactly this,
// HSA_AMD_AGENT_INFO_HDP_FLUSH) and launches; PCIe keeps posted writes in order, so the block is in HBM before the launch's
// doorbell arrives.  Without a large BAR, without the register, or with RD_PUSH_INPUT=0: the pinned slot, as before.
// ------------------------------------------------------------------------------------------
#define RD_MAX_DEVICES 64
struct rd_push_info {
    int state = 0;                        // 0 not probed, 1 available, -1 not available
    volatile uint32_t *hdp_flush = nullptr;
};
static rd_push_info g_push[RD_MAX_DEVICES];

struct rd_hdp_find { uint32_t bdf, domain; volatile uint32_t *reg; bool found; };
static hsa_status_t rd_hdp_cb(hsa_agent_t agent, void *data) {
    rd_hdp_find *f = (rd_hdp_find *)data;
    hsa_device_type_t t;
    if (hsa_agent_get_info(agent, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS || t != HSA_DEVICE_TYPE_GPU) return HSA_STATUS_SUCCESS;
    uint32_t bdf = 0, dom = 0;
    if (hsa_agent_get_info(agent, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
    if (hsa_agent_get_info(agent, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &dom) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
    if (bdf != f->bdf || dom != f->domain) return HSA_STATUS_SUCCESS;
    hsa_amd_hdp_flush_t h = {nullptr, nullptr};
    if (hsa_agent_get_info(agent, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_HDP_FLUSH, &h) == HSA_STATUS_SUCCESS) f->reg = h.HDP_MEM_FLUSH_CNTL;
    f->found = true;
    return HSA_STATUS_INFO_BREAK;
}

// 1 when the host may write blocks straight into device memory of `dev` (and knows how to flush the HDP behind them)
static int g_push_mode = -1;   // rd_set_input_push: 0 = never
static bool push_available(int dev) {
    if (dev < 0 || dev >= RD_MAX_DEVICES || g_push_mode == 0) return false;
    static std::mutex mu;   // (handles may be created from several threads)
    std::lock_guard<std::mutex> lock(mu);
    rd_push_info &p = g_push[dev];
    if (p.state) return p.state > 0;
    p.state = -1;
    const char *e = getenv("RD_PUSH_INPUT");
    if (e && atoi(e) == 0) return false;
    int large = 0;
    if (hipDeviceGetAttribute(&large, hipDeviceAttributeIsLargeBar, dev) != hipSuccess || !large) return false;
    char id[64] = {0};
    unsigned dom = 0, bus = 0, d = 0, fn = 0;
    if (hipDeviceGetPCIBusId(id, sizeof id, dev) != hipSuccess || sscanf(id, "%x:%x:%x.%x", &dom, &bus, &d, &fn) != 4) return false;
    if (hsa_init() != HSA_STATUS_SUCCESS) return false;   // (a reference on the runtime HIP already runs on; kept)
    rd_hdp_find f = {(bus << 8) | (d << 3) | fn, dom, nullptr, false};
    hsa_iterate_agents(rd_hdp_cb, &f);
    if (!f.found || !f.reg) return false;
    p.hdp_flush = f.reg;
    p.state = 1;
    return true;
}

extern "C" int rd_set_input_push(int mode) {
    int rc = ensure_device();
    if (rc) return rc;
    g_push_mode = mode == 0 ? 0 : -1;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return fail(RD_ERR_DEVICE, "hipGetDevice failed");
    return push_available(dev) ? 1 : 0;
}

struct rd_slot {
    uint8_t *h_in = nullptr;         // pinned + mapped: NS x 2B bytes (or B complex128)
    uint8_t *d_in_map = nullptr;     // device address of h_in (the one-launch block reads its input from there)
    uint8_t *d_in = nullptr;         // device staging: the H2D copy lands here on the copy stream
    uint8_t *d_push = nullptr;       // uncached device memory the HOST writes the block into (large BAR), or null
    uint32_t *h_cnt = nullptr;       // pinned
    rd_packet *h_recs = nullptr;     // pinned + mapped, rec_cap entries: written by the slice kernel
    rd_packet *d_recs_map = nullptr; // device address of h_recs
    // one-launch block (k_stream_block): per stream, matches found and the sequence number that says "done"
    uint32_t *h_sb = nullptr;        // pinned + mapped: [NS] counts, [NS] flags
    uint32_t *d_sb_map = nullptr;
    uint32_t seq = 0;                // what the flags read once this slot's block is done
    bool one = false;                // this slot's block went through k_stream_block
    hipEvent_t e_in = nullptr, e_done = nullptr;
};

struct rd_demod {
    rd_config cfg;
    rd_devcfg dc;
    bool dev_ready = false;
    bool cplx_mode = false;  // switched on by the first complex128 block, until reset
    long seen = 0;           // blocks submitted since reset
    int NS = 1;              // independent streams fed in lock step (rd_create_multi)
    int device = -1;         // the device the buffers live on
    size_t ring_stride = 0;  // bytes between the streams' rings
    // per stream, byte ring: [hdr 32 B][prev 2B][cur 2B]; complex ring (NS == 1 only):
    // [hdr 16][prev B][cur B] complex128
    uint8_t *d_ring = nullptr;
    double *d_cring = nullptr;
    uint32_t *d_sync = nullptr;   // k_stream_block_cplx's arrival counter (zero between launches)
    uint32_t *d_blockbits = nullptr, *d_win[2] = {nullptr, nullptr}, *d_fix = nullptr, *d_cnt = nullptr;
    int cur_win = 0;
    rd_match *d_matches = nullptr;
    double *d_tmp = nullptr;  // 2*(B+1) doubles for the state mirrors
    double *h_tmp = nullptr;        // pinned mirror of d_tmp (state accessors)
    // Two slots: block i+1's host-to-device copy (copy stream) runs beside block i's kernels (compute
    // stream).  rd_demod_block = submit + fetch of one block; rd_demod_submit / rd_demod_fetch expose the
    // pipeline (runners/rtlsdr.py:100-103 -> worker.py:34-58 without the pickled-queue hop).
    rd_slot slot[2];
    int head = 0, nflight = 0;      // oldest block in flight, blocks in flight (0..2)
    hipStream_t st = nullptr, st_copy = nullptr;
    std::vector<rd_packet> last;    // ordered, deduplicated records of the last fetched block
    rd_order_scratch order;         // host ordering scratch
    uint32_t fix_cap = 0, match_cap = 0, rec_cap = 0;
    bool fast_ok = false;
    bool one_ok = false;            // the configuration is one k_stream_block is built for (and RD_STREAM_IMPL != legacy)
    uint32_t seq = 0;               // blocks sent through k_stream_block
    std::vector<rd_packet> gather;  // records of a one-launch block, streams one after the other
    // A producer's buffer registered with the device (rd_demod_register_input): rd_demod_submit_from launches on blocks
    // that already lie there - a shared-memory ring an SDR process fills - without copying them anywhere
    uint8_t *ext_host = nullptr, *ext_dev = nullptr;
    size_t ext_bytes = 0;
    int stale = 0;                  // blocks in flight whose fetch timed out: dropped (waited for, results discarded) by
                                    // the next submit / reset - the caller has already been told (worker.py:56-58)
};

extern "C" int rd_create_multi(const rd_config *cfg, int n_streams, rd_demod **out) {
    if (!out) return fail(RD_ERR_ARG, "null out");
    rd_devcfg dc;
    int rc = make_devcfg(cfg, &dc);
    if (rc) return rc;
    if (n_streams < 1 || (uint64_t)n_streams * ((uint64_t)dc.B + 1) > 0x7FFFFFFFull)
        return fail(RD_ERR_ARG, "n_streams out of range");
    rd_demod *h = new rd_demod();
    h->cfg = *cfg;
    h->dc = dc;
    h->NS = n_streams;
    h->ring_stride = ((32 + 4 * (size_t)dc.B) + 15) & ~(size_t)15;
    h->fast_ok = (dc.B % 8) == 0;
    *out = h;
    return RD_OK;
}

extern "C" int rd_create(const rd_config *cfg, rd_demod **out) { return rd_create_multi(cfg, 1, out); }

static size_t ring_cur_off(const rd_demod *h) { return 32 + 2 * (size_t)h->dc.B; }

static int demod_alloc(rd_demod *h) {
    if (h->dev_ready) return use_device(h->device);
    int rc = ensure_device();
    if (rc) return rc;
    HIPCHK(hipGetDevice(&h->device));
    const size_t B = (size_t)h->dc.B, L = (size_t)h->dc.L, NS = (size_t)h->NS;
    const size_t ring_bytes = NS * h->ring_stride + RD_INPUT_PAD;
    HIPCHK(hipMalloc(&h->d_ring, ring_bytes));
    HIPCHK(hipMemset(h->d_ring, 127, ring_bytes));
    h->fix_cap = (uint32_t)(NS * ((B + 31) / 32));  // every run of a block
    h->match_cap = (uint32_t)(NS * (B + 1));        // every position of every window: cannot overflow
    h->rec_cap = h->match_cap;
    HIPCHK(hipMalloc(&h->d_blockbits, NS * ((B + 31) / 32) * 4));
    for (int i = 0; i < 2; i++) {
        HIPCHK(hipMalloc(&h->d_win[i], NS * ((L + 31) / 32) * 4));
        HIPCHK(hipMemset(h->d_win[i], 0, NS * ((L + 31) / 32) * 4));
    }
    HIPCHK(hipMalloc(&h->d_fix, (size_t)h->fix_cap * 4));
    HIPCHK(hipMalloc(&h->d_cnt, RD_CNT_TOTAL * 4));
    HIPCHK(hipMalloc(&h->d_matches, (size_t)h->match_cap * sizeof(rd_match)));
    HIPCHK(hipMalloc(&h->d_tmp, 2 * (2 * B + 2) * sizeof(double)));
    HIPCHK(hipHostMalloc((void **)&h->h_tmp, 2 * (2 * B + 2) * sizeof(double), hipHostMallocDefault));
    const size_t in_bytes = std::max(16 * B, NS * 2 * B);
    for (int i = 0; i < 2; i++) {
        rd_slot &sl = h->slot[i];
        HIPCHK(hipHostMalloc((void **)&sl.h_in, in_bytes, hipHostMallocMapped));
        HIPCHK(hipHostGetDevicePointer((void **)&sl.d_in_map, sl.h_in, 0));
        HIPCHK(hipMalloc(&sl.d_in, in_bytes));
        if (push_available(h->device) && hipExtMallocWithFlags((void **)&sl.d_push, in_bytes, hipDeviceMallocUncached) != hipSuccess) {
            (void)hipGetLastError();
            sl.d_push = nullptr;   // (the pinned slot serves)
        }
        HIPCHK(hipHostMalloc((void **)&sl.h_sb, 2 * NS * sizeof(uint32_t), hipHostMallocMapped));
        HIPCHK(hipHostGetDevicePointer((void **)&sl.d_sb_map, sl.h_sb, 0));
        memset(sl.h_sb, 0, 2 * NS * sizeof(uint32_t));
        HIPCHK(hipHostMalloc((void **)&sl.h_cnt, RD_CNT_SLOTS * 4, hipHostMallocDefault));
        HIPCHK(hipHostMalloc((void **)&sl.h_recs, (size_t)h->rec_cap * sizeof(rd_packet), hipHostMallocMapped));
        HIPCHK(hipHostGetDevicePointer((void **)&sl.d_recs_map, sl.h_recs, 0));
        HIPCHK(hipEventCreateWithFlags(&sl.e_in, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&sl.e_done, hipEventDisableTiming));
    }
    HIPCHK(hipStreamCreateWithFlags(&h->st, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&h->st_copy, hipStreamNonBlocking));
    // Nothing is allocated inside a demodulate() call: the complex ring of a single-stream handle exists from the start
    // (pyrtlsdr feeds complex blocks), and the state mirrors' float64 kernels run once here - the first launch of a
    // kernel that needs scratch memory costs milliseconds, which `.discriminated` (read by protocol.py:307-309 for the
    // first CRC-valid packet) used to pay inside a 30 ms block budget.
    if (NS == 1) {
        HIPCHK(hipMalloc(&h->d_cring, 2 * (16 + 2 * B) * sizeof(double)));
        HIPCHK(hipMemset(h->d_cring, 0, 2 * (16 + 2 * B) * sizeof(double)));
        HIPCHK(hipMalloc(&h->d_sync, 64));
        HIPCHK(hipMemset(h->d_sync, 0, 64));
    }
    HIPCHK(hipDeviceSynchronize());  // the memsets above ran on the null stream
    {
        rd_layout wl;
        wl.iq = h->d_ring + 32 + 2 * B; wl.stream_stride = h->ring_stride; wl.n_streams = (int)NS; wl.n_samples = (uint32_t)B;
        wl.hist_mode = 0; wl.valid_from = 0; wl.bits = h->d_blockbits; wl.bits_stride = (B + 31) / 32;
        rd_launch_disc(wl, 0, 0, 8, h->d_tmp, h->st);
        rd_launch_filtered(wl, 0, 0, 8, h->d_tmp, h->st);
        if (h->d_cring) {
            rd_cplx_layout cl;
            cl.x = h->d_cring + 2 * (16 + B); cl.valid_from = 0; cl.n = (long)B;
            rd_launch_cplx_disc(cl, 0, 8, h->d_tmp, h->st);
            rd_launch_cplx_filtered(cl, 0, 8, h->d_tmp, h->st);
        }
        HIPCHK(hipMemcpyAsync(h->h_tmp, h->d_tmp, 64, hipMemcpyDeviceToHost, h->st));
        HIPCHK(hipStreamSynchronize(h->st));
    }
    {   // one launch per block where the kernel exists for the configuration; RD_STREAM_IMPL=legacy: the multi-launch form (A/B)
        const char *e = getenv("RD_STREAM_IMPL");
        const rd_devcfg &c = h->dc;
        h->one_ok = !(e && e[0] == 'l') && h->fast_ok && c.S == 14 && c.P == 16 && c.K == 80 && c.pre_mask == 0x91D3ull &&
                    c.L == 2 * c.B && c.B % 32 == 0 && c.B >= 2048 && c.B <= 16384;
    }
    h->dev_ready = true;
    return RD_OK;
}

extern "C" void rd_destroy(rd_demod *h) {
    if (!h) return;
    if (h->dev_ready && g_hip_pid == getpid()) {
        if (h->device >= 0) hipSetDevice(h->device);
        if (h->st) hipStreamSynchronize(h->st);
        if (h->st_copy) hipStreamSynchronize(h->st_copy);
        if (h->ext_host) hipHostUnregister(h->ext_host);
        hipFree(h->d_ring); hipFree(h->d_cring); hipFree(h->d_sync); hipFree(h->d_blockbits);
        hipFree(h->d_win[0]); hipFree(h->d_win[1]); hipFree(h->d_fix); hipFree(h->d_cnt);
        hipFree(h->d_matches); hipFree(h->d_tmp);
        hipHostFree(h->h_tmp);
        for (int i = 0; i < 2; i++) {
            rd_slot &sl = h->slot[i];
            hipHostFree(sl.h_in); hipFree(sl.d_in); hipFree(sl.d_push); hipHostFree(sl.h_cnt); hipHostFree(sl.h_recs); hipHostFree(sl.h_sb);
            if (sl.e_in) hipEventDestroy(sl.e_in);
            if (sl.e_done) hipEventDestroy(sl.e_done);
        }
        if (h->st) hipStreamDestroy(h->st);
        if (h->st_copy) hipStreamDestroy(h->st_copy);
    }
    delete h;
}

// Wait (polling, with the deadline of every host wait) until the block in `sl` has completed.
static int demod_wait_slot(rd_demod *h, rd_slot &sl) {
    if (!sl.one) return wait_event(sl.e_done, "the block's kernels");
    // one-launch block: the streams' sequence numbers in pinned memory (an event query costs microseconds; a kernel
    // that died would never write them, so the stream is asked now and then)
    const size_t NS = (size_t)h->NS;
    volatile uint32_t *flag = sl.h_sb + NS;
    rd_waiter w(rd_wait_timeout_ms());
    uint32_t spins = 0;
    for (size_t s = 0; s < NS; s++) {
        while (flag[s] != sl.seq) {
            if (!w.relax())
                return fail(RD_ERR_DEVICE, "timed out after %.0f ms waiting for the block's kernel (stream %zu of %zu)",
                            w.waited_ms(), s, NS);
            if ((++spins & 0xFFF) == 0xFFF || w.polls >= 1024) {  // (>= 1024 polls: past the 50 us of spinning)
                const hipError_t e = hipStreamQuery(h->st);
                if (e != hipSuccess && e != hipErrorNotReady)
                    return fail(RD_ERR_DEVICE, "hipStreamQuery: %s", hipGetErrorString(e));
                if (e == hipSuccess && flag[s] != sl.seq)
                    return fail(RD_ERR_DEVICE, "the block's kernel finished without reporting stream %zu", s);
            }
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    return RD_OK;
}

// Blocks whose fetch timed out are still in flight: wait for them (deadline again) and discard what they found - the
// streams' state (ring, window) has advanced by them as the reference's would have, only their packets are lost.
static int demod_drop_stale(rd_demod *h) {
    while (h->stale > 0 && h->nflight > 0) {
        int rc = demod_wait_slot(h, h->slot[h->head]);
        if (rc) return rc;
        h->head ^= 1;
        h->nflight--;
        h->stale--;
    }
    h->stale = 0;
    return RD_OK;
}

// every block in flight has been fetched (state accessors and reset need a quiet handle)
static int demod_quiet(rd_demod *h) {
    if (h->stale) {
        int rc = demod_drop_stale(h);
        if (rc) return rc;
    }
    if (h->nflight) return fail(RD_ERR_STATE, "%d block(s) in flight: rd_demod_fetch them first", h->nflight);
    return RD_OK;
}

extern "C" int rd_reset(rd_demod *h) {
    if (!h) return fail(RD_ERR_ARG, "null handle");
    if (h->dev_ready) {
        // blocks still in flight are dropped: wait for them, then clear
        // (on a timeout the handle is left as it was: reset() may be called again once the device has caught up)
        for (int i = 0; i < h->nflight; i++) {
            int rc = demod_wait_slot(h, h->slot[(h->head + i) & 1]);
            if (rc) return rc;
        }
        const size_t L = (size_t)h->dc.L;
        for (int i = 0; i < 2; i++)
            HIPCHK(hipMemsetAsync(h->d_win[i], 0, (size_t)h->NS * ((L + 31) / 32) * 4, h->st));
        if (h->d_sync) HIPCHK(hipMemsetAsync(h->d_sync, 0, 64, h->st));   // (a launch that never finished may have left a count)
        int rc = wait_stream(h->st, "reset()'s clears");
        if (rc) return rc;
    }
    h->stale = 0;
    h->nflight = 0;
    h->head = 0;
    h->seen = 0;
    h->cplx_mode = false;
    h->last.clear();
    return RD_OK;
}

// first readable sample relative to the newest block's sample 0
static long demod_valid_from(const rd_demod *h, long seen_before) {
    const long B = h->dc.B;
    if (seen_before <= 0) return 0;
    if (seen_before == 1) return -B;
    return -(B + 16);
}

static rd_layout demod_layout(const rd_demod *h, long seen_before) {
    rd_layout l;
    l.iq = h->d_ring + ring_cur_off(h);
    l.stream_stride = h->ring_stride;
    l.n_streams = h->NS;
    l.n_samples = (uint32_t)h->dc.B;
    l.hist_mode = seen_before > 0 ? 1 : 0;
    l.valid_from = demod_valid_from(h, seen_before);
    l.bits = h->d_blockbits;
    l.bits_stride = (size_t)((h->dc.B + 31) / 32);
    return l;
}

static rd_cplx_layout demod_clayout(const rd_demod *h, long seen_before) {
    rd_cplx_layout l;
    l.x = h->d_cring + 2 * (16 + (size_t)h->dc.B);
    l.valid_from = demod_valid_from(h, seen_before);
    l.n = h->dc.B;
    return l;
}

// Switch to the complex ring: LUT-convert the byte ring (py:26,38-39 - what the reference
// keeps in raw_samples) so history carries over exactly.
static int demod_enter_cplx(rd_demod *h) {
    const size_t B = (size_t)h->dc.B;
    if (!h->d_cring) HIPCHK(hipMalloc(&h->d_cring, 2 * (16 + 2 * B) * sizeof(double)));  // (single-stream handles have it from demod_alloc)
    rd_launch_lut(h->d_ring, h->d_cring, 16 + 2 * B, h->st);
    HIPCHK(hipGetLastError());
    h->cplx_mode = true;
    return RD_OK;
}

// Queue one block per stream (uint8: NS x 2B bytes, stream-major; complex128: single stream only): the
// copy to the device on the copy stream, everything else behind it on the compute stream.  Returns at
// once; at most two blocks may be in flight.
// `ext_off` >= 0: the block lies at that offset of the registered producer buffer (no copy: the kernels read it there)
static int demod_submit(rd_demod *h, const void *samples, int is_complex, long ext_off = -1) {
    const size_t B = (size_t)h->dc.B, L = (size_t)h->dc.L, NS = (size_t)h->NS;
    const size_t bw = (B + 31) / 32, lw = (L + 31) / 32;
    int rc = demod_alloc(h);
    if (rc) return rc;
    if (h->stale && (rc = demod_drop_stale(h))) return rc;
    if (h->nflight >= 2) return fail(RD_ERR_STATE, "two blocks in flight: rd_demod_fetch one first");
    rd_slot &sl = h->slot[(h->head + h->nflight) & 1];
    hipStream_t st = h->st;
    const size_t nbytes = is_complex ? 2 * B * sizeof(double) : NS * 2 * B;
    const uint8_t *in_host = sl.h_in, *in_dev = sl.d_in_map;   // where this block's bytes are, host and device address
    const bool will_one = h->one_ok && (is_complex || h->cplx_mode ? (NS == 1 && h->dc.B <= 8192) : true);
    bool pushed = false;
    if (ext_off >= 0) {
        in_host = h->ext_host + ext_off;
        in_dev = h->ext_dev + ext_off;
    } else if (will_one && sl.d_push) {
        // the host writes the block into device memory (see rd_push_info): stores out of the write-combining buffers,
        // then the HDP flush, then - in posted order behind both - the launch's doorbell
        memcpy(sl.d_push, samples, nbytes);
        _mm_sfence();
        *g_push[h->device].hdp_flush = 1u;
        in_dev = sl.d_push;
        pushed = true;
    } else {
        memcpy(sl.h_in, samples, nbytes);
    }
    sl.one = false;
    if (h->one_ok && NS == 1 && h->dc.B <= 8192 && (is_complex || h->cplx_mode)) {
        // ONE launch for the complex-input branch as well (py:144-150: what pyrtlsdr's stream() feeds the live receiver,
        // /root/reference/src/rtldavis/runners/rtlsdr.py:100-103); a uint8 block on a handle in complex mode goes through
        // the LUT inside the kernel.  The first complex block converts the byte ring once (py:26,38-39: history carries over)
        if (!h->cplx_mode) {
            rc = demod_enter_cplx(h);
            if (rc) return rc;
        }
        rd_sbc_args a;
        a.cfg = h->dc;
        a.ring = h->d_cring;
        a.in = in_dev;
        a.in_is_u8 = is_complex ? 0 : 1;
        const int nw = h->cur_win ^ 1;
        a.win_in = h->d_win[h->cur_win];
        a.win_out = h->d_win[nw];
        a.recs_host = sl.d_recs_map;
        a.cnt_host = sl.d_sb_map;
        a.flag_host = sl.d_sb_map + NS;
        a.seq = ++h->seq;
        a.seen_before = h->seen;
        a.sync = h->d_sync;
        if (rd_launch_stream_block_cplx(a, st)) {
            HIPCHK(hipGetLastError());
            h->cur_win = nw;
            sl.seq = a.seq;
            sl.one = true;
            h->seen++;
            h->nflight++;
            return RD_OK;
        }
        --h->seq;
    }
    if (h->one_ok && !is_complex && !h->cplx_mode) {
        // ONE launch: the kernel takes the block from the pinned buffer, rolls the ring, decides the bits exactly,
        // searches, slices and leaves records, counts and a per-stream sequence number in mapped host memory
        rd_sb_args a;
        a.cfg = h->dc;
        a.ring = h->d_ring;
        a.ring_stride = h->ring_stride;
        a.in = in_dev;
        const int nw = h->cur_win ^ 1;
        a.win_in = h->d_win[h->cur_win];
        a.win_out = h->d_win[nw];
        a.recs_host = sl.d_recs_map;
        a.cnt_host = sl.d_sb_map;
        a.flag_host = sl.d_sb_map + NS;
        a.seq = ++h->seq;
        a.seen_before = h->seen;
        if (rd_launch_stream_block(a, h->NS, st)) {
            HIPCHK(hipGetLastError());
            h->cur_win = nw;
            sl.seq = a.seq;
            sl.one = true;
            h->seen++;
            h->nflight++;
            return RD_OK;
        }
        --h->seq;
    }
    if (pushed) memcpy(sl.h_in, samples, nbytes);   // (the one-launch form declined: the multi-launch form copies from the pinned slot)
    HIPCHK(hipMemcpyAsync(sl.d_in, in_host, nbytes, hipMemcpyHostToDevice, h->st_copy));
    HIPCHK(hipEventRecord(sl.e_in, h->st_copy));
    HIPCHK(hipStreamWaitEvent(st, sl.e_in, 0));
    if (is_complex && !h->cplx_mode) {
        rc = demod_enter_cplx(h);
        if (rc) return rc;
    }
    const long seen_before = h->seen;
    // roll the raw rings left by one block (py:140,154): hdr <- tail of prev, prev <- cur, cur <- the new block
    if (!h->cplx_mode) {
        uint8_t *r = h->d_ring;
        const size_t rs = h->ring_stride;
        if (seen_before > 0) {
            HIPCHK(hipMemcpy2DAsync(r, rs, r + 2 * B, rs, 32, NS, hipMemcpyDeviceToDevice, st));
            HIPCHK(hipMemcpy2DAsync(r + 32, rs, r + 32 + 2 * B, rs, 2 * B, NS, hipMemcpyDeviceToDevice, st));
        }
        HIPCHK(hipMemcpy2DAsync(r + 32 + 2 * B, rs, sl.d_in, 2 * B, 2 * B, NS, hipMemcpyDeviceToDevice, st));
    } else {
        double *r = h->d_cring;
        if (seen_before > 0) {
            HIPCHK(hipMemcpyAsync(r, r + 2 * B, 32 * sizeof(double), hipMemcpyDeviceToDevice, st));
            HIPCHK(hipMemcpyAsync(r + 32, r + 32 + 2 * B, 2 * B * sizeof(double), hipMemcpyDeviceToDevice, st));
        }
        if (is_complex) {
            HIPCHK(hipMemcpyAsync(r + 32 + 2 * B, sl.d_in, 2 * B * sizeof(double), hipMemcpyDeviceToDevice, st));
        } else {
            rd_launch_lut(sl.d_in, r + 32 + 2 * B, B, st);
        }
    }
    HIPCHK(hipMemsetAsync(h->d_cnt, 0, RD_CNT_TOTAL * 4, st));
    if (!h->cplx_mode) {
        const rd_layout lay = demod_layout(h, seen_before);
        if (h->fast_ok) rd_launch_demod(lay, h->d_fix, h->fix_cap, h->d_cnt, st);
        rd_launch_fixup(lay, h->d_fix, h->fix_cap, h->d_cnt, h->fast_ok ? 0 : 1, nullptr, st);
    } else {
        rd_launch_cplx_bits(demod_clayout(h, seen_before), h->d_blockbits, st);
    }
    // quantized <- roll(quantized, -B) with the new bits at the end (py:157,163-166)
    const int nw = h->cur_win ^ 1;
    rd_launch_window_update(h->d_win[nw], h->d_win[h->cur_win], (long)L, h->d_blockbits, (long)B, h->NS, lw, bw, st);
    h->cur_win = nw;
    // whole-buffer search, keep q <= B (py:171-188,194)
    rd_launch_search(h->d_win[nw], lw, h->NS, (long)L, 0, (long)B, h->dc, h->d_matches, h->match_cap, h->d_cnt, st);
    if (!h->cplx_mode)
        rd_launch_slice(demod_layout(h, seen_before), h->d_win[nw], lw, (long)L, h->dc, h->d_matches, h->match_cap, 0,
                        0, (int)seen_before, nullptr, sl.d_recs_map, h->d_cnt, st);
    else
        rd_launch_cplx_slice(demod_clayout(h, seen_before), h->d_win[nw], (long)L, h->dc, h->d_matches, h->match_cap,
                             (int)seen_before, nullptr, sl.d_recs_map, h->d_cnt, st);
    HIPCHK(hipGetLastError());
    // The counters come back with the block; the few records of a block are written by the slice
    // kernel straight into pinned host memory (no copy to wait for; for the thousands of records of
    // a batch run the same was measured slower than a device buffer + one copy).
    HIPCHK(hipMemcpyAsync(sl.h_cnt, h->d_cnt, RD_CNT_SLOTS * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipEventRecord(sl.e_done, st));
    h->seen = seen_before + 1;
    h->nflight++;
    return RD_OK;
}

static int demod_give(rd_demod *h, rd_packet *out, int cap, int *n) {
    *n = (int)h->last.size();
    if ((int)h->last.size() > cap)
        return fail(RD_ERR_CAPACITY, "need room for %zu packets (rd_demod_refetch returns them)", h->last.size());
    if (!h->last.empty()) {
        if (!out) return fail(RD_ERR_ARG, "null out");
        memcpy(out, h->last.data(), h->last.size() * sizeof(rd_packet));
    }
    return RD_OK;
}

// Wait (polling) for the oldest block in flight and return its packets in the reference's order.
static int demod_fetch(rd_demod *h, rd_packet *out, int cap, int *n) {
    if (h->nflight == 0) return fail(RD_ERR_STATE, "no block in flight");
    rd_slot &sl = h->slot[h->head];
    if (h->stale) return fail(RD_ERR_STATE, "the blocks in flight were given up on (their fetch timed out): submit the next block or reset()");
    {
        int rc = demod_wait_slot(h, sl);
        if (rc) {  // every block in flight is dropped: the next submit / reset waits for them and discards their packets
            if (rc == RD_ERR_DEVICE) h->stale = h->nflight;
            return rc;
        }
    }
    if (sl.one) {
        const size_t NS = (size_t)h->NS, per = (size_t)h->dc.B + 1;
        h->head ^= 1;
        h->nflight--;
        h->gather.clear();
        for (size_t s = 0; s < NS; s++) {
            const uint32_t c = std::min<uint32_t>(sl.h_sb[s], (uint32_t)per);
            for (uint32_t i = 0; i < c; i++) h->gather.push_back(sl.h_recs[s * per + i]);
        }
        order_and_dedupe(h->gather.data(), h->gather.size(), h->dc.S, h->order);
        h->last.clear();
        for (uint32_t k : h->order.kept) h->last.push_back(h->gather[k]);
        return demod_give(h, out, cap, n);
    }
    h->head ^= 1;
    h->nflight--;
    const uint32_t nrec = std::min(sl.h_cnt[RD_CNT_MATCH], h->match_cap);  // one record per match (no call overlap here)
    order_and_dedupe(sl.h_recs, nrec, h->dc.S, h->order);
    h->last.clear();
    for (uint32_t k : h->order.kept) h->last.push_back(sl.h_recs[k]);
    return demod_give(h, out, cap, n);
}

extern "C" int rd_demod_submit(rd_demod *h, const void *samples, size_t count, int is_complex) {
    if (!h || !samples) return fail(RD_ERR_ARG, "null argument");
    const size_t B = (size_t)h->dc.B, NS = (size_t)h->NS;
    // py:32-36 / py:145-149
    if (is_complex) {
        if (NS != 1) return fail(RD_ERR_STATE, "complex input needs a single-stream handle");
        if (count != B) return fail(RD_ERR_ARG, "Incompatible array sizes: got %zu, expected %zu", count, B);
    } else {
        if (count != NS * 2 * B) return fail(RD_ERR_ARG, "Incompatible array sizes: got %zu, expected %zu", count, NS * 2 * B);
        if (h->cplx_mode && NS != 1) return fail(RD_ERR_STATE, "handle is in complex mode: reset() first");
    }
    return demod_submit(h, samples, is_complex);
}

// Zero-copy input (the pinned ring that replaces the pickled-ndarray queue hop,
// /root/reference/src/rtldavis/runners/rtlsdr.py:100-103 -> /root/reference/src/rtldavis/worker.py:37).
extern "C" int rd_demod_register_input(rd_demod *h, void *host, size_t nbytes) {
    if (!h) return fail(RD_ERR_ARG, "null handle");
    int rc = demod_alloc(h);
    if (rc) return rc;
    if (h->nflight) return fail(RD_ERR_STATE, "%d block(s) in flight: fetch them before the input buffer changes", h->nflight);
    if (h->ext_host) {
        HIPCHK(hipHostUnregister(h->ext_host));
        h->ext_host = h->ext_dev = nullptr;
        h->ext_bytes = 0;
    }
    if (!host || nbytes == 0) return RD_OK;  // (unregister only)
    if ((uintptr_t)host % 16) return fail(RD_ERR_ARG, "the input buffer must be 16-byte aligned");
    HIPCHK(hipHostRegister(host, nbytes, hipHostRegisterMapped | hipHostRegisterPortable));
    void *dev = nullptr;
    const hipError_t e = hipHostGetDevicePointer(&dev, host, 0);
    if (e != hipSuccess) {
        hipHostUnregister(host);
        return fail(RD_ERR_DEVICE, "hipHostGetDevicePointer: %s", hipGetErrorString(e));
    }
    h->ext_host = (uint8_t *)host;
    h->ext_dev = (uint8_t *)dev;
    h->ext_bytes = nbytes;
    return RD_OK;
}

extern "C" int rd_demod_input_mode(rd_demod *h) {
    if (!h) return fail(RD_ERR_ARG, "null handle");
    int rc = demod_alloc(h);
    if (rc) return rc;
    return h->slot[0].d_push && h->slot[1].d_push ? 1 : 0;
}

extern "C" int rd_demod_submit_from(rd_demod *h, size_t offset, size_t count, int is_complex) {
    if (!h) return fail(RD_ERR_ARG, "null handle");
    if (!h->ext_host) return fail(RD_ERR_STATE, "no input buffer registered (rd_demod_register_input)");
    const size_t B = (size_t)h->dc.B, NS = (size_t)h->NS;
    size_t want = 0;
    if (is_complex && NS != 1) return fail(RD_ERR_STATE, "complex input needs a single-stream handle");
    if (rd_check_block_count(is_complex, count, B, NS, &want) != RD_OK)
        return fail(RD_ERR_ARG, "Incompatible array sizes: got %zu, expected %zu", count, want);  // py:32-36 / py:145-149
    if (!is_complex && h->cplx_mode && NS != 1) return fail(RD_ERR_STATE, "handle is in complex mode: reset() first");
    const size_t nbytes = is_complex ? 2 * B * sizeof(double) : NS * 2 * B;
    if (offset % 16 || offset > h->ext_bytes || nbytes > h->ext_bytes - offset)
        return fail(RD_ERR_ARG, "block at offset %zu (%zu bytes) does not lie 16-byte aligned inside the registered buffer (%zu bytes)",
                    offset, nbytes, h->ext_bytes);
    return demod_submit(h, nullptr, is_complex, (long)offset);
}

extern "C" int rd_demod_fetch(rd_demod *h, rd_packet *out, int cap, int *n) {
    if (!h || !n) return fail(RD_ERR_ARG, "null argument");
    return demod_fetch(h, out, cap, n);
}

extern "C" int rd_demod_refetch(rd_demod *h, rd_packet *out, int cap, int *n) {
    if (!h || !n) return fail(RD_ERR_ARG, "null argument");
    return demod_give(h, out, cap, n);
}

extern "C" int rd_demod_inflight(rd_demod *h) { return h ? h->nflight : 0; }

static int demod_blocks(rd_demod *h, const void *samples, int is_complex, rd_packet *out, int cap, int *n) {
    int rc = demod_quiet(h);
    if (rc) return rc;
    rc = demod_submit(h, samples, is_complex);
    if (rc) return rc;
    return demod_fetch(h, out, cap, n);
}

extern "C" int rd_demod_block(rd_demod *h, const void *samples, size_t count, int is_complex, rd_packet *out, int cap,
                              int *n) {
    if (!h || !samples || !n) return fail(RD_ERR_ARG, "null argument");
    if (h->NS != 1) return fail(RD_ERR_STATE, "handle holds %d streams: use rd_demod_blocks", h->NS);
    const size_t B = (size_t)h->dc.B;
    // py:32-36 / py:145-149
    if ((is_complex && count != B) || (!is_complex && count != 2 * B))
        return fail(RD_ERR_ARG, "Incompatible array sizes: got %zu, expected %zu", count, is_complex ? B : 2 * B);
    return demod_blocks(h, samples, is_complex, out, cap, n);
}

extern "C" int rd_demod_blocks(rd_demod *h, const uint8_t *iq, size_t nbytes, rd_packet *out, int cap, int *n) {
    if (!h || !iq || !n) return fail(RD_ERR_ARG, "null argument");
    const size_t want = (size_t)h->NS * 2 * (size_t)h->dc.B;
    if (nbytes != want) return fail(RD_ERR_ARG, "Incompatible array sizes: got %zu bytes, expected %zu", nbytes, want);
    if (h->cplx_mode) return fail(RD_ERR_STATE, "handle is in complex mode: reset() first");
    return demod_blocks(h, iq, 0, out, cap, n);
}

extern "C" int rd_copy_discriminated_stream(rd_demod *h, int stream, double *out, size_t n);
extern "C" int rd_copy_discriminated(rd_demod *h, double *out, size_t n) {
    return rd_copy_discriminated_stream(h, 0, out, n);
}

extern "C" int rd_copy_discriminated_stream(rd_demod *h, int stream, double *out, size_t n) {
    if (!h || !out) return fail(RD_ERR_ARG, "null argument");
    if (stream < 0 || stream >= h->NS) return fail(RD_ERR_ARG, "stream out of range");
    const size_t B = (size_t)h->dc.B;
    if (n != 2 * B) return fail(RD_ERR_ARG, "discriminated has %zu elements", 2 * B);
    if (h->seen == 0) {  // py:134 zeros
        memset(out, 0, n * sizeof(double));
        return RD_OK;
    }
    int rc = demod_quiet(h);
    if (rc) return rc;
    // discriminated = d over [-B, B) relative to the newest block (py:156,162)
    if (!h->cplx_mode) rd_launch_disc(demod_layout(h, h->seen - 1), stream, -(long)B, 2 * (long)B, h->d_tmp, h->st);
    else rd_launch_cplx_disc(demod_clayout(h, h->seen - 1), -(long)B, 2 * (long)B, h->d_tmp, h->st);
    HIPCHK(hipGetLastError());
    rc = copy_d2h(h->h_tmp, h->d_tmp, n * sizeof(double), h->st);  // pinned: no staging, polling wait
    if (rc) return rc;
    memcpy(out, h->h_tmp, n * sizeof(double));
    return RD_OK;
}

extern "C" int rd_copy_filtered(rd_demod *h, double *out_interleaved, size_t n_complex) {
    if (!h || !out_interleaved) return fail(RD_ERR_ARG, "null argument");
    const size_t B = (size_t)h->dc.B;
    if (n_complex != B + 1) return fail(RD_ERR_ARG, "filtered has %zu elements", B + 1);
    if (h->seen == 0) {
        memset(out_interleaved, 0, 2 * n_complex * sizeof(double));
        return RD_OK;
    }
    int rc = demod_quiet(h);
    if (rc) return rc;
    // filtered[j] = f[j-1] relative to the newest block (py:155,161)
    if (!h->cplx_mode) rd_launch_filtered(demod_layout(h, h->seen - 1), 0, -1, (long)B + 1, h->d_tmp, h->st);
    else rd_launch_cplx_filtered(demod_clayout(h, h->seen - 1), -1, (long)B + 1, h->d_tmp, h->st);
    HIPCHK(hipGetLastError());
    rc = copy_d2h(h->h_tmp, h->d_tmp, 2 * n_complex * sizeof(double), h->st);
    if (rc) return rc;
    memcpy(out_interleaved, h->h_tmp, 2 * n_complex * sizeof(double));
    return RD_OK;
}

extern "C" int rd_copy_quantized(rd_demod *h, uint8_t *out, size_t n) {
    if (!h || !out) return fail(RD_ERR_ARG, "null argument");
    const size_t L = (size_t)h->dc.L;
    if (n != L) return fail(RD_ERR_ARG, "quantized has %zu elements", L);
    if (!h->dev_ready) {  // py:135 zeros
        memset(out, 0, n);
        return RD_OK;
    }
    std::vector<uint32_t> words((L + 31) / 32);
    if (words.size() * 4 > 2 * (2 * (size_t)h->dc.B + 2) * sizeof(double)) return fail(RD_ERR_ARG, "window too large");
    int rc = demod_quiet(h);
    if (rc) return rc;
    rc = copy_d2h(h->h_tmp, h->d_win[h->cur_win], words.size() * 4, h->st);
    if (rc) return rc;
    memcpy(words.data(), h->h_tmp, words.size() * 4);
    for (size_t t = 0; t < L; t++) out[t] = (uint8_t)((words[t >> 5] >> (t & 31)) & 1u);  // unpack only
    return RD_OK;
}

// ------------------------------------------------------------------------------------------
// stage functions on host arrays
// ------------------------------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) hipFree(p); }
};

#define STAGE_PROLOGUE()            \
    do {                            \
        int rc_ = ensure_device();  \
        if (rc_) return rc_;        \
    } while (0)

extern "C" int rd_lut_execute(const uint8_t *in_bytes, size_t n_bytes, double *out_cplx, size_t n_cplx) {
    if (n_bytes != 2 * n_cplx) return fail(RD_ERR_ARG, "Incompatible array sizes: in_bytes.size=%zu, out_cplx.size=%zu", n_bytes, n_cplx);
    if (n_cplx == 0) return RD_OK;
    if (!in_bytes || !out_cplx) return fail(RD_ERR_ARG, "null argument");
    STAGE_PROLOGUE();
    DevBuf a, b;
    HIPCHK(hipMalloc(&a.p, n_bytes));
    HIPCHK(hipMalloc(&b.p, n_cplx * 16));
    HIPCHK(hipMemcpy(a.p, in_bytes, n_bytes, hipMemcpyHostToDevice));
    rd_launch_lut((const uint8_t *)a.p, (double *)b.p, n_cplx, nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(out_cplx, b.p, n_cplx * 16, hipMemcpyDeviceToHost));
    return RD_OK;
}

extern "C" int rd_rotate_fs4(const double *in_cplx, double *out_cplx, size_t n_cplx) {
    if (n_cplx == 0) return RD_OK;
    if (!in_cplx || !out_cplx) return fail(RD_ERR_ARG, "null argument");
    STAGE_PROLOGUE();
    DevBuf a, b;
    HIPCHK(hipMalloc(&a.p, n_cplx * 16));
    HIPCHK(hipMalloc(&b.p, n_cplx * 16));
    HIPCHK(hipMemcpy(a.p, in_cplx, n_cplx * 16, hipMemcpyHostToDevice));
    rd_launch_rotate((const double *)a.p, (double *)b.p, n_cplx, nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(out_cplx, b.p, n_cplx * 16, hipMemcpyDeviceToHost));
    return RD_OK;
}

extern "C" int rd_fir9(const double *in_cplx, size_t n_in, double *out_cplx, size_t n_out) {
    if (n_in < 9 || n_out > n_in - 8) return fail(RD_ERR_ARG, "fir9: n_out must be <= n_in - 8");
    if (n_out == 0) return RD_OK;
    if (!in_cplx || !out_cplx) return fail(RD_ERR_ARG, "null argument");
    STAGE_PROLOGUE();
    DevBuf a, b;
    HIPCHK(hipMalloc(&a.p, n_in * 16));
    HIPCHK(hipMalloc(&b.p, n_out * 16));
    HIPCHK(hipMemcpy(a.p, in_cplx, n_in * 16, hipMemcpyHostToDevice));
    rd_launch_fir9((const double *)a.p, (double *)b.p, n_out, nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(out_cplx, b.p, n_out * 16, hipMemcpyDeviceToHost));
    return RD_OK;
}

extern "C" int rd_discriminate(const double *in_cplx, size_t n_in, double *out, size_t n_out) {
    if (n_in < 1 || n_out != n_in - 1) return fail(RD_ERR_ARG, "discriminate: n_out must be n_in - 1");
    if (n_out == 0) return RD_OK;
    if (!in_cplx || !out) return fail(RD_ERR_ARG, "null argument");
    STAGE_PROLOGUE();
    DevBuf a, b;
    HIPCHK(hipMalloc(&a.p, n_in * 16));
    HIPCHK(hipMalloc(&b.p, n_out * 8));
    HIPCHK(hipMemcpy(a.p, in_cplx, n_in * 16, hipMemcpyHostToDevice));
    rd_launch_discriminate((const double *)a.p, (double *)b.p, n_out, nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(out, b.p, n_out * 8, hipMemcpyDeviceToHost));
    return RD_OK;
}

extern "C" int rd_quantize(const double *in, uint8_t *out, size_t n) {
    if (n == 0) return RD_OK;
    if (!in || !out) return fail(RD_ERR_ARG, "null argument");
    STAGE_PROLOGUE();
    DevBuf a, b;
    HIPCHK(hipMalloc(&a.p, n * 8));
    HIPCHK(hipMalloc(&b.p, n));
    HIPCHK(hipMemcpy(a.p, in, n * 8, hipMemcpyHostToDevice));
    rd_launch_quantize((const double *)a.p, (uint8_t *)b.p, n, nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(out, b.p, n, hipMemcpyDeviceToHost));
    return RD_OK;
}

extern "C" int rd_search(const rd_config *cfg, const uint8_t *quantized, size_t n, int32_t *indices, int cap,
                         int *count) {
    if (!count) return fail(RD_ERR_ARG, "null count");
    if (!cfg) return fail(RD_ERR_ARG, "null config");
    rd_devcfg dc;
    rd_config c2 = *cfg;
    if (c2.block_size < 32) c2.block_size = 32;  // block size is irrelevant to a bare search
    c2.block_size -= c2.block_size % 4;
    int rc = make_devcfg(&c2, &dc);
    if (rc) return rc;
    *count = 0;
    const long span = (long)(dc.P - 1) * dc.S;
    if ((long)n <= span) return RD_OK;
    if (!quantized) return fail(RD_ERR_ARG, "null argument");
    STAGE_PROLOGUE();
    DevBuf a, w, m, cnt;
    const uint32_t mcap = (uint32_t)n;
    HIPCHK(hipMalloc(&a.p, n));
    HIPCHK(hipMalloc(&w.p, ((n + 31) / 32) * 4));
    HIPCHK(hipMalloc(&m.p, (size_t)mcap * sizeof(rd_match)));
    HIPCHK(hipMalloc(&cnt.p, RD_CNT_TOTAL * 4));
    HIPCHK(hipMemset(cnt.p, 0, RD_CNT_TOTAL * 4));
    HIPCHK(hipMemcpy(a.p, quantized, n, hipMemcpyHostToDevice));
    rd_launch_pack_bytes((const uint8_t *)a.p, (uint32_t *)w.p, n, nullptr);
    rd_launch_search((const uint32_t *)w.p, 0, 1, (long)n, 0, (long)n - 1 - span, dc, (rd_match *)m.p, mcap,
                     (uint32_t *)cnt.p, nullptr);
    HIPCHK(hipGetLastError());
    uint32_t h_cnt[RD_CNT_SLOTS];
    HIPCHK(hipMemcpy(h_cnt, cnt.p, sizeof h_cnt, hipMemcpyDeviceToHost));
    const uint32_t nm = std::min(h_cnt[RD_CNT_MATCH], mcap);
    std::vector<rd_match> ms(nm);
    if (nm) HIPCHK(hipMemcpy(ms.data(), m.p, (size_t)nm * sizeof(rd_match), hipMemcpyDeviceToHost));
    const int S = dc.S;
    std::sort(ms.begin(), ms.end(), [S](const rd_match &x, const rd_match &y) {
        const int px = x.pos % S, py = y.pos % S;
        return px != py ? px < py : x.pos < y.pos;
    });
    *count = (int)nm;
    if ((int)nm > cap) return fail(RD_ERR_CAPACITY, "need room for %u indices", nm);
    for (uint32_t i = 0; i < nm; i++) indices[i] = ms[i].pos;
    return RD_OK;
}
