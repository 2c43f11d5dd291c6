// rd_host.h - the pure-host parts of the C ABI: nothing in here touches HIP, so the same translation unit
// (rd_host.cpp) is compiled into librtldavis_hip.so by hipcc AND into tests/host_asan by g++ with
// -fsanitize=address,undefined (SURVEY section 5's sanitizer stance; the GPU pool allows no sanitizer runs).
//   rd_make_devcfg        PacketConfig's derived constants (py:101-125) and the limits the kernels rely on
//   rd_order_and_dedupe   the per-call order and first-occurrence dedupe of Demodulator._slice (py:171-205)
//   rd_ord_bucket_cap     matches a stream's list in k_tail's LDS holds, from the stream's length
//   rd_check_block_count  "Incompatible array sizes" (py:32-36, py:145-149)
//   rd_waiter             the polling policy of every host wait: spin, then yield, then sleep, with a deadline
// py = /root/reference/src/rtldavis/dsp.py
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <vector>

#include "../../include/rtldavis_hip.h"

struct rd_devcfg {
    int32_t S, P, K, B, L, PL, nbytes;  // symbol_length, preamble/packet symbols, block, buffer, preamble_length
    uint64_t pre_mask;                  // bit m = preamble symbol m
    double fs;                          // sample rate = bit_rate * symbol_length (protocol.py:309)
};

// RD_OK or RD_ERR_ARG; *why (never null on failure) names the offending field
int rd_make_devcfg(const rd_config *c, rd_devcfg *d, const char **why);

// Reference order inside one call: search is phase-major then ascending (py:175-186), slice keeps the first
// occurrence of each byte string (py:203-205).  Records are ordered by (stream, call, index % S, index); records with
// stream < 0 (a match reported by no call) are skipped.  Fills sc.kept with the indices of the records to return.
struct rd_order_scratch {  // kept per handle: no allocation per call once warm
    std::vector<uint32_t> idx, itmp, kept;
    std::vector<uint64_t> key, ktmp;
};
void rd_order_and_dedupe(const rd_packet *recs, size_t n, int S, rd_order_scratch &sc);

// k_tail: matches one stream's list must hold.  Noise gives one raw preamble match per 2^16 positions
// (~4 per second of a Davis stream), a burst a handful around its true position; the
// capacity is 4x the noise expectation plus 12 for bursts, rounded up to a multiple of 32, at least 32 and at
// most RD_BUCKET_MAX.
#define RD_BUCKET_MIN 32
#define RD_BUCKET_MAX 512
uint32_t rd_ord_bucket_cap(long n_samples);

// py:32-36 / py:145-149: RD_OK when `count` elements are what one call takes (complex: B samples of one stream;
// uint8: NS * 2B bytes), else RD_ERR_ARG with *expected set
int rd_check_block_count(int is_complex, size_t count, size_t B, size_t NS, size_t *expected);

// Host waits poll (the runtime's blocking waits add 10-20 ms of wake-up latency on this platform).  A wait spins on
// `pause` for the first ~50 us, then yields the core, then sleeps 50 us at a time; it gives up at its deadline.
struct rd_waiter {
    double t_start_ms, timeout_ms;
    uint32_t polls;
    explicit rd_waiter(double timeout_ms_);
    // call after an unsuccessful poll: false when the deadline has passed (the caller reports a timeout)
    bool relax();
    double waited_ms() const;
};
// RD_WAIT_TIMEOUT_MS (read once; default 10000) unless rd_set_wait_timeout_ms has set another value
double rd_wait_timeout_ms(void);
// returns the previous value; ms < 0 restores the environment's / default value.  0 = every wait that is not
// already satisfied at its first poll times out (the test hook of tests/test_gpu_parity.py::test_host_wait_deadline)
double rd_wait_timeout_set(double ms);
