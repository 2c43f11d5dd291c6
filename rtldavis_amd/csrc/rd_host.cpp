// rd_host.cpp - see rd_host.h.  Pure host code: compiled by hipcc into the library and by g++ -fsanitize into
// tests/_build/host_asan (tests/test_host_sanitizers.py).
#include "rd_host.h"

#include <sched.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>

// ------------------------------------------------------------------------------------------
// configuration (py:101-125)
// ------------------------------------------------------------------------------------------
int rd_make_devcfg(const rd_config *c, rd_devcfg *d, const char **why) {
    static const char *none = "";
    const char *dummy;
    if (!why) why = &dummy;
    *why = none;
    if (!c || !d) { *why = "null config"; return RD_ERR_ARG; }
    if (c->symbol_length < 1 || c->preamble_symbols < 1 || c->preamble_symbols > RD_MAX_PREAMBLE ||
        c->packet_symbols < 1 || c->packet_symbols > 8 * RD_MAX_PKT_BYTES) {
        *why = "unsupported packet configuration";
        return RD_ERR_ARG;
    }
    if (c->packet_symbols < c->preamble_symbols) { *why = "packet_symbols < preamble_symbols is not supported"; return RD_ERR_ARG; }
    if (c->block_size < 32 || c->block_size % 4) {
        *why = "block_size must be a multiple of 4 and >= 32 (rotate_fs4, py:46-49)";
        return RD_ERR_ARG;
    }
    if (c->bit_rate < 1) { *why = "bit_rate must be positive"; return RD_ERR_ARG; }
    // (symbol_length is bounded by the products below; both fit 64 bits for any int32 inputs)
    const long packet_length = (long)c->packet_symbols * c->symbol_length;
    const long preamble_length = (long)c->preamble_symbols * c->symbol_length;
    const long L = (packet_length / c->block_size + 2) * (long)c->block_size;
    if (L > 0x3FFFFFFF || preamble_length > 0x3FFFFFFF) { *why = "buffer_length too large"; return RD_ERR_ARG; }
    d->S = c->symbol_length;
    d->P = c->preamble_symbols;
    d->K = c->packet_symbols;
    d->B = c->block_size;
    d->PL = (int32_t)preamble_length;
    d->L = (int32_t)L;
    d->nbytes = (c->packet_symbols + 7) / 8;
    d->fs = (double)c->bit_rate * (double)c->symbol_length;
    d->pre_mask = 0;
    for (int i = 0; i < c->preamble_symbols; i++) {
        if (c->preamble[i] > 1) { *why = "preamble symbols must be 0 or 1"; return RD_ERR_ARG; }
        d->pre_mask |= (uint64_t)c->preamble[i] << i;
    }
    return RD_OK;
}

int rd_check_block_count(int is_complex, size_t count, size_t B, size_t NS, size_t *expected) {
    const size_t want = is_complex ? B : NS * 2 * B;
    if (expected) *expected = want;
    return count == want ? RD_OK : RD_ERR_ARG;
}

uint32_t rd_ord_bucket_cap(long n_samples) {
    if (n_samples < 0) n_samples = 0;
    // 4 x (n / 2^16) + 12, rounded up to a multiple of 32: 33 blocks of 8192 -> 32 (what the bench shape has always
    // used), 330 blocks -> 192
    const long want = (4 * n_samples + 65535) / 65536 + 12;
    long cap = ((want + 31) / 32) * 32;
    if (cap < RD_BUCKET_MIN) cap = RD_BUCKET_MIN;
    if (cap > RD_BUCKET_MAX) cap = RD_BUCKET_MAX;
    return (uint32_t)cap;
}

// ------------------------------------------------------------------------------------------
// order + dedupe.  Small lists: std::sort.  Large lists: the key is packed into 64 bits (when the field widths
// allow) and sorted with LSD counting passes over (key, index) pairs - a comparison sort that moves 64-byte records
// costs tens of milliseconds at 7e4 records.  `recs` may be pinned memory.
// ------------------------------------------------------------------------------------------
void rd_order_and_dedupe(const rd_packet *recs, size_t n, int S, rd_order_scratch &sc) {
    std::vector<uint32_t> &idx = sc.idx, &kept = sc.kept;
    kept.clear();
    idx.clear();
    if (n == 0 || S < 1) return;
    idx.reserve(n);
    for (size_t i = 0; i < n; i++)
        if (recs[i].stream >= 0) idx.push_back((uint32_t)i);  // stream < 0: match reported by no call
    n = idx.size();
    if (n == 0) return;
    auto less = [&](uint32_t ia, uint32_t ib) {
        const rd_packet &a = recs[ia], &b = recs[ib];
        if (a.stream != b.stream) return a.stream < b.stream;
        if (a.call != b.call) return a.call < b.call;
        const int pa = a.index % S, pb = b.index % S;
        if (pa != pb) return pa < pb;
        return a.index < b.index;
    };
    bool radix = n >= 512;
    if (radix) {
        // field widths (negative calls / indices never come from the kernels; they take the comparison sort)
        uint32_t max_stream = 0, max_call = 0, max_index = 0;
        for (size_t i = 0; i < n && radix; i++) {
            const rd_packet &r = recs[idx[i]];
            if (r.call < 0 || r.index < 0) radix = false;
            max_stream = std::max(max_stream, (uint32_t)r.stream);
            max_call = std::max(max_call, (uint32_t)r.call);
            max_index = std::max(max_index, (uint32_t)r.index);
        }
        auto bits_for = [](uint32_t v) { int b = 1; while (b < 32 && (v >> b)) b++; return b; };
        const int bi = bits_for(max_index), bp = bits_for((uint32_t)(S - 1)), bc = bits_for(max_call),
                  bs = bits_for(max_stream);
        if (radix && bi + bp + bc + bs <= 64) {
            std::vector<uint64_t> &key = sc.key, &ktmp = sc.ktmp;
            std::vector<uint32_t> &itmp = sc.itmp;
            key.resize(n); ktmp.resize(n); itmp.resize(n);
            for (size_t i = 0; i < n; i++) {
                const rd_packet &r = recs[idx[i]];
                key[i] = ((((uint64_t)(uint32_t)r.stream << bc | (uint32_t)r.call) << bp | (uint32_t)(r.index % S)) << bi) |
                         (uint32_t)r.index;
            }
            // LSD passes of 12 bits (the bench shape's 36-bit key: three)
            const int total_bits = bi + bp + bc + bs;
            constexpr int DB = 12;
            static thread_local uint32_t count[(1 << DB) + 1];
            for (int shift = 0; shift < total_bits; shift += DB) {
                memset(count, 0, sizeof count);
                for (size_t i = 0; i < n; i++) count[((key[i] >> shift) & ((1u << DB) - 1)) + 1]++;
                for (int d = 0; d < (1 << DB); d++) count[d + 1] += count[d];
                for (size_t i = 0; i < n; i++) {
                    const uint32_t pos = count[(key[i] >> shift) & ((1u << DB) - 1)]++;
                    ktmp[pos] = key[i];
                    itmp[pos] = idx[i];
                }
                key.swap(ktmp);
                idx.swap(itmp);
            }
        } else {
            radix = false;
        }
    }
    if (!radix) std::sort(idx.begin(), idx.end(), less);
    // per-call dedupe (py:203-205): the first occurrence of a byte string inside (stream, call) wins
    kept.reserve(n);
    size_t group = 0;
    for (size_t i = 0; i < n; i++) {
        const rd_packet &r = recs[idx[i]];
        if (kept.empty() || r.stream != recs[kept.back()].stream || r.call != recs[kept.back()].call) group = kept.size();
        const size_t nb = r.nbytes < 0 ? 0 : r.nbytes > RD_MAX_PKT_BYTES ? (size_t)RD_MAX_PKT_BYTES : (size_t)r.nbytes;
        bool dup = false;
        for (size_t k = group; k < kept.size() && !dup; k++) dup = memcmp(recs[kept[k]].data, r.data, nb) == 0;
        if (!dup) kept.push_back(idx[i]);
    }
}

// ------------------------------------------------------------------------------------------
// host waits
// ------------------------------------------------------------------------------------------
static double mono_ms() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec * 1e3 + (double)ts.tv_nsec * 1e-6;
}

static double g_timeout_override = -1.0;  // < 0: none

static double env_timeout_ms() {
    static const double v = [] {
        const char *e = getenv("RD_WAIT_TIMEOUT_MS");
        if (e && *e) {
            char *end = nullptr;
            const double x = strtod(e, &end);
            if (end != e && x >= 0.0) return x;
        }
        return 10000.0;
    }();
    return v;
}

double rd_wait_timeout_ms(void) { return g_timeout_override >= 0.0 ? g_timeout_override : env_timeout_ms(); }

double rd_wait_timeout_set(double ms) {
    const double prev = rd_wait_timeout_ms();
    g_timeout_override = ms < 0.0 ? -1.0 : ms;
    return prev;
}

rd_waiter::rd_waiter(double timeout_ms_) : t_start_ms(mono_ms()), timeout_ms(timeout_ms_), polls(0) {}

double rd_waiter::waited_ms() const { return mono_ms() - t_start_ms; }

bool rd_waiter::relax() {
    polls++;
    if (timeout_ms <= 0.0) return false;  // (the test hook: whatever is not ready at its first poll times out)
    // the clock is read at the first poll and every 32 polls while spinning (a poll of a flag in pinned memory takes
    // tens of nanoseconds; a hipEventQuery about a microsecond), on every poll afterwards
    if (polls < 1024 && (polls & 31) != 1) {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
        return true;
    }
    const double waited = mono_ms() - t_start_ms;
    if (waited >= timeout_ms) return false;
    if (waited < 0.05) {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
        if (polls >= 1024) polls = 992;  // stay in the spinning regime until 50 us have passed
    } else if (waited < 1.0) {
        sched_yield();
    } else {
        struct timespec ts = {0, 50000};  // 50 us: an overdue wait does not pin a core
        nanosleep(&ts, nullptr);
    }
    return true;
}
