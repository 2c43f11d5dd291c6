// Sanitizer harness for the pure-host translation unit of the C ABI (rtldavis_amd/csrc/rd_host.cpp).
// TEST INFRASTRUCTURE ONLY: built by tests/test_host_sanitizers.py with
//   g++ -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all host_asan.cpp ../rtldavis_amd/csrc/rd_host.cpp
// and run on the CPU (the GPU pool allows no sanitizer runs).  Exit code 0 = every check held and
// no sanitizer report.
//   host_asan soak <rounds> <seed>     random record sets through rd_order_and_dedupe against a naive model
//   host_asan file <in> <out> <S>      rd_packet records from <in>, kept indices (uint32) to <out>
//   host_asan config <rounds> <seed>   rd_make_devcfg / rd_check_block_count / rd_ord_bucket_cap on random and extreme input
//   host_asan waiter                   rd_waiter: deadline 0, a short deadline, the override hook
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include <algorithm>

#include "../rtldavis_amd/csrc/rd_host.h"

static int fail(const char *what) { fprintf(stderr, "host_asan: %s\n", what); return 1; }

// the reference's rule, spelled out naively: stable order by (stream, call, index % S, index), then first occurrence
// of a byte string inside (stream, call) wins (py:171-205)
static std::vector<uint32_t> naive(const std::vector<rd_packet> &r, int S) {
    std::vector<uint32_t> idx;
    for (size_t i = 0; i < r.size(); i++) if (r[i].stream >= 0) idx.push_back((uint32_t)i);
    std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) {
        const rd_packet &x = r[a], &y = r[b];
        if (x.stream != y.stream) return x.stream < y.stream;
        if (x.call != y.call) return x.call < y.call;
        if (x.index % S != y.index % S) return x.index % S < y.index % S;
        return x.index < y.index;
    });
    std::vector<uint32_t> kept;
    for (uint32_t i : idx) {
        bool dup = false;
        for (uint32_t k : kept)
            if (r[k].stream == r[i].stream && r[k].call == r[i].call && memcmp(r[k].data, r[i].data, (size_t)r[i].nbytes) == 0) { dup = true; break; }
        if (!dup) kept.push_back(i);
    }
    return kept;
}

static int soak(int rounds, unsigned seed) {
    std::mt19937 g(seed);
    rd_order_scratch sc;  // reused across rounds, like a handle's
    for (int round = 0; round < rounds; round++) {
        const int S = 1 + (int)(g() % 20);
        const size_t n = (round % 3 == 0) ? g() % 40 : (round % 3 == 1) ? 400 + g() % 400 : 2000 + g() % 3000;
        const int n_streams = 1 + (int)(g() % (round % 5 == 0 ? 100000 : 50)), n_calls = 1 + (int)(g() % 40);
        const int n_index = 1 + (int)(g() % (round % 7 == 0 ? (1 << 24) : 9000)), n_bytes = 1 + (int)(g() % RD_MAX_PKT_BYTES);
        const int n_payloads = 1 + (int)(g() % 6);
        std::vector<rd_packet> r(n);
        for (auto &p : r) {
            memset(&p, 0, sizeof p);
            p.stream = (g() % 16 == 0) ? -1 : (int)(g() % n_streams);
            p.call = (int)(g() % n_calls);
            p.index = (int)(g() % n_index);
            p.nbytes = n_bytes;
            // few distinct payloads (duplicates inside a call are common), a function of the position like a real
            // packet's bytes: records with equal (stream, call, index) are then equal everywhere the order could show
            const unsigned pay = (unsigned)(p.index % n_payloads) + 7u * (unsigned)(p.stream & 1);
            for (int k = 0; k < n_bytes; k++) p.data[k] = (uint8_t)(pay * 37 + k);
        }
        rd_order_and_dedupe(r.data(), r.size(), S, sc);
        const std::vector<uint32_t> want = naive(r, S);
        if (want.size() != sc.kept.size()) return fail("kept count differs from the naive model");
        for (size_t i = 0; i < want.size(); i++) {
            const rd_packet &a = r[want[i]], &b = r[sc.kept[i]];
            if (a.stream != b.stream || a.call != b.call || a.index != b.index || memcmp(a.data, b.data, (size_t)a.nbytes))
                return fail("kept record differs from the naive model");
        }
    }
    rd_order_and_dedupe(nullptr, 0, 14, sc);
    if (!sc.kept.empty()) return fail("empty input kept something");
    printf("soak ok: %d rounds\n", rounds);
    return 0;
}

static int file_mode(const char *in, const char *out, int S) {
    FILE *f = fopen(in, "rb");
    if (!f) return fail("cannot open input");
    std::vector<rd_packet> r;
    rd_packet p;
    while (fread(&p, sizeof p, 1, f) == 1) r.push_back(p);
    fclose(f);
    rd_order_scratch sc;
    rd_order_and_dedupe(r.data(), r.size(), S, sc);
    f = fopen(out, "wb");
    if (!f) return fail("cannot open output");
    if (!sc.kept.empty() && fwrite(sc.kept.data(), 4, sc.kept.size(), f) != sc.kept.size()) return fail("short write");
    fclose(f);
    return 0;
}

static int config(int rounds, unsigned seed) {
    std::mt19937 g(seed);
    const int32_t extremes[] = {0, 1, -1, 4, 31, 32, 36, 512, 8192, 1 << 20, 0x3FFFFFFF, 0x7FFFFFFF, (int32_t)0x80000000};
    auto pick = [&]() -> int32_t { return (g() % 3) ? extremes[g() % (sizeof extremes / sizeof extremes[0])] : (int32_t)(g() % 100000) - 10; };
    int ok = 0;
    for (int i = 0; i < rounds; i++) {
        rd_config c;
        memset(&c, 0, sizeof c);
        c.bit_rate = pick(); c.symbol_length = pick(); c.preamble_symbols = (g() % 2) ? pick() : 1 + (int)(g() % 64);
        c.packet_symbols = (g() % 2) ? pick() : 1 + (int)(g() % 256); c.block_size = (g() % 2) ? pick() : 4 * (8 + (int)(g() % 4096));
        for (int k = 0; k < RD_MAX_PREAMBLE; k++) c.preamble[k] = (uint8_t)((g() % 50 == 0) ? 2 : g() & 1);
        rd_devcfg d;
        memset(&d, 0, sizeof d);
        const char *why = nullptr;
        const int rc = rd_make_devcfg(&c, &d, &why);
        if (rc != RD_OK && rc != RD_ERR_ARG) return fail("unexpected status");
        if (!why) return fail("why left null");
        if (rc == RD_OK) {
            ok++;
            if (d.L < 2 * d.B || d.L % d.B || d.PL != d.P * d.S || d.nbytes != (d.K + 7) / 8 || d.nbytes > RD_MAX_PKT_BYTES)
                return fail("derived constants inconsistent");
        } else if (!*why) return fail("failure without a reason");
    }
    {   // the production shape (protocol.py:68-76)
        rd_config c = {19200, 14, 16, 80, 8192, {1, 1, 0, 0, 1, 0, 1, 1, 1, 0, 0, 0, 1, 0, 0, 1}};
        rd_devcfg d;
        const char *why;
        if (rd_make_devcfg(&c, &d, &why) != RD_OK || d.L != 16384 || d.PL != 224 || d.nbytes != 10 || d.pre_mask != 0x91D3ull || d.fs != 268800.0)
            return fail("production shape");
        if (rd_make_devcfg(nullptr, &d, &why) != RD_ERR_ARG || rd_make_devcfg(&c, &d, nullptr) != RD_OK) return fail("null handling");
    }
    if (rd_ord_bucket_cap(33 * 8192) != 32 || rd_ord_bucket_cap(330 * 8192) != 192 || rd_ord_bucket_cap(0) != RD_BUCKET_MIN ||
        rd_ord_bucket_cap(-5) != RD_BUCKET_MIN || rd_ord_bucket_cap(0x7FFFFFF0L) != RD_BUCKET_MAX)
        return fail("bucket capacity");
    for (long n = 0; n < (1l << 31); n += 999983) {
        const uint32_t c = rd_ord_bucket_cap(n);
        if (c % 32 || c < RD_BUCKET_MIN || c > RD_BUCKET_MAX) return fail("bucket capacity range");
    }
    size_t want = 0;
    if (rd_check_block_count(0, 16384, 8192, 1, &want) != RD_OK || want != 16384 || rd_check_block_count(1, 8192, 8192, 1, &want) != RD_OK ||
        rd_check_block_count(1, 16384, 8192, 1, &want) != RD_ERR_ARG || want != 8192 || rd_check_block_count(0, 3 * 16384, 8192, 3, nullptr) != RD_OK)
        return fail("block count");
    printf("config ok: %d of %d random configurations accepted\n", ok, rounds);
    return 0;
}

static int waiter() {
    const double dflt = rd_wait_timeout_ms();
    if (dflt <= 0) return fail("default deadline");
    if (rd_wait_timeout_set(0) != dflt || rd_wait_timeout_ms() != 0) return fail("override");
    { rd_waiter w(rd_wait_timeout_ms()); if (w.relax()) return fail("deadline 0 must expire at the first poll"); }
    rd_wait_timeout_set(-1);
    if (rd_wait_timeout_ms() != dflt) return fail("restore");
    rd_waiter w(3.0);  // 3 ms: spins, yields, sleeps, then expires
    unsigned polls = 0;
    while (w.relax()) if (++polls > 100000000u) return fail("a 3 ms deadline never expired");
    if (w.waited_ms() < 3.0 || w.waited_ms() > 500.0) return fail("expired at the wrong time");
    printf("waiter ok: %u polls in %.2f ms\n", polls, w.waited_ms());
    return 0;
}

int main(int argc, char **argv) {
    if (argc >= 4 && !strcmp(argv[1], "soak")) return soak(atoi(argv[2]), (unsigned)atoi(argv[3]));
    if (argc >= 5 && !strcmp(argv[1], "file")) return file_mode(argv[2], argv[3], atoi(argv[4]));
    if (argc >= 4 && !strcmp(argv[1], "config")) return config(atoi(argv[2]), (unsigned)atoi(argv[3]));
    if (argc >= 2 && !strcmp(argv[1], "waiter")) return waiter();
    return fail("usage: soak <rounds> <seed> | file <in> <out> <S> | config <rounds> <seed> | waiter");
}
