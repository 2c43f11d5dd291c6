// CPU harness around rtldavis_amd/csrc/rd_math.h.  TEST INFRASTRUCTURE ONLY: it lets the
// CPU test-suite run the exact arithmetic the HIP kernels execute (same inline functions,
// same fp32 fma sequence) against the oracle without a GPU.  Nothing in the product links it.
#include <cstring>
#include <vector>

#include "../rtldavis_amd/csrc/rd_math.h"

extern "C" {

// Fast path over a whole stream from reset.  words[n/32]; flagged[n/32] = 4-bit mask of the
// 8-sample groups that must be re-evaluated exactly (guard band; run 0, whose history is the
// zero state, is always re-evaluated whole).
long hh_fast_stream(const uint8_t *iq, long n, uint32_t *words, uint8_t *flagged) {
    long nflag = 0;
    for (long t0 = 0; t0 + RD_RUN <= n; t0 += RD_RUN) {
        uint8_t win[2 * RD_WIN];
        for (int i = 0; i < RD_WIN; i++) {
            long s = t0 - RD_HALO + i;
            win[2 * i] = s >= 0 ? iq[2 * s] : 0;
            win[2 * i + 1] = s >= 0 ? iq[2 * s + 1] : 0;
        }
        rd_ptr_src src = {win};
        rd_run_result r = rd_fast_run(src);
        words[t0 / RD_RUN] = r.word;
        uint8_t f = (t0 == 0) ? 0xF : (uint8_t)rd_guard_mask(r);
        flagged[t0 / RD_RUN] = f;
        nflag += __builtin_popcount(f);
    }
    return nflag;
}

// exact bits of one 8-sample group (what k_fixup stores as one byte)
uint8_t hh_exact_group(const uint8_t *iq, long n, long t0) {
    rd_stream_view v = {iq, 0, n};
    return (uint8_t)rd_exact_run(v, t0, (int)(n - t0 < RD_GROUP ? n - t0 : RD_GROUP));
}

// the same group through the dword-window form k_fixup uses
uint8_t hh_exact_group_dw(const uint8_t *iq, long n, long t0) {
    uint32_t dw[10];
    for (int d = 0; d < 10; d++) {
        const long s0 = t0 - 10 + 2 * d;
        dw[d] = 0;
        if (s0 + 1 >= 0 && s0 < n + 8) {
            uint8_t b[4] = {0, 0, 0, 0};
            for (int k = 0; k < 4; k++) {
                const long idx = 2 * s0 + k;
                if (idx >= 0 && idx < 2 * n) b[k] = iq[idx];
            }
            memcpy(&dw[d], b, 4);
        }
    }
    return (uint8_t)rd_exact_group_dw(dw, t0, (int)(n - t0 < RD_GROUP ? n - t0 : RD_GROUP), 0);
}

void hh_exact_stream(const uint8_t *iq, long n, uint32_t *words) {
    rd_stream_view v = {iq, 0, n};
    for (long t0 = 0; t0 < n; t0 += RD_RUN) {
        int cnt = (int)(n - t0 < RD_RUN ? n - t0 : RD_RUN);
        words[t0 / RD_RUN] = rd_exact_run(v, t0, cnt);
    }
}

// float64 stages: f[-1..n-1] interleaved (2(n+1) doubles) and d[0..n-1].
void hh_f64_stream(const uint8_t *iq, long n, double *filt, double *disc) {
    rd_stream_view v = {iq, 0, n};
    rd_d2 prev = rd_f_f64(v, -1);
    filt[0] = prev.x; filt[1] = prev.y;
    for (long t = 0; t < n; t++) {
        rd_d2 cur = rd_f_f64(v, t);
        filt[2 * (t + 1)] = cur.x; filt[2 * (t + 1) + 1] = cur.y;
        disc[t] = rd_disc_f64(prev, cur);
        prev = cur;
    }
}

float hh_threshold(float F) { return rd_run_threshold(F); }
}
