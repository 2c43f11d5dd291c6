/*
 * Sanitizer driver for oracle/dsp_oracle.c.  TEST INFRASTRUCTURE ONLY (`make -C oracle asan`): the oracle is built
 * with -fsanitize=address,undefined together with this main and run on the CPU by tests/test_host_sanitizers.py
 * (the GPU pool allows no sanitizer runs).
 *   oracle_asan <iq.u8> <n_samples> <block_size> <symbol_length> <out.bin>
 * demodulates the stream (Davis preamble and packet length) and writes: int64 packet count, the packets
 * (oracle_pkt), the packed bits ((n + 7) / 8 bytes) - the same bytes liboracle.so must produce.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dsp_oracle.c"

int main(int argc, char **argv) {
    if (argc != 6) { fprintf(stderr, "usage: oracle_asan iq.u8 n_samples block_size symbol_length out.bin\n"); return 2; }
    const long n = atol(argv[2]);
    oracle_cfg c;
    memset(&c, 0, sizeof c);
    c.bit_rate = 19200; c.symbol_length = atoi(argv[4]); c.preamble_symbols = 16; c.packet_symbols = 80; c.block_size = atoi(argv[3]);
    const char *pre = "1100101110001001";
    for (int i = 0; i < 16; i++) c.preamble[i] = (uint8_t)(pre[i] - '0');
    uint8_t *iq = malloc((size_t)(2 * n));
    FILE *f = fopen(argv[1], "rb");
    if (!f || !iq || fread(iq, 1, (size_t)(2 * n), f) != (size_t)(2 * n)) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    fclose(f);
    const long cap = 4096;
    oracle_pkt *out = calloc((size_t)cap, sizeof *out);
    uint8_t *bits = malloc((size_t)((n + 7) / 8));
    double *disc = malloc(sizeof(double) * (size_t)n);
    const long cnt = oracle_demod_stream(iq, n, &c, 0, bits, disc, out, cap);
    if (cnt < 0) { fprintf(stderr, "oracle_demod_stream failed\n"); return 1; }
    /* the stage entry point and the small helpers as well */
    double *filt = malloc(sizeof(double) * 2 * (size_t)(n + 1));
    uint8_t *b01 = malloc((size_t)n);
    oracle_stages(iq, n, filt, disc, b01);
    (void)oracle_crc16_ccitt(iq, 8);
    (void)oracle_swap_bit_order(iq[0]);
    f = fopen(argv[5], "wb");
    if (!f) return 2;
    long long c64 = cnt;
    fwrite(&c64, sizeof c64, 1, f);
    fwrite(out, sizeof *out, (size_t)cnt, f);
    fwrite(bits, 1, (size_t)((n + 7) / 8), f);
    fclose(f);
    free(iq); free(out); free(bits); free(disc); free(filt); free(b01);
    return 0;
}
