/* oracle_sanitize.c -- runs the oracle's whole path (keygen, encrypt, every gate, MUX, exact step) under
 * AddressSanitizer + UndefinedBehaviorSanitizer on the CPU (GPU sanitizers are not available on the pool). */
#include "tfhe_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(void)
{
    orc_params p;
    if (orc_default_params(1, &p)) return 1;
    p.n = 10; /* Set B shape (l = 3), small LWE dimension */
    const int kpl = 2 * p.l, base = 1 << p.ks_basebit;
    int32_t *lwe = calloc(p.n, 4), *tlwe = calloc(ORC_N, 4);
    orc_keygen_secret(&p, 7, lwe, tlwe);
    size_t ksk_rows = (size_t)ORC_N * p.ks_t * (base - 1);
    int32_t *ksk = calloc(ksk_rows * (p.n + 1), 4);
    int32_t *bk = calloc((size_t)p.n * kpl * 2 * ORC_N, 4);
    double *bkfft = calloc((size_t)p.n * kpl * 2 * ORC_N, 8);
    orc_keygen_ksk(&p, 7, lwe, tlwe, ksk);
    orc_keygen_bk(&p, 7, lwe, tlwe, bk);
    orc_bk_to_fft(&p, bk, bkfft);
    size_t st = p.n + 1;
    int32_t *a = calloc(4 * st, 4), *b = calloc(4 * st, 4), *c = calloc(4 * st, 4), *o = calloc(4 * st, 4);
    int bad = 0;
    for (int i = 0; i < 4; i++) {
        orc_encrypt_bit(&p, lwe, 3, i, i >> 1, a + i * st);
        orc_encrypt_bit(&p, lwe, 4, i, i & 1, b + i * st);
        orc_encrypt_bit(&p, lwe, 5, i, (i + 1) & 1, c + i * st);
    }
    for (int op = 0; op <= 12; op++) {
        if (orc_gate_batch(&p, bkfft, ksk, op, NULL, a, b, c, o, 4, 2)) bad++;
        for (int i = 0; i < 4; i++) {
            int x = i >> 1, y = i & 1, z = (i + 1) & 1, want;
            switch (op) {
            case ORC_NAND: want = !(x & y); break;   case ORC_AND: want = x & y; break;
            case ORC_OR: want = x | y; break;        case ORC_NOR: want = !(x | y); break;
            case ORC_XOR: want = x ^ y; break;       case ORC_XNOR: want = !(x ^ y); break;
            case ORC_ANDNY: want = !x & y; break;    case ORC_ANDYN: want = x & !y; break;
            case ORC_ORNY: want = !x | y; break;     case ORC_ORYN: want = x | !y; break;
            case ORC_MUX: want = x ? y : z; break;   case ORC_NOT: want = !x; break;
            default: want = x;
            }
            if (orc_decrypt_bit(&p, lwe, o + i * st) != want) bad++;
        }
    }
    /* exact-integer CMux step next to the FFT step */
    int32_t acc1[2 * ORC_N], acc2[2 * ORC_N];
    for (int j = 0; j < 2 * ORC_N; j++) acc1[j] = acc2[j] = (int32_t)(orc_rng_u64(99, j) >> 32);
    orc_blind_rotate_step(&p, bkfft + (size_t)2 * kpl * 2 * ORC_N, NULL, 1234, acc1, 1);
    orc_blind_rotate_step(&p, NULL, bk + (size_t)2 * kpl * 2 * ORC_N, 1234, acc2, 0);
    for (int j = 0; j < 2 * ORC_N; j++) {
        int32_t d = (int32_t)((uint32_t)acc1[j] - (uint32_t)acc2[j]);
        if (d > 8 || d < -8) bad++;
    }
    if (orc_gate(&p, bkfft, ksk, 99, a, b, c, o) != -1) bad++; /* bad opcode is an error, not a crash */
    printf(bad ? "oracle_sanitize FAILED (%d)\n" : "oracle_sanitize OK\n", bad);
    free(lwe); free(tlwe); free(ksk); free(bk); free(bkfft); free(a); free(b); free(c); free(o);
    return bad != 0;
}
