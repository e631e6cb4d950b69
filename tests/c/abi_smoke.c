#define _DEFAULT_SOURCE /* mkstemp */
/* abi_smoke.c -- a plain-C host of libeoc_tfhe_gpu.so: proves include/eoc_tfhe_gpu.h is valid C and the
 * library is usable without Python or torch (this is what a Lua/Node binding sits on).
 *   gcc -std=c11 -Iinclude tests/c/abi_smoke.c -o abi_smoke -Leoc_tfhe_amd -leoc_tfhe_gpu -Wl,-rpath,...
 * usage: abi_smoke cpu   -> client-side calls only; the gate call must FAIL with EOC_ERR_NO_DEVICE when no GPU
 *        abi_smoke gpu   -> full NAND truth table through the host-buffer batch API and the string API */
#include "eoc_tfhe_gpu.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "FAILED: %s (line %d): %s\n", #c, __LINE__, eoc_last_error()); return 1; } } while (0)

int main(int argc, char **argv)
{
    int gpu = argc > 1 && strcmp(argv[1], "gpu") == 0;
    eoc_params p;
    CHECK(eoc_default_params(0, &p) == EOC_OK && p.n == 500);
    if (!gpu) p.n = 32; /* small key: the CPU leg only exercises client-side code */
    eoc_secret_key *sk = NULL;
    CHECK(eoc_keygen(&p, 1, 1, &sk) == EOC_OK);
    const uint8_t b0[4] = {0, 0, 1, 1}, b1[4] = {0, 1, 0, 1};
    size_t st = (size_t)p.n + 1;
    int32_t *c0 = malloc(4 * st * 4), *c1 = malloc(4 * st * 4), *out = malloc(4 * st * 4);
    uint8_t dec[4];
    CHECK(eoc_encrypt_bits(sk, 2, 0, b0, 4, c0) == EOC_OK && eoc_encrypt_bits(sk, 3, 0, b1, 4, c1) == EOC_OK);
    CHECK(eoc_decrypt_bits(sk, c0, 4, dec) == EOC_OK && memcmp(dec, b0, 4) == 0);
    /* key blobs round trip */
    size_t need = eoc_secret_key_export(sk, NULL, 0);
    void *blob = malloc(need);
    CHECK(eoc_secret_key_export(sk, blob, need) == need);
    eoc_secret_key *sk2 = NULL;
    CHECK(eoc_secret_key_import(blob, need, 0, &sk2) == EOC_OK);
    CHECK(memcmp(eoc_sk_lwe_key(sk), eoc_sk_lwe_key(sk2), (size_t)p.n * 4) == 0);
    eoc_secret_key_free(sk2);
    {   /* host-side netlist rewriting: NOT s; (s & x) | (~s & y)  ->  one MUX */
        const eoc_gate nl[4] = {{EOC_NOT, 0, -1, -1, 3}, {EOC_AND, 0, 1, -1, 4}, {EOC_AND, 3, 2, -1, 5},
                                {EOC_OR, 4, 5, -1, 6}};
        const int32_t outs[1] = {6};
        eoc_gate opt[4];
        CHECK(eoc_circuit_bootstraps(nl, 4) == 3);
        CHECK(eoc_netlist_optimize(nl, 4, outs, 1, opt) == 1);
        CHECK(opt[0].op == EOC_MUX && opt[0].in0 == 0 && opt[0].in1 == 1 && opt[0].in2 == 2 && opt[0].out == 6);
        const eoc_gate bad[2] = {{EOC_AND, 0, 1, -1, 2}, {EOC_OR, 0, 1, -1, 2}};
        CHECK(eoc_netlist_optimize(bad, 2, outs, 0, opt) == EOC_ERR_ARG);
        /* round 6: levels (the engine's: NOT is free and sits in the pre-pass of its reader's level) and the level-cost estimate */
        int32_t lev[4];
        int64_t depth = 0;
        CHECK(eoc_netlist_levels(nl, 4, lev, &depth) == 2 && depth == 2);
        CHECK(lev[0] == 1 && lev[1] == 1 && lev[2] == 1 && lev[3] == 2);
        CHECK(eoc_netlist_cost(nl, 4, 3, 0) == 36 && eoc_netlist_cost(nl, 4, 1024, 1024) == 60 + 30);
        CHECK(eoc_netlist_levels(bad, 2, NULL, NULL) == 2);                     /* running needs no single assignment */
        /* slots an opcode does not use may hold anything (here: far out of range, negative): ignored, never indexed */
        const eoc_gate sloppy[4] = {{EOC_NOT, 0, 31000, -7, 3}, {EOC_AND, 0, 1, 99999, 4}, {EOC_AND, 3, 2, -5, 5}, {EOC_OR, 4, 5, 77, 6}};
        CHECK(eoc_netlist_optimize(sloppy, 4, outs, 1, opt) == 1 && opt[0].op == EOC_MUX && opt[0].in2 == 2);
        CHECK(eoc_netlist_levels(sloppy, 4, lev, &depth) == 2 && eoc_netlist_cost(sloppy, 4, 3, 0) == 36);
        const eoc_gate huge[1] = {{EOC_AND, 0, 1, -1, 2147483647}};
        CHECK(eoc_netlist_optimize(huge, 1, outs, 1, opt) == EOC_ERR_ARG && eoc_netlist_levels(huge, 1, NULL, NULL) == EOC_ERR_ARG);
        const eoc_gate junk[1] = {{99, 0, 1, -1, 2}};
        CHECK(eoc_netlist_levels(junk, 1, NULL, NULL) == EOC_ERR_ARG && eoc_netlist_cost(junk, 1, 1, 0) == EOC_ERR_ARG);
    }
    {   /* BASELINE configs[2]'s literal 8-bit adder (full adder at bit 0, carry-in bootsCONSTANT(0): 40 bootstraps on 17
         * levels) through the optimizer: the carry OR(AND(a,b), AND(XOR(a,b),c)) rewritten + constant folding.  Wires: a 0..7, b 8..15, sum 16..24, carry-in 25, temporaries 26.. */
        eoc_gate ad[41], ad_opt[41];
        int32_t sum[9], nxt = 26, carry = 25;
        size_t ng = 0;
        ad[ng++] = (eoc_gate){EOC_CONST0, -1, -1, -1, 25};
        for (int i = 0; i < 8; i++) {
            const int32_t pw = nxt, gw = nxt + 1, pc = nxt + 2, nc = i == 7 ? 24 : nxt + 3;
            nxt += i == 7 ? 3 : 4;
            ad[ng++] = (eoc_gate){EOC_XOR, i, 8 + i, -1, pw};
            ad[ng++] = (eoc_gate){EOC_AND, i, 8 + i, -1, gw};
            ad[ng++] = (eoc_gate){EOC_XOR, pw, carry, -1, 16 + i};
            ad[ng++] = (eoc_gate){EOC_AND, pw, carry, -1, pc};
            ad[ng++] = (eoc_gate){EOC_OR, gw, pc, -1, nc};
            carry = nc;
        }
        for (int i = 0; i < 9; i++) sum[i] = 16 + i;
        int64_t d0 = 0, d1 = 0;
        CHECK(ng == 41 && eoc_circuit_bootstraps(ad, ng) == 40 && eoc_netlist_levels(ad, ng, NULL, &d0) > 0 && d0 == 17);
        /* inside libtfhe's gate family: the carry as MUX, the constant carry-in folded: 30 bootstraps on 8 levels */
        int64_t no = eoc_netlist_optimize_ex(ad, ng, sum, 9, ad_opt, EOC_NL_BOOTS_GATES_ONLY);
        CHECK(no > 0 && no <= 41 && eoc_circuit_bootstraps(ad_opt, (size_t)no) == 30);
        CHECK(eoc_netlist_levels(ad_opt, (size_t)no, NULL, &d1) > 0 && d1 == 8);
        int n_mux = 0, n_maj = 0, n_xor3 = 0;
        for (int64_t k = 0; k < no; k++) n_mux += ad_opt[k].op == EOC_MUX;
        CHECK(n_mux == 7);                                                       /* bit 0's MUX(p, 0, a) folded to ANDNY */
        CHECK(eoc_netlist_cost(ad_opt, (size_t)no, 8, 0) < eoc_netlist_cost(ad, ng, 8, 0));
        CHECK(eoc_netlist_cost(ad_opt, (size_t)no, 4096, 0) == 30 * 4 * 30 && eoc_netlist_cost(ad, ng, 4096, 0) == 40 * 4 * 30);
        /* default: with the extension gates a full adder is XOR3 + MAJ, one bootstrap each: 16 on 8 levels */
        no = eoc_netlist_optimize(ad, ng, sum, 9, ad_opt);
        CHECK(no > 0 && no <= 41 && eoc_circuit_bootstraps(ad_opt, (size_t)no) == 16);
        CHECK(eoc_netlist_levels(ad_opt, (size_t)no, NULL, &d1) > 0 && d1 == 8);
        for (int64_t k = 0; k < no; k++) { n_maj += ad_opt[k].op == EOC_MAJ; n_xor3 += ad_opt[k].op == EOC_XOR3; }
        CHECK(n_maj == 7 && n_xor3 == 7);                                        /* bit 0 folds to XOR + AND */
        CHECK(eoc_netlist_cost(ad_opt, (size_t)no, 4096, 0) == 16 * 4 * 30);
    }
    {   /* the optimizer on 4000 random single-assignment netlists over every opcode (this block also runs in the ASan /
         * UBSan build, tools/sanitize_host.sh): each wire's truth table over the 4 circuit inputs as a 16-bit mask, before
         * and after -- same outputs, never more bootstraps, never more levels, with and without the extension gates */
        uint32_t seed = 12345u;
#define RND(m) ((seed = seed * 1664525u + 1013904223u) >> 8) % (uint32_t)(m)
        for (int trial = 0; trial < 4000; trial++) {
            enum { NIN = 4, MAXG = 48 };
            eoc_gate g[MAXG], o[MAXG];
            const int ng = 1 + (int)(RND(MAXG));
            for (int k = 0; k < ng; k++) {
                const int avail = NIN + k, op = (int)(RND(17));
                g[k] = (eoc_gate){op, (int32_t)(RND(avail)), (int32_t)(RND(avail)), (int32_t)(RND(avail)), NIN + k};
                if (op != EOC_MUX && op != EOC_MAJ && op != EOC_XOR3 && RND(8) == 0)
                    g[k].in2 = -7 - (int32_t)(RND(1000));                         /* junk in a slot the gate does not use */
                if (op == EOC_CONST0 || op == EOC_CONST1) g[k].in0 = -1;
            }
            int32_t outs[3];
            const int no_ = ng < 3 ? ng : 3;
            for (int k = 0; k < no_; k++) outs[k] = NIN + (int32_t)(RND(ng));
            for (unsigned flags = 0; flags < 2; flags++) {
                const int64_t m = eoc_netlist_optimize_ex(g, (size_t)ng, outs, (size_t)no_, o, flags ? EOC_NL_BOOTS_GATES_ONLY : 0);
                CHECK(m >= 0 && m <= ng);
                uint16_t ref[NIN + MAXG], got[NIN + MAXG];
                const uint16_t in_mask[NIN] = {0xAAAA, 0xCCCC, 0xF0F0, 0xFF00};
                for (int pass = 0; pass < 2; pass++) {
                    uint16_t *w = pass ? got : ref;
                    const eoc_gate *nl = pass ? o : g;
                    const int cnt = pass ? (int)m : ng;
                    memset(w, 0, sizeof ref);
                    memcpy(w, in_mask, sizeof in_mask);
                    for (int k = 0; k < cnt; k++) {
                        const eoc_gate q = nl[k];
                        const int three = q.op == EOC_MUX || q.op == EOC_MAJ || q.op == EOC_XOR3, one = q.op == EOC_NOT || q.op == EOC_COPY;
                        const int zero = q.op == EOC_CONST0 || q.op == EOC_CONST1;
                        const uint16_t a = zero ? 0 : w[q.in0], b = (zero || one) ? 0 : w[q.in1], c = three ? w[q.in2] : 0;
                        uint16_t r;
                        switch (q.op) {
                        case EOC_NAND: r = (uint16_t)~(a & b); break;
                        case EOC_AND: r = a & b; break;
                        case EOC_OR: r = a | b; break;
                        case EOC_NOR: r = (uint16_t)~(a | b); break;
                        case EOC_XOR: r = a ^ b; break;
                        case EOC_XNOR: r = (uint16_t)~(a ^ b); break;
                        case EOC_ANDNY: r = (uint16_t)(~a & b); break;
                        case EOC_ANDYN: r = (uint16_t)(a & ~b); break;
                        case EOC_ORNY: r = (uint16_t)(~a | b); break;
                        case EOC_ORYN: r = (uint16_t)(a | ~b); break;
                        case EOC_MUX: r = (uint16_t)((a & b) | (~a & c)); break;
                        case EOC_NOT: r = (uint16_t)~a; break;
                        case EOC_COPY: r = a; break;
                        case EOC_CONST0: r = 0; break;
                        case EOC_CONST1: r = 0xFFFF; break;
                        case EOC_MAJ: r = (uint16_t)((a & b) | (a & c) | (b & c)); break;
                        default: r = a ^ b ^ c; break; /* EOC_XOR3 */
                        }
                        w[q.out] = r;
                    }
                }
                for (int k = 0; k < no_; k++) CHECK(ref[outs[k]] == got[outs[k]]);
                CHECK(eoc_circuit_bootstraps(o, (size_t)m) <= eoc_circuit_bootstraps(g, (size_t)ng));
                int64_t d_before = 0, d_after = 0;
                CHECK(eoc_netlist_levels(g, (size_t)ng, NULL, &d_before) >= 0 && eoc_netlist_levels(o, (size_t)m, NULL, &d_after) >= 0);
                CHECK(d_after <= d_before);
            }
        }
#undef RND
    }
    /* f2, server side: the cloud key alone as the process-global context (no secret in it) */
    size_t ck_len = eoc_cloud_key_blob_bytes(&p);
    void *ck = malloc(ck_len);
    CHECK(eoc_cloud_key_export(sk, ck, ck_len) == EOC_OK);
    {
        CHECK(eoc_global_key_mode() == 0 && eoc_global_cloud_key_export(NULL, 0) == 0);
        CHECK(eoc_global_import_cloud_key_blob(blob, need) == EOC_ERR_ARG);      /* a SECRET blob is refused */
        CHECK(eoc_global_import_cloud_key_blob(ck, ck_len - 1) == EOC_ERR_ARG);  /* truncated */
        CHECK(eoc_global_import_cloud_key_blob(ck, ck_len) == EOC_OK);
        CHECK(eoc_global_key_mode() == 2);
        CHECK(eoc_global_import_cloud_key_blob(ck, ck_len) == EOC_ERR_ARG);      /* one key per process */
        eoc_params q;
        CHECK(eoc_global_params(&q) == EOC_OK && q.n == p.n && q.l == p.l);
        CHECK(encryptBit(1, "") == NULL && exportSecretKey() == NULL && decryptBit("AAAA", "") == -1);
        CHECK(eoc_global_encrypt_bits(b0, 4, c0) == EOC_ERR_NO_KEY);
        void *back = malloc(ck_len);
        CHECK(eoc_global_cloud_key_export(back, ck_len) == ck_len && memcmp(back, ck, ck_len) == 0);
        free(back);
        const char *one = constantBit(1);                                         /* needs the parameters only */
        CHECK(one != NULL);
        free((void *)one);
        char path[] = "/tmp/eoc_abi_smoke_XXXXXX";
        int fd = mkstemp(path);
        CHECK(fd >= 0);
        close(fd);
        CHECK(exportCloudKeyToFile(path) == 0);
        resetGateKey();
        CHECK(eoc_global_key_mode() == 0 && importCloudKeyFromFile("/nonexistent/cloud.key") == -1);
        CHECK(importCloudKeyFromFile(path) == 0 && eoc_global_key_mode() == 2);
        remove(path);
        if (!gpu) resetGateKey();
    }
    if (!gpu) {
        free(ck);
        if (eoc_device_count() == 0) {
            CHECK(eoc_gate_batch(EOC_NAND, NULL, c0, c1, NULL, out, 4) == EOC_ERR_NO_DEVICE);
            eoc_engine *e = NULL;
            CHECK(eoc_engine_create(0, &p, &e) == EOC_ERR_NO_DEVICE && e == NULL);
        }
        printf("abi_smoke cpu OK (host threads %d)\n", eoc_host_threads());
        free(c0); free(c1); free(out); free(blob);
        eoc_secret_key_free(sk);
        return 0;
    }
    /* the cloud-key-only context evaluates gates: the engine comes up behind the imported key on first use */
    CHECK(eoc_global_gate_batch(EOC_NAND, NULL, c0, c1, NULL, out, 4) == EOC_OK);
    CHECK(eoc_decrypt_bits(sk, out, 4, dec) == EOC_OK);                           /* decrypted by the key's owner */
    for (int i = 0; i < 4; i++) CHECK(dec[i] == (uint8_t)(1 - (b0[i] & b1[i])));
    resetGateKey();
    free(ck);
    CHECK(eoc_gpu_init(0, &p) == EOC_OK);
    CHECK(eoc_upload_cloud_key(sk) == EOC_OK);
    CHECK(eoc_gate_batch(EOC_NAND, NULL, c0, c1, NULL, out, 4) == EOC_OK);
    CHECK(eoc_decrypt_bits(sk, out, 4, dec) == EOC_OK);
    for (int i = 0; i < 4; i++) CHECK(dec[i] == (uint8_t)(1 - (b0[i] & b1[i])));
    /* a two-gate netlist: w3 = XOR(w0, w1); w2 = MUX(w3, w0, w1) over 4 instances */
    eoc_gate gates[2] = {{EOC_XOR, 0, 1, -1, 3}, {EOC_MUX, 3, 0, 1, 2}};
    int32_t *wires = calloc(4 * 4 * st, 4);
    memcpy(wires, c0, 4 * st * 4);
    memcpy(wires + 4 * st, c1, 4 * st * 4);
    CHECK(eoc_circuit_run(gates, 2, wires, 4, 4) == EOC_OK);
    CHECK(eoc_decrypt_bits(sk, wires + 2 * 4 * st, 4, dec) == EOC_OK);
    for (int i = 0; i < 4; i++) CHECK(dec[i] == ((b0[i] ^ b1[i]) ? b0[i] : b1[i]));
    eoc_gpu_shutdown();
    /* string API, the reference's style */
    const char *tok = generateGateKey(80, 5);
    CHECK(tok != NULL);
    free((void *)tok);
    const char *e0 = encryptBit(0, ""), *e1 = encryptBit(1, "");
    const char *r = gateNAND(e1, e1, "");
    CHECK(r && decryptBit(r, "") == 0);
    free((void *)r);
    r = gateMUX(e0, e0, e1, "");
    CHECK(r && decryptBit(r, "") == 1);
    free((void *)r); free((void *)e0); free((void *)e1);
    resetGateKey();
    free(c0); free(c1); free(out); free(blob); free(wires);
    eoc_secret_key_free(sk);
    printf("abi_smoke gpu OK\n");
    return 0;
}
