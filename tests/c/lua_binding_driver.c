/* lua_binding_driver.c -- drives every l_* entry of integration/lua/eoc-tfhe-gate-bindings.c through the Lua C-API TEST
 * DOUBLE of tests/lua_double/ (NOT liblua: no Lua SDK exists in the build image).  What runs here is this repository's
 * binding C -- argument marshalling, length arithmetic, NULL -> nil, freeing of heap results -- on top of the real
 * libeoc_tfhe_gpu.so.  CPU legs only (client-side calls, cloud-key export / import, netlist helpers, every refusal);
 * built with ASan/UBSan by tests/test_lua_binding.py.  The GPU legs (gateBatch / circuitRun against the oracle) are
 * driven from Python through the same double (tests/test_lua_binding.py).
 *   usage: lua_binding_driver <file with base64(EOCSK1 secret key blob)> <scratch dir> */
#include "lauxlib.h"
#include "eoc_tfhe_gpu.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int luaopen_tfhe_gates(lua_State *L);

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "FAILED: %s (line %d); double says: %s\n", #c, __LINE__, ld_error(L)); return 1; } } while (0)

static lua_State *L;
static const luaL_Reg *g_regs;
static lua_CFunction fn[64];
static const char *fn_name[64];
static int n_fn;

static lua_CFunction F(const char *name)
{
    for (int k = 0; k < n_fn; k++)
        if (strcmp(fn_name[k], name) == 0) return fn[k];
    fprintf(stderr, "no such entry: %s\n", name);
    exit(2);
}
/* call with the values pushed since the last call; returns the number of results */
static int call(const char *name, int nargs) { return ld_call(L, F(name), nargs); }
static int is_nil(int idx) { return ld_type(L, idx) == LUA_TNIL; }
static char *dup_result(size_t *len)
{
    const void *p = ld_to_lstr(L, 1, len);
    if (!p) return NULL;
    char *c = malloc(*len + 1);
    memcpy(c, p, *len + 1);
    return c;
}

int main(int argc, char **argv)
{
    (void)g_regs;
    if (argc < 3) return 2;
    L = ld_new();
    /* ---- the module table: luaopen_tfhe_gates, like luaopen_tfhe (ao-tfhe/eoc-tfhe-bindings.c:128-148) ---- */
    CHECK(luaopen_tfhe_gates(L) == 1 && ld_type(L, 1) == LUA_TTABLE);
    n_fn = ld_table_size(L, 1);
    CHECK(n_fn == 34);
    for (int k = 0; k < n_fn; k++) {
        fn_name[k] = ld_table_name(L, 1, k);
        fn[k] = ld_table_get(L, 1, fn_name[k]);
        CHECK(fn[k] != NULL);
        for (int j = 0; j < k; j++) CHECK(strcmp(fn_name[j], fn_name[k]) != 0);
    }
    ld_settop0(L);

    /* ---- no key yet ---- */
    CHECK(call("keyMode", 0) == 1 && ld_to_int(L, 1) == 0);
    CHECK(call("sampleInts", 0) == 1 && ld_to_int(L, 1) == -1);
    ld_push_lstr(L, "\1\0\1", 3);
    CHECK(call("encryptBits", 1) == 1 && is_nil(1));
    CHECK(call("exportSecretKey", 0) == 1 && is_nil(1));           /* NULL -> nil, as lua_pushstring does (:21) */
    CHECK(call("exportCloudKey", 0) == 1 && is_nil(1));
    ld_push_int(L, 1);
    CHECK(call("encryptBit", 1) == 1 && is_nil(1));
    CHECK(call("deviceCount", 0) == 1 && ld_to_int(L, 1) >= 0);
    CHECK(call("engineCount", 0) == 1 && ld_to_int(L, 1) == 0);

    /* ---- argument checks: luaL_check* must raise, not crash ---- */
    ld_push_lstr(L, "x", 1);
    CHECK(call("encryptBit", 1) == -1 && strstr(ld_error(L), "bad argument #1"));
    CHECK(call("decryptBit", 0) == -1 && strstr(ld_error(L), "bad argument #1"));
    ld_push_int(L, 80);
    CHECK(call("generateGateKey", 1) == -1 && strstr(ld_error(L), "bad argument #2"));
    ld_push_nil(L);
    CHECK(call("importSecretKey", 1) == -1);
    ld_push_lstr(L, "a", 1);
    ld_push_nil(L);
    CHECK(call("gateNAND", 2) == -1 && strstr(ld_error(L), "bad argument #2"));

    /* ---- a seeded secret key set arrives as base64 (Tfhe.importSecretKey) ---- */
    FILE *f = fopen(argv[1], "rb");
    CHECK(f != NULL);
    fseek(f, 0, SEEK_END);
    long blen = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *b64 = malloc((size_t)blen + 1);
    CHECK(fread(b64, 1, (size_t)blen, f) == (size_t)blen);
    fclose(f);
    b64[blen] = 0;
    ld_push_lstr(L, "not base64 of a key", 19);
    CHECK(call("importSecretKey", 1) == 1 && ld_to_int(L, 1) == -1);
    ld_push_lstr(L, b64, (size_t)blen);
    CHECK(call("importSecretKey", 1) == 1 && ld_to_int(L, 1) == 0);
    ld_push_lstr(L, b64, (size_t)blen);
    CHECK(call("importSecretKey", 1) == 1 && ld_to_int(L, 1) == -1);     /* one key per process (eoc-tfhe-run.cpp:245-249) */
    CHECK(call("keyMode", 0) == 1 && ld_to_int(L, 1) == 1);
    CHECK(call("sampleInts", 0) == 1);
    const long long row_ints = ld_to_int(L, 1);
    CHECK(row_ints == 17);                                                /* the test key has n = 16 */
    const size_t row = (size_t)row_ints * 4;

    /* ---- raw batches: encryptBits / decryptBits ---- */
    const char bits[9] = {1, 0, 0, 1, 1, 1, 0, 1, 0};
    ld_push_lstr(L, bits, 9);
    CHECK(call("encryptBits", 1) == 1);
    size_t ct_len = 0;
    char *cts = dup_result(&ct_len);
    CHECK(cts && ct_len == 9 * row);
    ld_push_lstr(L, cts, ct_len);
    CHECK(call("decryptBits", 1) == 1);
    size_t dl = 0;
    const void *dec = ld_to_lstr(L, 1, &dl);
    CHECK(dec && dl == 9 && memcmp(dec, bits, 9) == 0);
    ld_push_lstr(L, cts, ct_len - 1);                                     /* not a whole number of samples */
    CHECK(call("decryptBits", 1) == 1 && is_nil(1));
    ld_push_lstr(L, "", 0);                                               /* zero bits: an empty string, not nil */
    CHECK(call("encryptBits", 1) == 1 && ld_to_lstr(L, 1, &dl) && dl == 0);

    /* ---- string API: encryptBit / decryptBit / constantBit ---- */
    char *one = NULL, *zero = NULL;
    size_t l1 = 0, l0 = 0;
    ld_push_int(L, 1);
    ld_push_lstr(L, "ignored key", 11);                                   /* ignored, as :63 passes NULL */
    CHECK(call("encryptBit", 2) == 1 && (one = dup_result(&l1)) && l1 > row);
    ld_push_int(L, 0);
    CHECK(call("encryptBit", 1) == 1 && (zero = dup_result(&l0)));
    ld_push_lstr(L, one, l1);
    CHECK(call("decryptBit", 1) == 1 && ld_to_int(L, 1) == 1);
    ld_push_lstr(L, zero, l0);
    CHECK(call("decryptBit", 1) == 1 && ld_to_int(L, 1) == 0);
    ld_push_lstr(L, "AAAA", 4);
    CHECK(call("decryptBit", 1) == 1 && ld_to_int(L, 1) == -1);           /* malformed: -1, like decryptInteger (:397) */
    ld_push_int(L, 1);
    CHECK(call("constantBit", 1) == 1);
    size_t lc = 0;
    char *cst = dup_result(&lc);
    CHECK(cst != NULL);
    ld_push_lstr(L, cst, lc);
    CHECK(call("decryptBit", 1) == 1 && ld_to_int(L, 1) == 1);

    /* ---- the gates need a GPU engine: on a box without a device every one answers nil (no CPU fallback) ---- */
    if (eoc_device_count() == 0) {
        static const char *g2[] = {"gateNAND", "gateAND", "gateOR", "gateNOR", "gateXOR", "gateXNOR"};
        for (int k = 0; k < 6; k++) {
            ld_push_lstr(L, one, l1);
            ld_push_lstr(L, zero, l0);
            CHECK(call(g2[k], 2) == 1 && is_nil(1));
        }
        ld_push_lstr(L, one, l1);
        CHECK(call("gateNOT", 1) == 1 && is_nil(1));
        ld_push_lstr(L, one, l1);
        ld_push_lstr(L, zero, l0);
        ld_push_lstr(L, one, l1);
        ld_push_lstr(L, "pk", 2);
        CHECK(call("gateMUX", 4) == 1 && is_nil(1));
        ld_push_lstr(L, one, l1);                                          /* the extension gates: three ciphertexts, like gateMUX */
        ld_push_lstr(L, zero, l0);
        ld_push_lstr(L, one, l1);
        CHECK(call("gateMAJ", 3) == 1 && is_nil(1));
        ld_push_lstr(L, one, l1);
        ld_push_lstr(L, zero, l0);
        ld_push_lstr(L, one, l1);
        CHECK(call("gateXOR3", 3) == 1 && is_nil(1));
        ld_push_int(L, 0);
        ld_push_lstr(L, cts, ct_len);
        ld_push_lstr(L, cts, ct_len);
        CHECK(call("gateBatch", 3) == 1 && is_nil(1));
    }

    /* ---- gateBatch's length arithmetic (refused before anything reaches the engine) ---- */
    ld_push_int(L, 0);
    ld_push_nil(L);
    ld_push_lstr(L, cts, ct_len);
    CHECK(call("gateBatch", 3) == 1 && is_nil(1));                        /* in0 missing */
    ld_push_int(L, 0);
    ld_push_lstr(L, cts, ct_len);
    ld_push_lstr(L, cts, ct_len - row);
    CHECK(call("gateBatch", 3) == 1 && is_nil(1));                        /* in1 shorter than in0 */
    ld_push_int(L, 0);
    ld_push_lstr(L, cts, ct_len - 2);
    ld_push_lstr(L, cts, ct_len - 2);
    CHECK(call("gateBatch", 3) == 1 && is_nil(1));                        /* not a whole number of samples */
    ld_push_int(L, 10);
    ld_push_lstr(L, cts, ct_len);
    ld_push_lstr(L, cts, ct_len);
    ld_push_lstr(L, cts, ct_len);
    ld_push_lstr(L, "\0\4\12", 3);
    CHECK(call("gateBatch", 5) == 1 && is_nil(1));                        /* 3 opcodes for 9 gates */
    ld_push_lstr(L, "op", 2);
    ld_push_lstr(L, cts, ct_len);
    CHECK(call("gateBatch", 2) == -1);                                    /* op must be an integer */

    /* ---- netlist helpers (host side): NOT s; (s & x) | (~s & y)  ->  one MUX ---- */
    const eoc_gate nl[4] = {{EOC_NOT, 0, -1, -1, 3}, {EOC_AND, 0, 1, -1, 4}, {EOC_AND, 3, 2, -1, 5}, {EOC_OR, 4, 5, -1, 6}};
    const int32_t outs[1] = {6};
    ld_push_lstr(L, nl, sizeof nl);
    CHECK(call("circuitBootstraps", 1) == 1 && ld_to_int(L, 1) == 3);
    ld_push_lstr(L, nl, sizeof nl - 1);
    CHECK(call("circuitBootstraps", 1) == 1 && ld_to_int(L, 1) == -1);
    /* netlistCost / netlistDepth (round 6): the levels are the engine's -- NOT is free and sits in the pre-pass of its reader's
     * level, so AND(s, x) and AND(NOT s, y) share a level and OR is the second: 2 x 18 units (0.1 ms) for 3 instances; at
     * 1024 instances the first level is two full launches (60), the second one (30) */
    ld_push_lstr(L, nl, sizeof nl);
    CHECK(call("netlistDepth", 1) == 1 && ld_to_int(L, 1) == 2);
    ld_push_lstr(L, nl, sizeof nl - 2);
    CHECK(call("netlistDepth", 1) == 1 && ld_to_int(L, 1) == -1);
    ld_push_lstr(L, nl, sizeof nl);
    ld_push_int(L, 3);
    CHECK(call("netlistCost", 2) == 1 && ld_to_int(L, 1) == 36);
    ld_push_lstr(L, nl, sizeof nl);
    ld_push_int(L, 1024);
    CHECK(call("netlistCost", 2) == 1 && ld_to_int(L, 1) == 90);
    ld_push_lstr(L, nl, sizeof nl);
    ld_push_int(L, -1);
    CHECK(call("netlistCost", 2) == 1 && ld_to_int(L, 1) == -1);
    ld_push_lstr(L, nl, sizeof nl);
    ld_push_lstr(L, outs, sizeof outs);
    CHECK(call("netlistOptimize", 2) == 1);
    size_t ol = 0;
    const eoc_gate *opt = ld_to_lstr(L, 1, &ol);
    CHECK(opt && ol == sizeof(eoc_gate) && opt->op == EOC_MUX && opt->in0 == 0 && opt->in1 == 1 && opt->in2 == 2 && opt->out == 6);
    ld_push_lstr(L, nl, sizeof nl - 4);
    ld_push_lstr(L, outs, sizeof outs);
    CHECK(call("netlistOptimize", 2) == 1 && is_nil(1));
    ld_push_lstr(L, nl, sizeof nl);
    ld_push_lstr(L, outs, 3);
    CHECK(call("netlistOptimize", 2) == 1 && is_nil(1));
    /* circuitRun: wires must be nWires x instances x (n + 1) int32 */
    ld_push_lstr(L, nl, sizeof nl);
    ld_push_lstr(L, cts, ct_len);
    ld_push_int(L, 7);
    ld_push_int(L, 2);
    CHECK(call("circuitRun", 4) == 1 && is_nil(1));                       /* 9 samples are not 7 x 2 */
    ld_push_lstr(L, nl, sizeof nl - 3);
    ld_push_lstr(L, cts, ct_len);
    ld_push_int(L, 9);
    ld_push_int(L, 1);
    CHECK(call("circuitRun", 4) == 1 && is_nil(1));                       /* broken netlist bytes */

    /* ---- key export; the cloud key leaves for a server ---- */
    CHECK(call("exportSecretKey", 0) == 1);
    size_t sl = 0;
    const char *skb = ld_to_lstr(L, 1, &sl);
    CHECK(skb && sl == (size_t)blen && memcmp(skb, b64, sl) == 0);        /* seeded keys re-export byte for byte */
    char path[512], bad[512];
    snprintf(path, sizeof path, "%s/cloud.key", argv[2]);
    snprintf(bad, sizeof bad, "%s/no-such-dir/cloud.key", argv[2]);
    ld_push_lstr(L, path, strlen(path));
    CHECK(call("exportCloudKeyToFile", 1) == 1 && ld_to_int(L, 1) == 0);
    ld_push_lstr(L, bad, strlen(bad));
    CHECK(call("exportCloudKeyToFile", 1) == 1 && ld_to_int(L, 1) != 0);
    CHECK(call("exportCloudKey", 0) == 1);
    size_t ckl = 0;
    char *ck = dup_result(&ckl);
    CHECK(ck && ckl > 100000 && strncmp(ck, "RU9DQ0sx", 8) == 0);          /* base64("EOCCK1") */

    /* ---- server side: cloud key only ---- */
    CHECK(call("resetGateKey", 0) == 0);
    CHECK(call("keyMode", 0) == 1 && ld_to_int(L, 1) == 0);
    ld_push_lstr(L, b64, (size_t)blen);
    CHECK(call("importCloudKey", 1) == 1 && ld_to_int(L, 1) != 0);        /* a SECRET blob is refused */
    ld_push_lstr(L, ck, ckl);
    CHECK(call("importCloudKey", 1) == 1 && ld_to_int(L, 1) == 0);
    CHECK(call("keyMode", 0) == 1 && ld_to_int(L, 1) == 2);
    CHECK(call("sampleInts", 0) == 1 && ld_to_int(L, 1) == row_ints);
    ld_push_lstr(L, bits, 9);
    CHECK(call("encryptBits", 1) == 1 && is_nil(1));                      /* no secret here */
    ld_push_lstr(L, cts, ct_len);
    CHECK(call("decryptBits", 1) == 1 && is_nil(1));
    ld_push_lstr(L, one, l1);
    CHECK(call("decryptBit", 1) == 1 && ld_to_int(L, 1) == -1);
    CHECK(call("exportSecretKey", 0) == 1 && is_nil(1));
    ld_push_int(L, 0);
    CHECK(call("constantBit", 1) == 1 && ld_to_lstr(L, 1, &dl));          /* needs no key at all */
    CHECK(call("resetGateKey", 0) == 0);
    ld_push_lstr(L, path, strlen(path));
    CHECK(call("importCloudKeyFromFile", 1) == 1 && ld_to_int(L, 1) == 0);
    CHECK(call("keyMode", 0) == 1 && ld_to_int(L, 1) == 2);
    ld_push_lstr(L, bad, strlen(bad));
    CHECK(call("importCloudKeyFromFile", 1) == 1 && ld_to_int(L, 1) != 0);
    CHECK(call("resetGateKey", 0) == 0);

    /* ---- setDevices: integers only; no arguments = forget ---- */
    ld_push_int(L, 0);
    ld_push_lstr(L, "1", 1);
    CHECK(call("setDevices", 2) == -1 && strstr(ld_error(L), "bad argument #2"));
    CHECK(call("setDevices", 0) == 1 && ld_to_int(L, 1) == 0);

    ld_settop0(L);
    CHECK(ld_live_bytes(L) == 0);
    ld_close(L);
    free(b64); free(cts); free(one); free(zero); free(cst); free(ck);
    printf("lua_binding_driver OK: %d entries driven through the Lua C-API test double\n", n_fn);
    return 0;
}
