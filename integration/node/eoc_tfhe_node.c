/* eoc_tfhe_node.c -- N-API addon: the Node.js host binding of libeoc_tfhe_gpu.so.
 *
 * Plays the role ao-tfhe/eoc-tfhe-bindings.c plays for Lua (l_* wrappers + luaopen_tfhe's table,
 * :12-148): every export reads its arguments, calls the C ABI of include/eoc_tfhe_gpu.h, converts the
 * result and releases heap strings with free() (as the Lua binding does, :21).  NULL results become
 * JS null (Lua nil in the reference).  On the GPU box this replaces the wasm `handle` path of
 * tests/tfhe.test.js:52,74.  Build: see integration/node/build.sh (gcc, headers from /usr/include/node).
 */
#include <node_api.h>
#include <stdbool.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "eoc_tfhe_gpu.h"

#define NAPI_OK(call) do { if ((call) != napi_ok) { napi_throw_error(env, NULL, "N-API call failed: " #call); return NULL; } } while (0)

static char *arg_string(napi_env env, napi_value v)
{ /* malloc'ed copy of a JS string (or "" for undefined/null, like luaL_optstring) */
    napi_valuetype t;
    if (napi_typeof(env, v, &t) != napi_ok || t != napi_string) return strdup("");
    size_t len = 0;
    napi_get_value_string_utf8(env, v, NULL, 0, &len);
    char *s = malloc(len + 1);
    napi_get_value_string_utf8(env, v, s, len + 1, &len);
    return s;
}
static napi_value ret_string(napi_env env, const char *s)
{ /* copy into a JS string and free() the library's buffer; NULL -> null */
    napi_value out;
    if (!s) {
        napi_get_null(env, &out);
        return out;
    }
    napi_create_string_utf8(env, s, NAPI_AUTO_LENGTH, &out);
    free((void *)s);
    return out;
}
static napi_value ret_int(napi_env env, int v)
{
    napi_value out;
    napi_create_int32(env, v, &out);
    return out;
}
static int arg_int(napi_env env, napi_value v)
{
    int32_t x = 0;
    napi_get_value_int32(env, v, &x);
    return x;
}
#define ARGS(N)                                           \
    size_t argc = N;                                      \
    napi_value argv[N ? N : 1];                           \
    for (int _i = 0; _i < (N ? N : 1); _i++) argv[_i] = NULL; \
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL))

/* ---- the reference's 11 functions (names of luaopen_tfhe's table, eoc-tfhe-bindings.c:130-144) ---- */
static napi_value n_info(napi_env env, napi_callback_info info_) { (void)info_; info(); return ret_int(env, 0); }
static napi_value n_testJWT(napi_env env, napi_callback_info info_) { (void)info_; testJWT(); return ret_int(env, 0); }
static napi_value n_generateSecretKey(napi_env env, napi_callback_info info)
{
    ARGS(2);
    char *jwt = arg_string(env, argv[0]), *jwks = arg_string(env, argv[1]);
    const char *r = generateSecretKey(jwt, jwks);
    free(jwt); free(jwks);
    return ret_string(env, r);
}
static napi_value n_generatePublicKey(napi_env env, napi_callback_info info)
{
    (void)info;
    return ret_string(env, generatePublicKey());
}
static napi_value n_encryptInteger(napi_env env, napi_callback_info info)
{
    ARGS(2);
    return ret_string(env, encryptInteger(arg_int(env, argv[0]), NULL)); /* key ignored, as :63 passes NULL */
}
static napi_value n_encryptInteger_dummy(napi_env env, napi_callback_info info)
{
    ARGS(2);
    return ret_string(env, encryptInteger(arg_int(env, argv[0]), NULL)); /* :73 calls encryptInteger too */
}
static napi_value n_decryptInteger(napi_env env, napi_callback_info info)
{
    ARGS(4);
    char *ct = arg_string(env, argv[0]), *jwt = arg_string(env, argv[2]), *jwks = arg_string(env, argv[3]);
    int v = decryptInteger(ct, NULL, jwt, jwks);
    free(ct); free(jwt); free(jwks);
    return ret_int(env, v);
}
static napi_value n_binary(napi_env env, napi_callback_info info,
                           const char *(*fn)(const char *, const char *, const char *))
{
    ARGS(3);
    char *a = arg_string(env, argv[0]), *b = arg_string(env, argv[1]), *pk = arg_string(env, argv[2]);
    const char *r = fn(a, b, pk);
    free(a); free(b); free(pk);
    return ret_string(env, r);
}
static napi_value n_addCiphertexts(napi_env env, napi_callback_info info) { return n_binary(env, info, addCiphertexts); }
static napi_value n_subtractCiphertexts(napi_env env, napi_callback_info info) { return n_binary(env, info, subtractCiphertexts); }
static napi_value n_encryptASCIIString(napi_env env, napi_callback_info info)
{
    ARGS(3);
    char *s = arg_string(env, argv[0]);
    const char *r = encrypt8BitASCIIString(s, (int16_t)arg_int(env, argv[1]), NULL);
    free(s);
    return ret_string(env, r);
}
static napi_value n_decryptASCIIString(napi_env env, napi_callback_info info)
{
    ARGS(5);
    char *ct = arg_string(env, argv[0]), *jwt = arg_string(env, argv[3]), *jwks = arg_string(env, argv[4]);
    const char *r = decrypt8BitASCIIString(ct, (int16_t)arg_int(env, argv[1]), NULL, jwt, jwks);
    free(ct); free(jwt); free(jwks);
    return ret_string(env, r);
}

/* ---- Boolean path ---- */
static napi_value n_generateGateKey(napi_env env, napi_callback_info info)
{
    ARGS(2);
    int64_t seed = 0;
    napi_get_value_int64(env, argv[1], &seed);
    return ret_string(env, generateGateKey(arg_int(env, argv[0]), (uint64_t)seed));
}
static napi_value n_resetGateKey(napi_env env, napi_callback_info info) { (void)info; resetGateKey(); return ret_int(env, 0); }
static napi_value n_encryptBit(napi_env env, napi_callback_info info)
{
    ARGS(2);
    return ret_string(env, encryptBit(arg_int(env, argv[0]), NULL));
}
static napi_value n_decryptBit(napi_env env, napi_callback_info info)
{
    ARGS(2);
    char *ct = arg_string(env, argv[0]);
    int v = decryptBit(ct, NULL);
    free(ct);
    return ret_int(env, v);
}
#define GATE2(NAME) static napi_value n_##NAME(napi_env env, napi_callback_info info) { return n_binary(env, info, NAME); }
GATE2(gateNAND) GATE2(gateAND) GATE2(gateOR) GATE2(gateNOR) GATE2(gateXOR) GATE2(gateXNOR)
static napi_value n_gateNOT(napi_env env, napi_callback_info info)
{
    ARGS(2);
    char *a = arg_string(env, argv[0]);
    const char *r = gateNOT(a, "");
    free(a);
    return ret_string(env, r);
}
static napi_value n_gateMUX(napi_env env, napi_callback_info info)
{
    ARGS(4);
    char *a = arg_string(env, argv[0]), *b = arg_string(env, argv[1]), *c = arg_string(env, argv[2]);
    const char *r = gateMUX(a, b, c, "");
    free(a); free(b); free(c);
    return ret_string(env, r);
}
/* the extension gates (EOC_MAJ, EOC_XOR3): three ciphertexts in, one out, like gateMUX */
static napi_value n_gateMAJ(napi_env env, napi_callback_info info)
{
    ARGS(4);
    char *a = arg_string(env, argv[0]), *b = arg_string(env, argv[1]), *c = arg_string(env, argv[2]);
    const char *r = gateMAJ(a, b, c, "");
    free(a); free(b); free(c);
    return ret_string(env, r);
}
static napi_value n_gateXOR3(napi_env env, napi_callback_info info)
{
    ARGS(4);
    char *a = arg_string(env, argv[0]), *b = arg_string(env, argv[1]), *c = arg_string(env, argv[2]);
    const char *r = gateXOR3(a, b, c, "");
    free(a); free(b); free(c);
    return ret_string(env, r);
}
static napi_value n_exportSecretKey(napi_env env, napi_callback_info info) { (void)info; return ret_string(env, exportSecretKey()); }
static napi_value n_importSecretKey(napi_env env, napi_callback_info info)
{
    ARGS(1);
    char *k = arg_string(env, argv[0]);
    int v = importSecretKey(k);
    free(k);
    return ret_int(env, v);
}
/* the cloud ("public") key: exported by the client, installed by a server that never holds the secret key
 * (eoc-tfhe-run.cpp:232-234 aliases it as globalPublicKey; generatePublicKey, eoc-tfhe-run.h:10, is the same export) */
static napi_value n_exportCloudKey(napi_env env, napi_callback_info info) { (void)info; return ret_string(env, exportCloudKey()); }
static napi_value n_importCloudKey(napi_env env, napi_callback_info info)
{
    ARGS(1);
    char *k = arg_string(env, argv[0]);
    int v = importCloudKey(k);
    free(k);
    return ret_int(env, v);
}
static napi_value n_exportCloudKeyToFile(napi_env env, napi_callback_info info)
{
    ARGS(1);
    char *path = arg_string(env, argv[0]);
    int v = exportCloudKeyToFile(path);
    free(path);
    return ret_int(env, v);
}
static napi_value n_importCloudKeyFromFile(napi_env env, napi_callback_info info)
{
    ARGS(1);
    char *path = arg_string(env, argv[0]);
    int v = importCloudKeyFromFile(path);
    free(path);
    return ret_int(env, v);
}
static napi_value n_keyMode(napi_env env, napi_callback_info info) { (void)info; return ret_int(env, eoc_global_key_mode()); }

/* ---- raw-buffer batch calls on the global key: Buffers of bytes / int32 LWE samples ---- */
static napi_value n_sampleInts(napi_env env, napi_callback_info info)
{ /* n + 1: int32 per LWE sample of the global key, or -1 */
    (void)info;
    eoc_params p;
    return ret_int(env, eoc_global_params(&p) == EOC_OK ? p.n + 1 : -1);
}
static napi_value n_constantBit(napi_env env, napi_callback_info info)
{
    size_t argc = 1;
    napi_value argv[1];
    napi_get_cb_info(env, info, &argc, argv, NULL, NULL);
    return ret_string(env, constantBit(arg_int(env, argv[0])));
}
static napi_value n_encryptBits(napi_env env, napi_callback_info info)
{ /* (Buffer bits[count]) -> Buffer int32[count][n+1] */
    ARGS(1);
    void *bits; size_t count;
    NAPI_OK(napi_get_buffer_info(env, argv[0], &bits, &count));
    eoc_params p;
    napi_value out;
    if (eoc_global_params(&p) != EOC_OK) { napi_get_null(env, &out); return out; }
    void *cts;
    NAPI_OK(napi_create_buffer(env, count * (size_t)(p.n + 1) * 4, &cts, &out));
    if (eoc_global_encrypt_bits((const uint8_t *)bits, count, (int32_t *)cts) != EOC_OK) napi_get_null(env, &out);
    return out;
}
static napi_value n_decryptBits(napi_env env, napi_callback_info info)
{ /* (Buffer int32[count][n+1]) -> Buffer bits[count] */
    ARGS(1);
    void *cts; size_t bytes;
    NAPI_OK(napi_get_buffer_info(env, argv[0], &cts, &bytes));
    eoc_params p;
    napi_value out;
    if (eoc_global_params(&p) != EOC_OK) { napi_get_null(env, &out); return out; }
    size_t count = bytes / ((size_t)(p.n + 1) * 4);
    void *bits;
    NAPI_OK(napi_create_buffer(env, count, &bits, &out));
    if (eoc_global_decrypt_bits((const int32_t *)cts, count, (uint8_t *)bits) != EOC_OK) napi_get_null(env, &out);
    return out;
}
static napi_value ret_null(napi_env env)
{
    napi_value out;
    napi_get_null(env, &out);
    return out;
}
static int get_buffer(napi_env env, napi_value v, void **p, size_t *bytes)
{ /* 1 if v is a Buffer (pointer and length returned), 0 otherwise */
    bool isbuf = false;
    *p = NULL;
    *bytes = 0;
    if (!v || napi_is_buffer(env, v, &isbuf) != napi_ok || !isbuf) return 0;
    return napi_get_buffer_info(env, v, p, bytes) == napi_ok;
}
static int get_i32_array(napi_env env, napi_value v, int32_t **p, size_t *len)
{ /* 1 if v is an Int32Array */
    bool ista = false;
    *p = NULL;
    *len = 0;
    if (!v || napi_is_typedarray(env, v, &ista) != napi_ok || !ista) return 0;
    napi_typedarray_type t;
    void *data;
    napi_value ab;
    size_t off;
    if (napi_get_typedarray_info(env, v, &t, len, &data, &ab, &off) != napi_ok || t != napi_int32_array) return 0;
    *p = (int32_t *)data;
    return 1;
}
static napi_value n_gateBatch(napi_env env, napi_callback_info info)
{ /* (op, Buffer in0 | null, Buffer in1 | null, Buffer in2 | null, Buffer ops | null, count?) -> Buffer out, or null on
   * error (no GPU, a supplied operand shorter than in0, length not a multiple of a sample, ...).  `ops` = one opcode
   * byte per gate (mixed batch).  bootsCONSTANT (op 13 / 14) takes no operand: pass null and the gate count. */
    ARGS(6);
    int op = arg_int(env, argv[0]);
    void *in[3] = {NULL, NULL, NULL}, *ops = NULL;
    size_t bytes[3] = {0, 0, 0}, nops = 0;
    for (int k = 0; k < 3; k++) get_buffer(env, argv[1 + k], &in[k], &bytes[k]);
    get_buffer(env, argv[4], &ops, &nops);
    eoc_params p;
    if (eoc_global_params(&p) != EOC_OK) return ret_null(env);
    const size_t row = (size_t)(p.n + 1) * 4;
    size_t count;
    if (in[0]) {
        if (bytes[0] % row) return ret_null(env);
        count = bytes[0] / row;
    } else {
        if (ops || (op != EOC_CONST0 && op != EOC_CONST1)) return ret_null(env);
        count = (size_t)arg_int(env, argv[5]);
    }
    for (int k = 1; k < 3; k++)
        if (in[k] && bytes[k] != count * row) return ret_null(env); /* every supplied operand has in0's length */
    if (ops && nops != count) return ret_null(env);
    void *o;
    napi_value out;
    NAPI_OK(napi_create_buffer(env, count * row, &o, &out));
    if (eoc_global_gate_batch(op, (const uint8_t *)ops, (const int32_t *)in[0], (const int32_t *)in[1],
                              (const int32_t *)in[2], (int32_t *)o, count) != EOC_OK)
        return ret_null(env);
    return out;
}
/* ---- asynchronous batches on pinned buffers (eoc_gate_batch_submit / _wait): a Node host keeps two batches in flight
 * and the PCIe time of one disappears behind the kernels of the other ---- */
static void free_pinned(napi_env env, void *data, void *hint)
{
    (void)env;
    (void)hint;
    eoc_host_free(data);
}
static napi_value n_hostAlloc(napi_env env, napi_callback_info info)
{ /* (bytes) -> Buffer over pinned, device-mapped host memory (eoc_host_alloc); freed when the Buffer is collected */
    ARGS(1);
    int64_t bytes = 0;
    napi_get_value_int64(env, argv[0], &bytes);
    if (bytes <= 0) return ret_null(env);
    void *p = eoc_host_alloc((size_t)bytes);
    if (!p) return ret_null(env);
    napi_value out;
    if (napi_create_external_buffer(env, (size_t)bytes, p, free_pinned, NULL, &out) != napi_ok) {
        eoc_host_free(p);
        return ret_null(env);
    }
    return out;
}
static napi_ref g_held[2][5];      /* Buffers of the (at most two) submissions in flight */
static uint64_t g_held_ticket[2];
static void release_held(napi_env env, int slot)
{
    for (int k = 0; k < 5; k++)
        if (g_held[slot][k]) {
            napi_delete_reference(env, g_held[slot][k]);
            g_held[slot][k] = NULL;
        }
    g_held_ticket[slot] = 0;
}
static napi_value n_gateBatchSubmit(napi_env env, napi_callback_info info)
{ /* (op, in0, in1 | null, in2 | null, ops | null, out) -- every Buffer from hostAlloc -> ticket (number), or null.
   * The Buffers must stay referenced and untouched until gateBatchWait(ticket). */
    ARGS(6);
    int op = arg_int(env, argv[0]);
    void *in[3] = {NULL, NULL, NULL}, *ops = NULL, *out = NULL;
    size_t bytes[3] = {0, 0, 0}, nops = 0, obytes = 0;
    for (int k = 0; k < 3; k++) get_buffer(env, argv[1 + k], &in[k], &bytes[k]);
    get_buffer(env, argv[4], &ops, &nops);
    if (!get_buffer(env, argv[5], &out, &obytes) || !in[0]) return ret_null(env);
    eoc_params p;
    if (eoc_global_params(&p) != EOC_OK) return ret_null(env);
    const size_t row = (size_t)(p.n + 1) * 4;
    if (bytes[0] % row || obytes != bytes[0]) return ret_null(env);
    const size_t count = bytes[0] / row;
    for (int k = 1; k < 3; k++)
        if (in[k] && bytes[k] != bytes[0]) return ret_null(env);
    if (ops && nops != count) return ret_null(env);
    uint64_t ticket = 0;
    if (eoc_global_gate_batch_submit(op, (const uint8_t *)ops, (const int32_t *)in[0], (const int32_t *)in[1],
                                     (const int32_t *)in[2], (int32_t *)out, count, &ticket) != EOC_OK)
        return ret_null(env);
    /* the pinned Buffers stay referenced until gateBatchWait(ticket): a collected Buffer would run free_pinned
     * (eoc_host_free) under a DMA in flight.  Slot ticket & 1 is free again: the library completed its previous
     * occupant (ticket - 2) before it accepted this submission. */
    release_held(env, (int)(ticket & 1));
    g_held_ticket[ticket & 1] = ticket;
    const napi_value keep[5] = {argv[1], argv[2], argv[3], argv[4], argv[5]};
    for (int k = 0; k < 5; k++) {
        void *q; size_t qb;
        if (get_buffer(env, keep[k], &q, &qb)) napi_create_reference(env, keep[k], 1, &g_held[ticket & 1][k]);
    }
    napi_value t;
    NAPI_OK(napi_create_double(env, (double)ticket, &t));
    return t;
}
static napi_value n_gateBatchWait(napi_env env, napi_callback_info info)
{ /* (ticket) -> 0, or a negative error code */
    ARGS(1);
    double t = 0;
    napi_get_value_double(env, argv[0], &t);
    const int rc = eoc_gate_batch_wait((uint64_t)t);
    if (g_held_ticket[(uint64_t)t & 1] == (uint64_t)t) release_held(env, (int)((uint64_t)t & 1));
    return ret_int(env, rc);
}
static napi_value n_circuitRun(napi_env env, napi_callback_info info)
{ /* (Int32Array gates [5 per gate: op, in0, in1, in2, out], Buffer wires [nWires][instances][n+1], nWires, instances)
   * -> the same Buffer, evaluated in place (eoc_global_circuit_run), or null on error */
    ARGS(4);
    int32_t *g;
    size_t glen, wbytes;
    void *w;
    eoc_params p;
    if (!get_i32_array(env, argv[0], &g, &glen) || glen % 5 || !get_buffer(env, argv[1], &w, &wbytes) ||
        eoc_global_params(&p) != EOC_OK)
        return ret_null(env);
    const size_t n_wires = (size_t)arg_int(env, argv[2]), inst = (size_t)arg_int(env, argv[3]);
    if (wbytes != n_wires * inst * (size_t)(p.n + 1) * 4) return ret_null(env);
    if (eoc_global_circuit_run((const eoc_gate *)g, glen / 5, (int32_t *)w, n_wires, inst) != EOC_OK) return ret_null(env);
    return argv[1];
}
static napi_value n_netlistOptimize(napi_env env, napi_callback_info info)
{ /* (Int32Array gates, Int32Array outputs) -> Int32Array gates (NOT folding, MUX fusion, dead gates dropped) or null */
    ARGS(2);
    int32_t *g, *outs;
    size_t glen, nout;
    if (!get_i32_array(env, argv[0], &g, &glen) || glen % 5 || !get_i32_array(env, argv[1], &outs, &nout)) return ret_null(env);
    eoc_gate *tmp = malloc((glen / 5 + 1) * sizeof(eoc_gate));
    int64_t n = eoc_netlist_optimize((const eoc_gate *)g, glen / 5, outs, nout, tmp);
    if (n < 0) {
        free(tmp);
        return ret_null(env);
    }
    napi_value ab, ta;
    void *data;
    NAPI_OK(napi_create_arraybuffer(env, (size_t)n * sizeof(eoc_gate), &data, &ab));
    memcpy(data, tmp, (size_t)n * sizeof(eoc_gate));
    free(tmp);
    NAPI_OK(napi_create_typedarray(env, napi_int32_array, (size_t)n * 5, ab, 0, &ta));
    return ta;
}
static napi_value n_circuitBootstraps(napi_env env, napi_callback_info info)
{
    ARGS(1);
    int32_t *g;
    size_t glen;
    if (!get_i32_array(env, argv[0], &g, &glen) || glen % 5) return ret_int(env, -1);
    return ret_int(env, (int)eoc_circuit_bootstraps((const eoc_gate *)g, glen / 5));
}
static napi_value n_netlistCost(napi_env env, napi_callback_info info)
{ /* (Int32Array gates, instances) -> estimated run time in 0.1 ms units (eoc_netlist_cost), or -1 */
    ARGS(2);
    int32_t *g, inst;
    size_t glen;
    if (!get_i32_array(env, argv[0], &g, &glen) || glen % 5 || napi_get_value_int32(env, argv[1], &inst) != napi_ok || inst < 0)
        return ret_int(env, -1);
    int64_t c = eoc_netlist_cost((const eoc_gate *)g, glen / 5, (size_t)inst, 0);
    return ret_int(env, c > 0x7fffffff ? 0x7fffffff : (int)c);
}
static napi_value n_netlistDepth(napi_env env, napi_callback_info info)
{ /* (Int32Array gates) -> dependent levels that hold a blind rotation, or -1 */
    ARGS(1);
    int32_t *g;
    size_t glen;
    int64_t depth = -1;
    if (!get_i32_array(env, argv[0], &g, &glen) || glen % 5 || eoc_netlist_levels((const eoc_gate *)g, glen / 5, NULL, &depth) < 0)
        return ret_int(env, -1);
    return ret_int(env, (int)depth);
}
static napi_value n_engineCount(napi_env env, napi_callback_info info) { (void)info; return ret_int(env, eoc_gpu_engine_count()); }
static napi_value n_setDevices(napi_env env, napi_callback_info info)
{ /* (Int32Array devices) -> 0 or a negative code: the devices the next generateGateKey / importCloudKey brings up, one
   * engine each (a device may repeat; an empty array returns to EOC_TFHE_DEVICES / device 0) */
    ARGS(1);
    int32_t *d;
    size_t n;
    if (!get_i32_array(env, argv[0], &d, &n)) return ret_int(env, EOC_ERR_ARG);
    return ret_int(env, eoc_gpu_set_devices((const int *)d, (int)n));
}
static napi_value n_deviceCount(napi_env env, napi_callback_info info) { (void)info; return ret_int(env, eoc_device_count()); }

static napi_value init(napi_env env, napi_value exports)
{
    static const struct { const char *name; napi_callback fn; } tab[] = {
        {"info", n_info}, {"testJWT", n_testJWT}, {"generateSecretKey", n_generateSecretKey},
        {"generatePublicKey", n_generatePublicKey}, {"encryptInteger", n_encryptInteger},
        {"encryptInteger_dummy", n_encryptInteger_dummy}, {"decryptInteger", n_decryptInteger},
        {"addCiphertexts", n_addCiphertexts}, {"subtractCiphertexts", n_subtractCiphertexts},
        {"encryptASCIIString", n_encryptASCIIString}, {"decryptASCIIString", n_decryptASCIIString},
        {"generateGateKey", n_generateGateKey}, {"resetGateKey", n_resetGateKey}, {"encryptBit", n_encryptBit}, {"constantBit", n_constantBit},
        {"decryptBit", n_decryptBit}, {"gateNAND", n_gateNAND}, {"gateAND", n_gateAND}, {"gateOR", n_gateOR},
        {"gateNOR", n_gateNOR}, {"gateXOR", n_gateXOR}, {"gateXNOR", n_gateXNOR}, {"gateNOT", n_gateNOT},
        {"gateMUX", n_gateMUX}, {"gateMAJ", n_gateMAJ}, {"gateXOR3", n_gateXOR3}, {"exportSecretKey", n_exportSecretKey}, {"importSecretKey", n_importSecretKey},
        {"exportCloudKey", n_exportCloudKey}, {"importCloudKey", n_importCloudKey},
        {"exportCloudKeyToFile", n_exportCloudKeyToFile}, {"importCloudKeyFromFile", n_importCloudKeyFromFile},
        {"keyMode", n_keyMode},
        {"sampleInts", n_sampleInts}, {"encryptBits", n_encryptBits}, {"decryptBits", n_decryptBits},
        {"gateBatch", n_gateBatch}, {"deviceCount", n_deviceCount}, {"circuitRun", n_circuitRun},
        {"netlistOptimize", n_netlistOptimize}, {"circuitBootstraps", n_circuitBootstraps}, {"engineCount", n_engineCount},
        {"netlistCost", n_netlistCost}, {"netlistDepth", n_netlistDepth},
        {"setDevices", n_setDevices}, {"hostAlloc", n_hostAlloc}, {"gateBatchSubmit", n_gateBatchSubmit},
        {"gateBatchWait", n_gateBatchWait},
    };
    for (size_t i = 0; i < sizeof tab / sizeof tab[0]; i++) {
        napi_value fn;
        if (napi_create_function(env, tab[i].name, NAPI_AUTO_LENGTH, tab[i].fn, NULL, &fn) != napi_ok) return NULL;
        if (napi_set_named_property(env, exports, tab[i].name, fn) != napi_ok) return NULL;
    }
    return exports;
}
NAPI_MODULE(eoc_tfhe, init)
