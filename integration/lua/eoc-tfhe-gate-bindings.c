/* eoc-tfhe-gate-bindings.c -- Lua 5.3 binding of the gate API, in the pattern of ao-tfhe/eoc-tfhe-bindings.c
 * (l_addCiphertexts, :12-24: read the arguments, call the extern "C" function, lua_pushstring the heap result, free() it,
 * return 1).  A maintainer of the reference either merges the l_* functions and the entries of eoc_gate_functions[] into
 * luaopen_tfhe's luaL_Reg table (ao-tfhe/eoc-tfhe-bindings.c:130-144), or builds this file as a module of its own
 * (luaopen_tfhe_gates below; require("tfhe_gates")).
 *
 * Status: COMPILED AND EXECUTED AGAINST A TEST DOUBLE of the Lua C API (tests/lua_double/: a value stack with the dozen
 * calls used here) -- every l_* entry is driven from C under ASan/UBSan (tests/c/lua_binding_driver.c, CPU legs) and from
 * Python on the GPU against the oracle (tests/test_lua_binding.py).  NEVER against liblua: the build image has no Lua SDK.
 * The equivalent Node binding, integration/node/eoc_tfhe_node.c, is built against the real N-API and has the same entries.
 *
 * Raw-buffer calls take and return Lua strings holding binary data (lua_pushlstring / luaL_checklstring): LWE samples
 * are int32 little-endian [count][n+1], bit arrays one byte per bit, netlists int32 [5 per gate: op, in0, in1, in2, out]. */
#include <lua.h>
#include <lauxlib.h>
#include <stdlib.h>
#include <string.h>
#include "eoc_tfhe_gpu.h"

static int l_generateGateKey(lua_State *L) {          /* like l_generateSecretKey, :38-48 */
  int lambda = (int)luaL_checkinteger(L, 1);
  uint64_t seed = (uint64_t)luaL_checkinteger(L, 2);
  const char *r = generateGateKey(lambda, seed);
  lua_pushstring(L, r);                               /* NULL -> nil, as in the reference */
  free((void *)r);
  return 1;
}
static int l_resetGateKey(lua_State *L) { (void)L; resetGateKey(); return 0; }
static int l_setDevices(lua_State *L) {               /* (dev, dev, ...) -> 0 or a negative code; no arguments = forget */
  int devs[64], n = lua_gettop(L);                    /* the devices the next generateGateKey / importCloudKey brings up */
  if (n > 64) n = 64;
  for (int i = 0; i < n; i++) devs[i] = (int)luaL_checkinteger(L, i + 1);
  lua_pushinteger(L, eoc_gpu_set_devices(devs, n));
  return 1;
}
static int l_encryptBit(lua_State *L) {               /* like l_encryptInteger, :59-67 */
  int bit = (int)luaL_checkinteger(L, 1);
  const char *key = luaL_optstring(L, 2, "");
  (void)key;                                          /* ignored, as :63 passes NULL */
  const char *r = encryptBit(bit, NULL);
  lua_pushstring(L, r);
  free((void *)r);
  return 1;
}
static int l_constantBit(lua_State *L) {              /* bootsCONSTANT: noiseless trivial sample, no key involved */
  const char *r = constantBit((int)luaL_checkinteger(L, 1));
  lua_pushstring(L, r);
  free((void *)r);
  return 1;
}
static int l_decryptBit(lua_State *L) {               /* like l_decryptInteger, :90-100 */
  const char *ct = luaL_checkstring(L, 1);
  lua_pushinteger(L, decryptBit(ct, NULL));
  return 1;
}
#define EOC_GATE2(NAME)                                                   \
  static int l_##NAME(lua_State *L) {                                     \
    const char *a = luaL_checkstring(L, 1), *b = luaL_checkstring(L, 2);  \
    const char *pk = luaL_optstring(L, 3, "");                            \
    const char *r = NAME(a, b, pk);                                       \
    lua_pushstring(L, r);                                                 \
    free((void *)r);                                                      \
    return 1;                                                             \
  }
EOC_GATE2(gateNAND) EOC_GATE2(gateAND) EOC_GATE2(gateOR) EOC_GATE2(gateNOR)
EOC_GATE2(gateXOR)  EOC_GATE2(gateXNOR)
static int l_gateNOT(lua_State *L) {
  const char *r = gateNOT(luaL_checkstring(L, 1), luaL_optstring(L, 2, ""));
  lua_pushstring(L, r); free((void *)r); return 1;
}
static int l_gateMUX(lua_State *L) {
  const char *r = gateMUX(luaL_checkstring(L, 1), luaL_checkstring(L, 2), luaL_checkstring(L, 3),
                          luaL_optstring(L, 4, ""));
  lua_pushstring(L, r); free((void *)r); return 1;
}
/* the extension gates (EOC_MAJ, EOC_XOR3): three ciphertexts in, one out, like gateMUX */
static int l_gateMAJ(lua_State *L) {
  const char *r = gateMAJ(luaL_checkstring(L, 1), luaL_checkstring(L, 2), luaL_checkstring(L, 3),
                          luaL_optstring(L, 4, ""));
  lua_pushstring(L, r); free((void *)r); return 1;
}
static int l_gateXOR3(lua_State *L) {
  const char *r = gateXOR3(luaL_checkstring(L, 1), luaL_checkstring(L, 2), luaL_checkstring(L, 3),
                           luaL_optstring(L, 4, ""));
  lua_pushstring(L, r); free((void *)r); return 1;
}
static int l_exportSecretKey(lua_State *L) {
  const char *r = exportSecretKey();
  lua_pushstring(L, r); free((void *)r); return 1;
}
static int l_importSecretKey(lua_State *L) {
  lua_pushinteger(L, importSecretKey(luaL_checkstring(L, 1)));
  return 1;
}
/* the cloud ("public") key (eoc-tfhe-run.cpp:232-234): exported by the client, installed by a secret-free server.
 * l_generatePublicKey's commented-out body (ao-tfhe/eoc-tfhe-bindings.c:51-57) can be restored as it stands:
 * generatePublicKey() now returns this same export. */
static int l_exportCloudKey(lua_State *L) {
  const char *r = exportCloudKey();
  lua_pushstring(L, r); free((void *)r); return 1;
}
static int l_importCloudKey(lua_State *L) {
  lua_pushinteger(L, importCloudKey(luaL_checkstring(L, 1)));
  return 1;
}
static int l_exportCloudKeyToFile(lua_State *L) {
  lua_pushinteger(L, exportCloudKeyToFile(luaL_checkstring(L, 1)));
  return 1;
}
static int l_importCloudKeyFromFile(lua_State *L) {
  lua_pushinteger(L, importCloudKeyFromFile(luaL_checkstring(L, 1)));
  return 1;
}
static int l_keyMode(lua_State *L) { lua_pushinteger(L, eoc_global_key_mode()); return 1; }

/* ---- raw-buffer batch calls on the global key ---- */
static int l_sampleInts(lua_State *L) {               /* n + 1 of the global key, or -1 */
  eoc_params p;
  lua_pushinteger(L, eoc_global_params(&p) == EOC_OK ? p.n + 1 : -1);
  return 1;
}
static int l_encryptBits(lua_State *L) {              /* (bits: string, one byte per bit) -> samples */
  size_t count;
  const char *bits = luaL_checklstring(L, 1, &count);
  eoc_params p;
  if (eoc_global_params(&p) != EOC_OK) { lua_pushnil(L); return 1; }
  size_t bytes = count * (size_t)(p.n + 1) * 4;
  int32_t *cts = malloc(bytes ? bytes : 1);
  if (cts && eoc_global_encrypt_bits((const uint8_t *)bits, count, cts) == EOC_OK) lua_pushlstring(L, (const char *)cts, bytes);
  else lua_pushnil(L);
  free(cts);
  return 1;
}
static int l_decryptBits(lua_State *L) {              /* (samples) -> bits: string, one byte per bit */
  size_t bytes;
  const char *cts = luaL_checklstring(L, 1, &bytes);
  eoc_params p;
  if (eoc_global_params(&p) != EOC_OK || bytes % ((size_t)(p.n + 1) * 4)) { lua_pushnil(L); return 1; }
  size_t count = bytes / ((size_t)(p.n + 1) * 4);
  uint8_t *bits = malloc(count ? count : 1);
  if (bits && eoc_global_decrypt_bits((const int32_t *)cts, count, bits) == EOC_OK) lua_pushlstring(L, (const char *)bits, count);
  else lua_pushnil(L);
  free(bits);
  return 1;
}
static int l_gateBatch(lua_State *L) {                /* (op, in0, in1 | nil, in2 | nil, ops | nil) -> samples or nil */
  int op = (int)luaL_checkinteger(L, 1);
  size_t bytes[3] = {0, 0, 0}, nops = 0;
  const char *in[3] = {NULL, NULL, NULL}, *ops = NULL;
  for (int k = 0; k < 3; k++)
    if (!lua_isnoneornil(L, 2 + k)) in[k] = luaL_checklstring(L, 2 + k, &bytes[k]);
  if (!lua_isnoneornil(L, 5)) ops = luaL_checklstring(L, 5, &nops);
  eoc_params p;
  if (!in[0] || eoc_global_params(&p) != EOC_OK) { lua_pushnil(L); return 1; }
  const size_t row = (size_t)(p.n + 1) * 4, count = bytes[0] / row;
  if (bytes[0] % row || (in[1] && bytes[1] != bytes[0]) || (in[2] && bytes[2] != bytes[0]) || (ops && nops != count)) {
    lua_pushnil(L);                                   /* every supplied operand has in0's length */
    return 1;
  }
  int32_t *out = malloc(bytes[0] ? bytes[0] : 1);
  if (out && eoc_global_gate_batch(op, (const uint8_t *)ops, (const int32_t *)in[0], (const int32_t *)in[1],
                                   (const int32_t *)in[2], out, count) == EOC_OK)
    lua_pushlstring(L, (const char *)out, bytes[0]);
  else lua_pushnil(L);
  free(out);
  return 1;
}
static int l_circuitRun(lua_State *L) {               /* (gates, wires, nWires, instances) -> wires after evaluation, or nil */
  size_t gbytes, wbytes;
  const char *g = luaL_checklstring(L, 1, &gbytes), *w = luaL_checklstring(L, 2, &wbytes);
  size_t n_wires = (size_t)luaL_checkinteger(L, 3), inst = (size_t)luaL_checkinteger(L, 4);
  eoc_params p;
  if (gbytes % sizeof(eoc_gate) || eoc_global_params(&p) != EOC_OK || wbytes != n_wires * inst * (size_t)(p.n + 1) * 4) {
    lua_pushnil(L);
    return 1;
  }
  int32_t *wires = malloc(wbytes ? wbytes : 1);       /* Lua strings are immutable: evaluate on a copy */
  if (wires) memcpy(wires, w, wbytes);
  if (wires && eoc_global_circuit_run((const eoc_gate *)g, gbytes / sizeof(eoc_gate), wires, n_wires, inst) == EOC_OK)
    lua_pushlstring(L, (const char *)wires, wbytes);
  else lua_pushnil(L);
  free(wires);
  return 1;
}
static int l_netlistOptimize(lua_State *L) {          /* (gates, outputs: int32 wire ids) -> gates or nil */
  size_t gbytes, obytes;
  const char *g = luaL_checklstring(L, 1, &gbytes), *o = luaL_checklstring(L, 2, &obytes);
  if (gbytes % sizeof(eoc_gate) || obytes % 4) { lua_pushnil(L); return 1; }
  eoc_gate *tmp = malloc(gbytes + sizeof(eoc_gate));
  int64_t n = tmp ? eoc_netlist_optimize((const eoc_gate *)g, gbytes / sizeof(eoc_gate), (const int32_t *)o, obytes / 4, tmp) : -1;
  if (n >= 0) lua_pushlstring(L, (const char *)tmp, (size_t)n * sizeof(eoc_gate));
  else lua_pushnil(L);
  free(tmp);
  return 1;
}

static int l_circuitBootstraps(lua_State *L) {        /* (gates) -> blind rotations per instance (MUX = 2, NOT / COPY = 0), or -1 */
  size_t gbytes;
  const char *g = luaL_checklstring(L, 1, &gbytes);
  if (gbytes % sizeof(eoc_gate)) { lua_pushinteger(L, -1); return 1; }
  lua_pushinteger(L, (lua_Integer)eoc_circuit_bootstraps((const eoc_gate *)g, gbytes / sizeof(eoc_gate)));
  return 1;
}
static int l_netlistCost(lua_State *L) {              /* (gates, instances) -> estimated run time in 0.1 ms units (eoc_netlist_cost), or -1 */
  size_t gbytes;
  const char *g = luaL_checklstring(L, 1, &gbytes);
  lua_Integer inst = luaL_checkinteger(L, 2);
  if (gbytes % sizeof(eoc_gate) || inst < 0) { lua_pushinteger(L, -1); return 1; }
  lua_pushinteger(L, (lua_Integer)eoc_netlist_cost((const eoc_gate *)g, gbytes / sizeof(eoc_gate), (size_t)inst, 0));
  return 1;
}
static int l_netlistDepth(lua_State *L) {             /* (gates) -> dependent levels that hold a blind rotation, or -1 */
  size_t gbytes;
  const char *g = luaL_checklstring(L, 1, &gbytes);
  int64_t depth = -1;
  if (gbytes % sizeof(eoc_gate) || eoc_netlist_levels((const eoc_gate *)g, gbytes / sizeof(eoc_gate), NULL, &depth) < 0) depth = -1;
  lua_pushinteger(L, (lua_Integer)depth);
  return 1;
}
static int l_deviceCount(lua_State *L) { lua_pushinteger(L, eoc_device_count()); return 1; }
static int l_engineCount(lua_State *L) { lua_pushinteger(L, eoc_gpu_engine_count()); return 1; }
/* NOT mirrored from the Node addon (tests/test_binding_surfaces.py lists them): hostAlloc / gateBatchSubmit /
 * gateBatchWait -- the asynchronous batch path works on pinned, MUTABLE host buffers the caller keeps alive, which a
 * Lua string (immutable, garbage collected) cannot be. */

/* the entries to append to the luaL_Reg table of luaopen_tfhe (ao-tfhe/eoc-tfhe-bindings.c:130-144) */
static const luaL_Reg eoc_gate_functions[] = {
  {"generateGateKey", l_generateGateKey}, {"resetGateKey", l_resetGateKey}, {"setDevices", l_setDevices},
  {"encryptBit", l_encryptBit},
  {"constantBit", l_constantBit}, {"decryptBit", l_decryptBit},
  {"gateNAND", l_gateNAND}, {"gateAND", l_gateAND}, {"gateOR", l_gateOR}, {"gateNOR", l_gateNOR},
  {"gateXOR", l_gateXOR}, {"gateXNOR", l_gateXNOR}, {"gateNOT", l_gateNOT}, {"gateMUX", l_gateMUX},
  {"gateMAJ", l_gateMAJ}, {"gateXOR3", l_gateXOR3},
  {"exportSecretKey", l_exportSecretKey}, {"importSecretKey", l_importSecretKey},
  {"exportCloudKey", l_exportCloudKey}, {"importCloudKey", l_importCloudKey},
  {"exportCloudKeyToFile", l_exportCloudKeyToFile}, {"importCloudKeyFromFile", l_importCloudKeyFromFile},
  {"keyMode", l_keyMode},
  {"sampleInts", l_sampleInts}, {"encryptBits", l_encryptBits}, {"decryptBits", l_decryptBits},
  {"gateBatch", l_gateBatch}, {"circuitRun", l_circuitRun}, {"netlistOptimize", l_netlistOptimize},
  {"circuitBootstraps", l_circuitBootstraps}, {"netlistCost", l_netlistCost}, {"netlistDepth", l_netlistDepth},
  {"deviceCount", l_deviceCount}, {"engineCount", l_engineCount},
  {NULL, NULL}
};
/* the same functions as a module of their own, like luaopen_tfhe (:128-148) */
int luaopen_tfhe_gates(lua_State *L) {
  luaL_newlib(L, eoc_gate_functions);
  return 1;
}
