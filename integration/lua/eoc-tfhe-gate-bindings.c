/* eoc-tfhe-gate-bindings.c -- text to merge into ao-tfhe/eoc-tfhe-bindings.c (pattern of l_addCiphertexts, :12-24).
 * NOT compiled in this repository: the build image has no Lua 5.3 SDK (lua.h).  The equivalent Node binding,
 * integration/node/eoc_tfhe_node.c, is built and tested.  Needs <lua.h>, <lauxlib.h>, <stdlib.h>. */
#include "eoc_tfhe_gpu.h"

static int l_generateGateKey(lua_State *L) {          /* like l_generateSecretKey, :38-48 */
  int lambda = (int)luaL_checkinteger(L, 1);
  uint64_t seed = (uint64_t)luaL_checkinteger(L, 2);
  const char *r = generateGateKey(lambda, seed);
  lua_pushstring(L, r);                               /* NULL -> nil, as in the reference */
  free((void *)r);
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

/* appended to the luaL_Reg table of luaopen_tfhe (ao-tfhe/eoc-tfhe-bindings.c:130-144) */
  {"generateGateKey", l_generateGateKey}, {"encryptBit", l_encryptBit}, {"decryptBit", l_decryptBit},
  {"gateNAND", l_gateNAND}, {"gateAND", l_gateAND}, {"gateOR", l_gateOR}, {"gateNOR", l_gateNOR},
  {"gateXOR", l_gateXOR}, {"gateXNOR", l_gateXNOR}, {"gateNOT", l_gateNOT}, {"gateMUX", l_gateMUX},
