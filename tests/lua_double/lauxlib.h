/* lauxlib.h -- TEST DOUBLE (see lua.h in this directory): the auxiliary-library calls the binding text uses. */
#ifndef EOC_LUA_DOUBLE_LAUXLIB_H
#define EOC_LUA_DOUBLE_LAUXLIB_H
#include "lua.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct luaL_Reg {
    const char *name;
    lua_CFunction func;
} luaL_Reg;

lua_Integer luaL_checkinteger(lua_State *L, int arg);
const char *luaL_checklstring(lua_State *L, int arg, size_t *len);
const char *luaL_optlstring(lua_State *L, int arg, const char *def, size_t *len);
#define luaL_checkstring(L, n) (luaL_checklstring((L), (n), NULL))
#define luaL_optstring(L, n, d) (luaL_optlstring((L), (n), (d), NULL))
void ld_newlib(lua_State *L, const luaL_Reg *regs, size_t nregs);   /* pushes a table of the entries before {NULL, NULL} */
#define luaL_newlib(L, l) ld_newlib((L), (l), sizeof(l) / sizeof((l)[0]) - 1)

#ifdef __cplusplus
}
#endif
#endif
