/* lua.h -- TEST DOUBLE of the dozen Lua 5.3 C-API calls integration/lua/eoc-tfhe-gate-bindings.c uses.
 *
 * NOT Lua.  The build image has no Lua SDK (no lua.h, no liblua), so the binding text of integration/lua/ could never
 * be compiled, let alone run.  This double lets it compile and lets a C or Python driver call every l_* entry the way
 * the Lua VM would: arguments on a value stack (1-based indices), results pushed on it, luaL_check* failures raised as
 * errors.  What is tested through it is THIS REPOSITORY'S C (argument marshalling, length arithmetic, NULL -> nil,
 * ownership of heap results), not Lua and not the reference's binding (ao-tfhe/eoc-tfhe-bindings.c:12-24, 128-148, whose
 * calling pattern the text follows).  Semantics kept from the Lua 5.3 manual: lua_pushstring(L, NULL) pushes nil;
 * luaL_checkinteger accepts integers (strings convertible to integers are NOT accepted here: stricter, on purpose);
 * luaL_checklstring accepts strings and integers (converted in place, as Lua does); luaL_opt* treat none and nil alike;
 * a failed check does not return (longjmp to the driver's ld_call).
 */
#ifndef EOC_LUA_DOUBLE_LUA_H
#define EOC_LUA_DOUBLE_LUA_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lua_State lua_State;
typedef long long lua_Integer;
typedef int (*lua_CFunction)(lua_State *L);

#define LUA_TNONE (-1)
#define LUA_TNIL 0
#define LUA_TNUMBER 3
#define LUA_TSTRING 4
#define LUA_TTABLE 5

int lua_gettop(lua_State *L);
int lua_type(lua_State *L, int idx);
#define lua_isnoneornil(L, n) (lua_type((L), (n)) <= 0)
void lua_pushnil(lua_State *L);
void lua_pushinteger(lua_State *L, lua_Integer v);
const char *lua_pushstring(lua_State *L, const char *s);            /* NULL -> nil */
const char *lua_pushlstring(lua_State *L, const char *s, size_t len);

/* ---- driver side (not part of the Lua API) ---- */
lua_State *ld_new(void);
void ld_close(lua_State *L);
void ld_settop0(lua_State *L);                                      /* drop everything */
void ld_push_nil(lua_State *L);
void ld_push_int(lua_State *L, long long v);
void ld_push_lstr(lua_State *L, const void *s, size_t len);
/* calls fn with the nargs topmost values as its arguments 1..nargs (values below them are discarded first);
 * returns the number of results (left on the stack as 1..nres) or -1 when a luaL_check* raised (see ld_error) */
int ld_call(lua_State *L, lua_CFunction fn, int nargs);
const char *ld_error(lua_State *L);
int ld_type(lua_State *L, int idx);
long long ld_to_int(lua_State *L, int idx);
const void *ld_to_lstr(lua_State *L, int idx, size_t *len);         /* NULL unless a string */
/* looks a function up in the table on top of the stack (what luaL_newlib built); NULL if absent */
lua_CFunction ld_table_get(lua_State *L, int idx, const char *name);
int ld_table_size(lua_State *L, int idx);
const char *ld_table_name(lua_State *L, int idx, int k);
size_t ld_live_bytes(lua_State *L);                                 /* bytes the state holds for strings (leak accounting) */

#ifdef __cplusplus
}
#endif
#endif
