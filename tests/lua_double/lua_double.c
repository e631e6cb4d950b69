/* lua_double.c -- implementation of the test double declared in lua.h / lauxlib.h (this directory).  NOT Lua. */
#include "lauxlib.h"

#include <setjmp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int type;             /* LUA_TNIL, LUA_TNUMBER (integers only), LUA_TSTRING, LUA_TTABLE */
    long long i;
    char *s;              /* LUA_TSTRING: owned, len bytes + a terminating 0 (Lua strings are 0-terminated for C) */
    size_t len;
    const luaL_Reg *regs; /* LUA_TTABLE */
    size_t nregs;
} ld_value;

struct lua_State {
    ld_value *v;
    int top, cap;
    jmp_buf *catcher;
    char err[256];
    size_t live;
};

static void drop(lua_State *L, ld_value *x)
{
    if (x->type == LUA_TSTRING) {
        L->live -= x->len + 1;
        free(x->s);
    }
    memset(x, 0, sizeof *x);
}
static ld_value *push_slot(lua_State *L)
{
    if (L->top == L->cap) {
        L->cap = L->cap ? 2 * L->cap : 16;
        L->v = realloc(L->v, (size_t)L->cap * sizeof(ld_value));
        if (!L->v) abort();
    }
    ld_value *x = &L->v[L->top++];
    memset(x, 0, sizeof *x);
    return x;
}
static ld_value *at(lua_State *L, int idx) { return (idx >= 1 && idx <= L->top) ? &L->v[idx - 1] : NULL; }
static void raise(lua_State *L, int arg, const char *what)
{
    snprintf(L->err, sizeof L->err, "bad argument #%d (%s expected, got %s)", arg, what,
             lua_type(L, arg) == LUA_TNONE ? "no value" : lua_type(L, arg) == LUA_TNIL ? "nil"
             : lua_type(L, arg) == LUA_TNUMBER ? "number" : lua_type(L, arg) == LUA_TSTRING ? "string" : "table");
    if (!L->catcher) {
        fprintf(stderr, "lua_double: %s outside ld_call\n", L->err);
        abort();
    }
    longjmp(*L->catcher, 1);
}

/* ---- the Lua API subset ---- */
int lua_gettop(lua_State *L) { return L->top; }
int lua_type(lua_State *L, int idx)
{
    ld_value *x = at(L, idx);
    return x ? x->type : LUA_TNONE;
}
void lua_pushnil(lua_State *L) { push_slot(L)->type = LUA_TNIL; }
void lua_pushinteger(lua_State *L, lua_Integer v)
{
    ld_value *x = push_slot(L);
    x->type = LUA_TNUMBER;
    x->i = v;
}
const char *lua_pushlstring(lua_State *L, const char *s, size_t len)
{
    ld_value *x = push_slot(L);
    x->type = LUA_TSTRING;
    x->s = malloc(len + 1);
    if (!x->s) abort();
    if (len) memcpy(x->s, s, len);  /* Lua copies: the caller may free its buffer right after (the binding does) */
    x->s[len] = 0;
    x->len = len;
    L->live += len + 1;
    return x->s;
}
const char *lua_pushstring(lua_State *L, const char *s)
{
    if (!s) {
        lua_pushnil(L);
        return NULL;
    }
    return lua_pushlstring(L, s, strlen(s));
}
lua_Integer luaL_checkinteger(lua_State *L, int arg)
{
    ld_value *x = at(L, arg);
    if (!x || x->type != LUA_TNUMBER) raise(L, arg, "number");
    return x->i;
}
const char *luaL_checklstring(lua_State *L, int arg, size_t *len)
{
    ld_value *x = at(L, arg);
    if (x && x->type == LUA_TNUMBER) { /* lua_tolstring converts a number in place */
        char buf[32];
        int n = snprintf(buf, sizeof buf, "%lld", x->i);
        x->type = LUA_TSTRING;
        x->s = malloc((size_t)n + 1);
        if (!x->s) abort();
        memcpy(x->s, buf, (size_t)n + 1);
        x->len = (size_t)n;
        L->live += x->len + 1;
    }
    if (!x || x->type != LUA_TSTRING) raise(L, arg, "string");
    if (len) *len = x->len;
    return x->s;
}
const char *luaL_optlstring(lua_State *L, int arg, const char *def, size_t *len)
{
    if (lua_isnoneornil(L, arg)) {
        if (len) *len = def ? strlen(def) : 0;
        return def;
    }
    return luaL_checklstring(L, arg, len);
}
void ld_newlib(lua_State *L, const luaL_Reg *regs, size_t nregs)
{
    ld_value *x = push_slot(L);
    x->type = LUA_TTABLE;
    x->regs = regs;
    x->nregs = nregs;
}

/* ---- driver side ---- */
lua_State *ld_new(void) { return calloc(1, sizeof(lua_State)); }
void ld_settop0(lua_State *L)
{
    while (L->top) drop(L, &L->v[--L->top]);
}
void ld_close(lua_State *L)
{
    if (!L) return;
    ld_settop0(L);
    free(L->v);
    free(L);
}
void ld_push_nil(lua_State *L) { lua_pushnil(L); }
void ld_push_int(lua_State *L, long long v) { lua_pushinteger(L, v); }
void ld_push_lstr(lua_State *L, const void *s, size_t len) { lua_pushlstring(L, (const char *)s, len); }
int ld_call(lua_State *L, lua_CFunction fn, int nargs)
{
    if (nargs < 0 || nargs > L->top) return -1;
    /* the callee's frame starts at its first argument */
    const int below = L->top - nargs;
    for (int k = 0; k < below; k++) drop(L, &L->v[k]);
    memmove(L->v, L->v + below, (size_t)nargs * sizeof(ld_value));
    L->top = nargs;
    jmp_buf jb;
    L->catcher = &jb;
    L->err[0] = 0;
    int nres;
    if (setjmp(jb)) {
        L->catcher = NULL;
        ld_settop0(L);
        return -1;
    }
    nres = fn(L);
    L->catcher = NULL;
    if (nres < 0 || nres > L->top) {
        snprintf(L->err, sizeof L->err, "C function returned %d with %d values on the stack", nres, L->top);
        ld_settop0(L);
        return -1;
    }
    /* results are the nres topmost values */
    const int keep_from = L->top - nres;
    for (int k = 0; k < keep_from; k++) drop(L, &L->v[k]);
    memmove(L->v, L->v + keep_from, (size_t)nres * sizeof(ld_value));
    L->top = nres;
    return nres;
}
const char *ld_error(lua_State *L) { return L->err; }
int ld_type(lua_State *L, int idx) { return lua_type(L, idx); }
long long ld_to_int(lua_State *L, int idx)
{
    ld_value *x = at(L, idx);
    return x && x->type == LUA_TNUMBER ? x->i : 0;
}
const void *ld_to_lstr(lua_State *L, int idx, size_t *len)
{
    ld_value *x = at(L, idx);
    if (!x || x->type != LUA_TSTRING) return NULL;
    if (len) *len = x->len;
    return x->s;
}
lua_CFunction ld_table_get(lua_State *L, int idx, const char *name)
{
    ld_value *x = at(L, idx);
    if (!x || x->type != LUA_TTABLE) return NULL;
    for (size_t k = 0; k < x->nregs; k++)
        if (strcmp(x->regs[k].name, name) == 0) return x->regs[k].func;
    return NULL;
}
int ld_table_size(lua_State *L, int idx)
{
    ld_value *x = at(L, idx);
    return x && x->type == LUA_TTABLE ? (int)x->nregs : -1;
}
const char *ld_table_name(lua_State *L, int idx, int k)
{
    ld_value *x = at(L, idx);
    return x && x->type == LUA_TTABLE && k >= 0 && (size_t)k < x->nregs ? x->regs[k].name : NULL;
}
size_t ld_live_bytes(lua_State *L) { return L->live; }
