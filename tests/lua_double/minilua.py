"""minilua -- a SMALL interpreter for the subset of Lua 5.3 that integration/lua/tfhe_gates.lua is written in.

TEST DOUBLE, NOT LUA.  The build image has no Lua interpreter, so the facade text appended to ao-tfhe/tfhe.lua could
never be run.  This module lexes, parses and evaluates exactly the language features that text uses, so that
tests/test_lua_facade.py can EXECUTE the real file: its netlist builders against eoc_tfhe_amd/circuits.py, its wire
packing (string.pack, planes(), runNetlist) against the C binding driven through the Lua C-API double, and the whole
chain  facade text -> l_* binding C -> libeoc_tfhe_gpu.so -> GPU  against plaintext and the oracle.  What it checks is
THIS REPOSITORY'S Lua text; it says nothing about a real Lua VM beyond the semantics restated here from the Lua 5.3
reference manual:

  values      nil, booleans, integers and floats (Python int / float), strings = BYTE strings (Python bytes), tables,
              functions; multiple assignment and multiple returns with Lua's truncate / expand rules
  statements  local, local function, function a.b.c(...) / a.b:m(...), assignment, call, if / elseif / else, while,
              numeric for (optional step), generic for over pairs / ipairs, repeat-until, do-end, return, break
  expressions the full operator precedence table (or, and, comparisons, | ~ & << >>, .., + -, * / // %, unary not # - ~, ^),
              table constructors ({v, k = v, [e] = v}), calls, method calls (a:m(...)), varargs (...), closures
  library     string.pack / unpack (little-endian fixed-size integer formats: the "<i4" family), rep, char, byte, sub, len,
              format (%d %s), table.concat / insert / unpack, math.floor / max / min / huge, select, type, tostring, tonumber,
              pairs, ipairs, assert, error, print, the string metatable (s:sub, s:byte, ...), # on strings and tables

Anything outside that subset raises LuaError("minilua: unsupported ...") rather than guessing.
"""
import re
import struct


class LuaError(Exception):
    pass


class LuaTable:
    __slots__ = ("h",)

    def __init__(self):
        self.h = {}

    def get(self, k):
        if isinstance(k, float) and k.is_integer():
            k = int(k)
        return self.h.get(k)

    def set(self, k, v):
        if k is None:
            raise LuaError("table index is nil")
        if isinstance(k, float) and k.is_integer():
            k = int(k)
        if v is None:
            self.h.pop(k, None)
        else:
            self.h[k] = v

    def length(self):          # a border: t[n] ~= nil and t[n + 1] == nil (the one counting up from 1)
        n = 0
        while (n + 1) in self.h:
            n += 1
        return n


class LuaFunction:
    __slots__ = ("params", "vararg", "body", "env", "name")

    def __init__(self, params, vararg, body, env, name):
        self.params, self.vararg, self.body, self.env, self.name = params, vararg, body, env, name


class _Break(Exception):
    pass


class _Return(Exception):
    def __init__(self, values):
        self.values = values


# ---- lexer ----------------------------------------------------------------------------------------------------------
KEYWORDS = {"and", "break", "do", "else", "elseif", "end", "false", "for", "function", "goto", "if", "in", "local", "nil",
            "not", "or", "repeat", "return", "then", "true", "until", "while"}
TOKEN_RE = re.compile(rb"""
    (?P<ws>\s+|--\[\[.*?\]\]|--[^\n]*) |
    (?P<num>0[xX][0-9a-fA-F]+|\d+\.\d*(?:[eE][+-]?\d+)?|\d+(?:[eE][+-]?\d+)?|\.\d+) |
    (?P<name>[A-Za-z_][A-Za-z_0-9]*) |
    (?P<str>"(?:\\.|[^"\\\n])*"|'(?:\\.|[^'\\\n])*') |
    (?P<op>\.\.\.|\.\.|==|~=|<=|>=|<<|>>|//|::|[-+*/%^\#&~|<>=(){}\[\];:,.])
""", re.X | re.S)
ESC = {b"n": b"\n", b"t": b"\t", b"r": b"\r", b"a": b"\a", b"b": b"\b", b"f": b"\f", b"v": b"\v", b"\\": b"\\", b'"': b'"',
       b"'": b"'", b"\n": b"\n"}


def _unescape(body):
    out, i = bytearray(), 0
    while i < len(body):
        c = body[i:i + 1]
        if c != b"\\":
            out += c
            i += 1
            continue
        nx = body[i + 1:i + 2]
        if nx in ESC:
            out += ESC[nx]
            i += 2
        elif nx == b"x":
            out.append(int(body[i + 2:i + 4], 16))
            i += 4
        elif nx.isdigit():
            j = i + 1
            while j < len(body) and j < i + 4 and body[j:j + 1].isdigit():
                j += 1
            out.append(int(body[i + 1:j]))
            i = j
        elif nx == b"z":
            i += 2
            while i < len(body) and body[i:i + 1].isspace():
                i += 1
        else:
            raise LuaError("minilua: unsupported escape \\" + nx.decode("latin1"))
    return bytes(out)


def lex(src):
    if isinstance(src, str):
        src = src.encode("utf-8")
    toks, pos, line = [], 0, 1
    while pos < len(src):
        m = TOKEN_RE.match(src, pos)
        if not m:
            raise LuaError(f"minilua: cannot lex at line {line}: {src[pos:pos + 20]!r}")
        text = m.group(0)
        if m.lastgroup == "num":
            t = text.decode()
            v = int(t, 16) if t[:2].lower() == "0x" else (float(t) if any(c in t for c in ".eE") else int(t))
            toks.append(("num", v, line))
        elif m.lastgroup == "name":
            t = text.decode()
            toks.append(("kw" if t in KEYWORDS else "name", t, line))
        elif m.lastgroup == "str":
            toks.append(("str", _unescape(text[1:-1]), line))
        elif m.lastgroup == "op":
            toks.append(("op", text.decode(), line))
        line += text.count(b"\n")
        pos = m.end()
    toks.append(("eof", None, line))
    return toks


# ---- parser (AST = nested tuples) -----------------------------------------------------------------------------------
BINPRI = {"or": (1, 1), "and": (2, 2), "<": (3, 3), ">": (3, 3), "<=": (3, 3), ">=": (3, 3), "~=": (3, 3), "==": (3, 3),
          "|": (4, 4), "~": (5, 5), "&": (6, 6), "<<": (7, 7), ">>": (7, 7), "..": (9, 8), "+": (10, 10), "-": (10, 10),
          "*": (11, 11), "/": (11, 11), "//": (11, 11), "%": (11, 11), "^": (14, 13)}
UNARY_PRI = 12


class Parser:
    def __init__(self, toks):
        self.t, self.i = toks, 0

    def peek(self):
        return self.t[self.i]

    def next(self):
        tok = self.t[self.i]
        self.i += 1
        return tok

    def check(self, kind, val=None):
        k, v, _ = self.t[self.i]
        return k == kind and (val is None or v == val)

    def accept(self, kind, val=None):
        if self.check(kind, val):
            return self.next()
        return None

    def expect(self, kind, val=None):
        if not self.check(kind, val):
            k, v, ln = self.t[self.i]
            raise LuaError(f"minilua: line {ln}: expected {val or kind}, got {v!r}")
        return self.next()

    def block_end(self):
        k, v, _ = self.peek()
        return k == "eof" or (k == "kw" and v in ("end", "else", "elseif", "until"))

    def block(self):
        stmts = []
        while not self.block_end():
            if self.accept("op", ";"):
                continue
            if self.check("kw", "return"):
                self.next()
                exprs = [] if (self.block_end() or self.check("op", ";")) else self.exprlist()
                self.accept("op", ";")
                stmts.append(("return", exprs))
                break
            stmts.append(self.statement())
        return stmts

    def statement(self):
        k, v, ln = self.peek()
        if k == "kw":
            if v == "if":
                self.next()
                clauses, orelse = [], None
                cond = self.expr()
                self.expect("kw", "then")
                clauses.append((cond, self.block()))
                while True:
                    if self.accept("kw", "elseif"):
                        cond = self.expr()
                        self.expect("kw", "then")
                        clauses.append((cond, self.block()))
                    elif self.accept("kw", "else"):
                        orelse = self.block()
                        self.expect("kw", "end")
                        break
                    else:
                        self.expect("kw", "end")
                        break
                return ("if", clauses, orelse)
            if v == "while":
                self.next()
                cond = self.expr()
                self.expect("kw", "do")
                body = self.block()
                self.expect("kw", "end")
                return ("while", cond, body)
            if v == "repeat":
                self.next()
                body = self.block()
                self.expect("kw", "until")
                return ("repeat", body, self.expr())
            if v == "do":
                self.next()
                body = self.block()
                self.expect("kw", "end")
                return ("do", body)
            if v == "for":
                self.next()
                n1 = self.expect("name")[1]
                if self.accept("op", "="):
                    a = self.expr()
                    self.expect("op", ",")
                    b = self.expr()
                    c = self.expr() if self.accept("op", ",") else None
                    self.expect("kw", "do")
                    body = self.block()
                    self.expect("kw", "end")
                    return ("fornum", n1, a, b, c, body)
                names = [n1]
                while self.accept("op", ","):
                    names.append(self.expect("name")[1])
                self.expect("kw", "in")
                exprs = self.exprlist()
                self.expect("kw", "do")
                body = self.block()
                self.expect("kw", "end")
                return ("forin", names, exprs, body)
            if v == "function":
                self.next()
                target = ("name", self.expect("name")[1])
                is_method = False
                while True:
                    if self.accept("op", "."):
                        target = ("index", target, ("const", self.expect("name")[1].encode()))
                    elif self.accept("op", ":"):
                        target = ("index", target, ("const", self.expect("name")[1].encode()))
                        is_method = True
                        break
                    else:
                        break
                return ("assign", [target], [self.funcbody(is_method, "function")])
            if v == "local":
                self.next()
                if self.accept("kw", "function"):
                    name = self.expect("name")[1]
                    return ("localfunc", name, self.funcbody(False, name))
                names = [self.expect("name")[1]]
                while self.accept("op", ","):
                    names.append(self.expect("name")[1])
                exprs = self.exprlist() if self.accept("op", "=") else []
                return ("local", names, exprs)
            if v == "break":
                self.next()
                return ("break",)
            raise LuaError(f"minilua: unsupported statement '{v}' at line {ln}")
        e = self.suffixedexp()
        if self.check("op", "=") or self.check("op", ","):
            targets = [e]
            while self.accept("op", ","):
                targets.append(self.suffixedexp())
            self.expect("op", "=")
            for tg in targets:
                if tg[0] not in ("name", "index"):
                    raise LuaError(f"minilua: line {ln}: cannot assign to this expression")
            return ("assign", targets, self.exprlist())
        if e[0] not in ("call", "method"):
            raise LuaError(f"minilua: line {ln}: syntax error (expression is not a statement)")
        return ("callstat", e)

    def funcbody(self, is_method, name):
        self.expect("op", "(")
        params, vararg = (["self"] if is_method else []), False
        if not self.check("op", ")"):
            while True:
                if self.accept("op", "..."):
                    vararg = True
                    break
                params.append(self.expect("name")[1])
                if not self.accept("op", ","):
                    break
        self.expect("op", ")")
        body = self.block()
        self.expect("kw", "end")
        return ("function", params, vararg, body, name)

    def exprlist(self):
        out = [self.expr()]
        while self.accept("op", ","):
            out.append(self.expr())
        return out

    def primaryexp(self):
        k, v, ln = self.next()
        if k == "name":
            return ("name", v)
        if k == "op" and v == "(":
            e = self.expr()
            self.expect("op", ")")
            return ("paren", e)
        raise LuaError(f"minilua: line {ln}: unexpected {v!r}")

    def suffixedexp(self):
        e = self.primaryexp()
        while True:
            if self.accept("op", "."):
                e = ("index", e, ("const", self.expect("name")[1].encode()))
            elif self.accept("op", "["):
                k = self.expr()
                self.expect("op", "]")
                e = ("index", e, k)
            elif self.accept("op", ":"):
                name = self.expect("name")[1]
                e = ("method", e, name.encode(), self.callargs())
            elif self.check("op", "(") or self.check("str") or self.check("op", "{"):
                e = ("call", e, self.callargs())
            else:
                return e

    def callargs(self):
        if self.check("str"):
            return [("const", self.next()[1])]
        if self.check("op", "{"):
            return [self.tablecons()]
        self.expect("op", "(")
        args = [] if self.check("op", ")") else self.exprlist()
        self.expect("op", ")")
        return args

    def tablecons(self):
        self.expect("op", "{")
        items = []          # ("pos", e) | ("kv", k, v)
        while not self.check("op", "}"):
            if self.check("op", "["):
                self.next()
                k = self.expr()
                self.expect("op", "]")
                self.expect("op", "=")
                items.append(("kv", k, self.expr()))
            elif self.check("name") and self.t[self.i + 1][0] == "op" and self.t[self.i + 1][1] == "=":
                k = ("const", self.next()[1].encode())
                self.next()
                items.append(("kv", k, self.expr()))
            else:
                items.append(("pos", self.expr()))
            if not (self.accept("op", ",") or self.accept("op", ";")):
                break
        self.expect("op", "}")
        return ("table", items)

    def simpleexp(self):
        k, v, ln = self.peek()
        if k == "num" or k == "str":
            self.next()
            return ("const", v)
        if k == "kw":
            if v == "nil":
                self.next()
                return ("const", None)
            if v == "true":
                self.next()
                return ("const", True)
            if v == "false":
                self.next()
                return ("const", False)
            if v == "function":
                self.next()
                return self.funcbody(False, "anonymous")
        if k == "op" and v == "...":
            self.next()
            return ("vararg",)
        if k == "op" and v == "{":
            return self.tablecons()
        return self.suffixedexp()

    def expr(self, limit=0):
        k, v, _ = self.peek()
        if (k == "kw" and v == "not") or (k == "op" and v in ("-", "#", "~")):
            self.next()
            left = ("unop", v, self.expr(UNARY_PRI))
        else:
            left = self.simpleexp()
        while True:
            k, v, _ = self.peek()
            op = v if ((k == "op" and v in BINPRI) or (k == "kw" and v in ("and", "or"))) else None
            if op is None or BINPRI[op][0] <= limit:
                return left
            self.next()
            right = self.expr(BINPRI[op][1])
            left = ("binop", op, left, right)


# ---- evaluator ------------------------------------------------------------------------------------------------------
class Env:
    __slots__ = ("vars", "parent")

    def __init__(self, parent=None):
        self.vars, self.parent = {}, parent

    def lookup(self, name):
        e = self
        while e is not None:
            if name in e.vars:
                return e
            e = e.parent
        return None


def _truthy(v):
    return v is not None and v is not False


def _tostr(v):
    if isinstance(v, bytes):
        return v
    if isinstance(v, bool):
        return b"true" if v else b"false"
    if v is None:
        return b"nil"
    if isinstance(v, int):
        return str(v).encode()
    if isinstance(v, float):
        return (repr(v) if not v.is_integer() else "%.1f" % v).encode()
    return b"<" + type(v).__name__.encode() + b">"


def _arith_operand(v, op):
    if isinstance(v, bool) or not isinstance(v, (int, float)):
        if isinstance(v, bytes):
            try:
                return int(v)
            except ValueError:
                try:
                    return float(v)
                except ValueError:
                    pass
        raise LuaError(f"attempt to perform arithmetic ({op}) on a {_typename(v)} value")
    return v


def _int_operand(v, op):
    v = _arith_operand(v, op)
    if isinstance(v, float):
        if not v.is_integer():
            raise LuaError("number has no integer representation")
        v = int(v)
    return v


def _wrap64(v):
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >= (1 << 63) else v


def _typename(v):
    if v is None:
        return "nil"
    if isinstance(v, bool):
        return "boolean"
    if isinstance(v, (int, float)):
        return "number"
    if isinstance(v, bytes):
        return "string"
    if isinstance(v, LuaTable):
        return "table"
    return "function"


class Interpreter:
    def __init__(self):
        self.globals = Env()
        self.string_lib = LuaTable()
        self._install_library()

    # -- public --
    def run(self, source, chunkname="chunk"):
        ast = Parser(lex(source)).block()
        fn = LuaFunction([], True, ast, self.globals, chunkname)
        return self.call(fn, [])

    def set_global(self, name, value):
        self.globals.vars[name] = value

    def get_global(self, name):
        return self.globals.vars.get(name)

    def call(self, fn, args):
        """-> list of results"""
        if isinstance(fn, LuaFunction):
            env = Env(fn.env)
            for i, p in enumerate(fn.params):
                env.vars[p] = args[i] if i < len(args) else None
            varargs = list(args[len(fn.params):]) if fn.vararg else []
            try:
                self.exec_block(fn.body, env, varargs)
            except _Return as r:
                return r.values
            return []
        if callable(fn):
            r = fn(*args)
            if r is None:
                return []
            return list(r) if isinstance(r, (list, tuple)) else [r]
        raise LuaError(f"attempt to call a {_typename(fn)} value")

    # -- statements --
    def exec_block(self, stmts, env, varargs):
        for s in stmts:
            self.exec(s, env, varargs)

    def exec(self, s, env, va):
        kind = s[0]
        if kind == "local":
            vals = self.eval_list(s[2], env, va)
            for i, n in enumerate(s[1]):
                env.vars[n] = vals[i] if i < len(vals) else None
        elif kind == "assign":
            vals = self.eval_list(s[2], env, va)
            # targets are resolved left to right before any store (Lua leaves the order open; this text does not depend on it)
            refs = []
            for tg in s[1]:
                if tg[0] == "name":
                    refs.append(("n", tg[1]))
                else:
                    refs.append(("i", self.eval(tg[1], env, va), self.eval(tg[2], env, va)))
            for i, r in enumerate(refs):
                v = vals[i] if i < len(vals) else None
                if r[0] == "n":
                    e = env.lookup(r[1]) or self.globals
                    e.vars[r[1]] = v
                else:
                    self.setindex(r[1], r[2], v)
        elif kind == "callstat":
            self.eval_multi(s[1], env, va)
        elif kind == "localfunc":
            env.vars[s[1]] = None
            env.vars[s[1]] = self.eval(s[2], env, va)
        elif kind == "if":
            for cond, body in s[1]:
                if _truthy(self.eval(cond, env, va)):
                    self.exec_block(body, Env(env), va)
                    return
            if s[2] is not None:
                self.exec_block(s[2], Env(env), va)
        elif kind == "while":
            try:
                while _truthy(self.eval(s[1], env, va)):
                    self.exec_block(s[2], Env(env), va)
            except _Break:
                pass
        elif kind == "repeat":
            try:
                while True:
                    inner = Env(env)
                    self.exec_block(s[1], inner, va)
                    if _truthy(self.eval(s[2], inner, va)):
                        break
            except _Break:
                pass
        elif kind == "do":
            self.exec_block(s[1], Env(env), va)
        elif kind == "fornum":
            a, b = self.eval(s[2], env, va), self.eval(s[3], env, va)
            c = self.eval(s[4], env, va) if s[4] is not None else 1
            a, b, c = _arith_operand(a, "for"), _arith_operand(b, "for"), _arith_operand(c, "for")
            if c == 0:
                raise LuaError("'for' step is zero")
            try:
                i = a
                while (i <= b) if c > 0 else (i >= b):
                    inner = Env(env)
                    inner.vars[s[1]] = i
                    self.exec_block(s[5], inner, va)
                    i += c
            except _Break:
                pass
        elif kind == "forin":
            vals = self.eval_list(s[2], env, va)
            f, st, ctl = (vals + [None, None, None])[:3]
            try:
                while True:
                    rs = self.call(f, [st, ctl])
                    if not rs or rs[0] is None:
                        break
                    ctl = rs[0]
                    inner = Env(env)
                    for i, n in enumerate(s[1]):
                        inner.vars[n] = rs[i] if i < len(rs) else None
                    self.exec_block(s[3], inner, va)
            except _Break:
                pass
        elif kind == "return":
            raise _Return(self.eval_list(s[1], env, va))
        elif kind == "break":
            raise _Break()
        else:
            raise LuaError(f"minilua: unsupported statement {kind}")

    # -- expressions --
    def eval_list(self, exprs, env, va):
        out = []
        for i, e in enumerate(exprs):
            if i == len(exprs) - 1 and e[0] in ("call", "method", "vararg"):
                out.extend(self.eval_multi(e, env, va))
            else:
                out.append(self.eval(e, env, va))
        return out

    def eval_multi(self, e, env, va):
        if e[0] == "call":
            return self.call(self.eval(e[1], env, va), self.eval_list(e[2], env, va))
        if e[0] == "method":
            obj = self.eval(e[1], env, va)
            return self.call(self.index(obj, e[2]), [obj] + self.eval_list(e[3], env, va))
        if e[0] == "vararg":
            return list(va)
        return [self.eval(e, env, va)]

    def index(self, obj, key):
        if isinstance(obj, LuaTable):
            return obj.get(key)
        if isinstance(obj, bytes):
            return self.string_lib.get(key)
        raise LuaError(f"attempt to index a {_typename(obj)} value (key {key!r})")

    def setindex(self, obj, key, val):
        if not isinstance(obj, LuaTable):
            raise LuaError(f"attempt to index a {_typename(obj)} value (key {key!r})")
        obj.set(key, val)

    def eval(self, e, env, va):
        kind = e[0]
        if kind == "const":
            return e[1]
        if kind == "name":
            holder = env.lookup(e[1])
            return holder.vars[e[1]] if holder else None
        if kind == "index":
            return self.index(self.eval(e[1], env, va), self.eval(e[2], env, va))
        if kind in ("call", "method", "vararg"):
            r = self.eval_multi(e, env, va)
            return r[0] if r else None
        if kind == "paren":
            return self.eval(e[1], env, va)
        if kind == "function":
            return LuaFunction(e[1], e[2], e[3], env, e[4])
        if kind == "table":
            t, pos = LuaTable(), 1
            for j, it in enumerate(e[1]):
                if it[0] == "kv":
                    t.set(self.eval(it[1], env, va), self.eval(it[2], env, va))
                elif j == len(e[1]) - 1 and it[1][0] in ("call", "method", "vararg"):
                    for v in self.eval_multi(it[1], env, va):
                        t.set(pos, v)
                        pos += 1
                else:
                    t.set(pos, self.eval(it[1], env, va))
                    pos += 1
            return t
        if kind == "unop":
            v = self.eval(e[2], env, va)
            op = e[1]
            if op == "not":
                return not _truthy(v)
            if op == "#":
                if isinstance(v, bytes):
                    return len(v)
                if isinstance(v, LuaTable):
                    return v.length()
                raise LuaError(f"attempt to get length of a {_typename(v)} value")
            if op == "-":
                return -_arith_operand(v, "unm")
            return _wrap64(~_int_operand(v, "bnot"))
        if kind == "binop":
            op = e[1]
            if op == "and":
                l = self.eval(e[2], env, va)
                return self.eval(e[3], env, va) if _truthy(l) else l
            if op == "or":
                l = self.eval(e[2], env, va)
                return l if _truthy(l) else self.eval(e[3], env, va)
            l, r = self.eval(e[2], env, va), self.eval(e[3], env, va)
            return self.binop(op, l, r)
        raise LuaError(f"minilua: unsupported expression {kind}")

    def binop(self, op, l, r):
        if op == "==":
            return self.rawequal(l, r)
        if op == "~=":
            return not self.rawequal(l, r)
        if op in ("<", "<=", ">", ">="):
            num = isinstance(l, (int, float)) and isinstance(r, (int, float)) and not isinstance(l, bool) and not isinstance(r, bool)
            if not (num or (isinstance(l, bytes) and isinstance(r, bytes))):
                raise LuaError(f"attempt to compare {_typename(l)} with {_typename(r)}")
            return {"<": l < r, "<=": l <= r, ">": l > r, ">=": l >= r}[op]
        if op == "..":
            if not isinstance(l, (bytes, int, float)) or not isinstance(r, (bytes, int, float)) or isinstance(l, bool) or isinstance(r, bool):
                raise LuaError(f"attempt to concatenate a {_typename(l if not isinstance(l, (bytes, int, float)) else r)} value")
            return _tostr(l) + _tostr(r)
        if op in ("&", "|", "~", "<<", ">>"):
            a, b = _int_operand(l, op), _int_operand(r, op)
            if op == "&":
                return _wrap64(a & b)
            if op == "|":
                return _wrap64(a | b)
            if op == "~":
                return _wrap64(a ^ b)
            if op == "<<":
                a, b = (a, b) if b >= 0 else (a, b)
                return _wrap64((a & ((1 << 64) - 1)) << b) if 0 <= b < 64 else (self.binop(">>", a, -b) if b < 0 else 0)
            return _wrap64((a & ((1 << 64) - 1)) >> b) if 0 <= b < 64 else (self.binop("<<", a, -b) if b < 0 else 0)
        a, b = _arith_operand(l, op), _arith_operand(r, op)
        both_int = isinstance(a, int) and isinstance(b, int)
        if op == "+":
            return _wrap64(a + b) if both_int else a + b
        if op == "-":
            return _wrap64(a - b) if both_int else a - b
        if op == "*":
            return _wrap64(a * b) if both_int else a * b
        if op == "/":
            return float(a) / float(b)
        if op == "//":
            if both_int:
                if b == 0:
                    raise LuaError("attempt to perform 'n//0'")
                return a // b
            return float(a) // float(b)
        if op == "%":
            if both_int:
                if b == 0:
                    raise LuaError("attempt to perform 'n%%0'")
                return a % b
            return float(a) % float(b)
        if op == "^":
            return float(a) ** float(b)
        raise LuaError(f"minilua: unsupported operator {op}")

    @staticmethod
    def rawequal(l, r):
        if isinstance(l, bool) or isinstance(r, bool):
            return l is r
        if isinstance(l, (int, float)) and isinstance(r, (int, float)):
            return l == r
        if isinstance(l, bytes) and isinstance(r, bytes):
            return l == r
        return l is r

    # -- library --
    def _install_library(self):
        G = self.globals.vars
        S, T, M = self.string_lib, LuaTable(), LuaTable()

        def lua_sub(s, i=1, j=-1):
            n = len(s)
            i, j = _int_operand(i, "sub"), _int_operand(j, "sub")
            if i < 0:
                i = max(n + i + 1, 1)
            elif i == 0:
                i = 1
            if j < 0:
                j = n + j + 1
            elif j > n:
                j = n
            return s[i - 1:j] if i <= j else b""

        def lua_byte(s, i=1, j=None):
            j = i if j is None else j
            return [b for b in lua_sub(s, i, j)]

        def pack_fmt(fmt):
            fmt = fmt.decode()
            if not re.fullmatch(r"[<>=]?(?:[iI][1248]|[bBhHlLjJ])*", fmt) or ">" in fmt:
                raise LuaError(f"minilua: unsupported string.pack format {fmt!r} (little-endian fixed-size integers only)")
            codes = {"i1": "b", "I1": "B", "i2": "h", "I2": "H", "i4": "i", "I4": "I", "i8": "q", "I8": "Q", "b": "b", "B": "B",
                     "h": "h", "H": "H", "l": "q", "L": "Q", "j": "q", "J": "Q"}
            return "<" + "".join(codes[m] for m in re.findall(r"[iI][1248]|[bBhHlLjJ]", fmt))

        def lua_pack(fmt, *vals):
            st = pack_fmt(fmt)
            try:
                return struct.pack(st, *[_int_operand(v, "pack") for v in vals])
            except struct.error as ex:
                raise LuaError(f"bad argument to 'pack' ({ex})")

        def lua_unpack(fmt, s, pos=1):
            st = pack_fmt(fmt)
            size = struct.calcsize(st)
            return list(struct.unpack_from(st, s, pos - 1)) + [pos + size]

        def lua_format(fmt, *args):
            out, it = bytearray(), iter(args)
            i = 0
            while i < len(fmt):
                c = fmt[i:i + 1]
                if c != b"%":
                    out += c
                    i += 1
                    continue
                d = fmt[i + 1:i + 2]
                if d == b"%":
                    out += b"%"
                elif d == b"d":
                    out += str(_int_operand(next(it), "format")).encode()
                elif d == b"s":
                    out += _tostr(next(it))
                else:
                    raise LuaError("minilua: unsupported string.format directive %" + d.decode())
                i += 2
            return bytes(out)

        for name, fn in (("sub", lua_sub), ("byte", lua_byte), ("len", lambda s: len(s)), ("pack", lua_pack),
                         ("unpack", lua_unpack), ("format", lua_format),
                         ("rep", lambda s, n, sep=b"": sep.join([s] * max(0, _int_operand(n, "rep")))),
                         ("char", lambda *cs: bytes(_int_operand(c, "char") for c in cs)),
                         ("upper", lambda s: s.upper()), ("lower", lambda s: s.lower())):
            S.set(name.encode(), fn)

        def t_concat(t, sep=b"", i=1, j=None):
            j = t.length() if j is None else j
            parts = []
            for k in range(i, j + 1):
                v = t.get(k)
                if not isinstance(v, (bytes, int, float)) or isinstance(v, bool):
                    raise LuaError(f"invalid value (at index {k}) in table for 'concat'")
                parts.append(_tostr(v))
            return sep.join(parts)

        def t_insert(t, *a):
            if len(a) == 1:
                t.set(t.length() + 1, a[0])
            else:
                pos, v = _int_operand(a[0], "insert"), a[1]
                for k in range(t.length(), pos - 1, -1):
                    t.set(k + 1, t.get(k))
                t.set(pos, v)

        T.set(b"concat", t_concat)
        T.set(b"insert", t_insert)
        T.set(b"unpack", lambda t, i=1, j=None: [t.get(k) for k in range(i, (t.length() if j is None else j) + 1)])
        M.set(b"floor", lambda x: int(x // 1))
        M.set(b"max", lambda *a: max(a))
        M.set(b"min", lambda *a: min(a))
        M.set(b"huge", float("inf"))
        M.set(b"maxinteger", (1 << 63) - 1)

        def lua_next(t, k=None):
            keys = list(t.h.keys())
            if k is None:
                return [keys[0], t.h[keys[0]]] if keys else [None]
            i = keys.index(k) + 1
            return [keys[i], t.h[keys[i]]] if i < len(keys) else [None]

        def ipairs_iter(t, i):
            v = t.get(i + 1)
            return [None] if v is None else [i + 1, v]

        def lua_select(n, *a):
            if n == b"#":
                return len(a)
            return list(a[_int_operand(n, "select") - 1:])

        def lua_error(msg=None, *_):
            raise LuaError(_tostr(msg).decode("latin1"))

        def lua_assert(v=None, msg=b"assertion failed!", *rest):
            if not _truthy(v):
                raise LuaError(_tostr(msg).decode("latin1"))
            return [v, msg] + list(rest)

        def lua_tonumber(v, base=None):
            if isinstance(v, (int, float)) and not isinstance(v, bool):
                return v
            if isinstance(v, bytes):
                try:
                    return int(v, base or 10) if base or re.fullmatch(rb"\s*-?\d+\s*", v) else float(v)
                except ValueError:
                    return [None]
            return [None]

        G.update({"string": S, "table": T, "math": M, "pairs": lambda t: [lua_next, t, None],
                  "ipairs": lambda t: [ipairs_iter, t, 0], "next": lua_next, "select": lua_select,
                  "type": lambda v: _typename(v).encode(), "tostring": _tostr, "tonumber": lua_tonumber,
                  "assert": lua_assert, "error": lua_error, "rawequal": self.rawequal,
                  "print": lambda *a: print(*[_tostr(x).decode("latin1") for x in a]),
                  "unpack": T.get(b"unpack")})


# ---- conveniences for tests -----------------------------------------------------------------------------------------
def to_python(v):
    """LuaTable -> list (1..n border) or dict (otherwise), recursively; bytes and numbers unchanged"""
    if isinstance(v, LuaTable):
        n = v.length()
        if n == len(v.h):
            return [to_python(v.get(i)) for i in range(1, n + 1)]
        return {k: to_python(x) for k, x in v.h.items()}
    return v


def table_from(obj):
    """dict / list -> LuaTable (str keys become Lua strings)"""
    t = LuaTable()
    if isinstance(obj, dict):
        for k, v in obj.items():
            t.set(k.encode() if isinstance(k, str) else k, v)
    else:
        for i, v in enumerate(obj):
            t.set(i + 1, v)
    return t
