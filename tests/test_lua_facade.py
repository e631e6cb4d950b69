"""The Lua FACADE text (integration/lua/tfhe_gates.lua: what a maintainer appends to ao-tfhe/tfhe.lua, SURVEY.md 8 f3/f4;
pattern /root/reference/ao-tfhe/tfhe.lua:4-53) EXECUTED -- by tests/lua_double/minilua.py, a small interpreter for the
subset of Lua 5.3 the file is written in, because the build image has no Lua.  This runs THIS REPOSITORY'S Lua text; it
is not a Lua VM and proves nothing about one beyond the manual's semantics restated in minilua.py.

  * the interpreter itself against hand-computed results of the language features the facade relies on;
  * every netlist builder of the facade, evaluated gate by gate on plaintext bits over all / random inputs
    (adder, adder with constant carry-in, equality, min / max, subtractor, multiplier) and its bootstrap count;
  * runNetlist / planes / the *BitsBatch functions and encryptStringBits against a plaintext stand-in of the backend
    (same call signatures and string formats as the C binding);
  * GPU: the facade on top of the REAL binding C (through the Lua C-API double) on top of the real library:
    generateGateKey, encryptStringBits + equalStrings, addBitsBatch, minMaxBitsBatch -- decrypted against plaintext, and one
    circuit's bytes against the oracle.
"""
import os
import struct
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "lua_double"))
import minilua as ml  # noqa: E402

FACADE = os.path.join(ROOT, "integration", "lua", "tfhe_gates.lua")
OPS = dict(NAND=0, AND=1, OR=2, NOR=3, XOR=4, XNOR=5, ANDNY=6, ANDYN=7, ORNY=8, ORYN=9, MUX=10, NOT=11, COPY=12, CONST0=13,
           CONST1=14, MAJ=15, XOR3=16)


# ---- the interpreter ------------------------------------------------------------------------------------------------
def run(src):
    it = ml.Interpreter()
    return [ml.to_python(v) for v in it.run(src)]


def test_minilua_language_features_the_facade_uses():
    assert run("return 7 // 2, -7 // 2, 7 % 3, -7 % 3, 2^10, 7 / 2, (5 >> 1) & 1, 1 << 4, 6 ~ 3, ~0") == \
        [3, -4, 1, 2, 1024.0, 3.5, 0, 16, 5, -1]
    assert run("local t = {10, 20, 30, x = 1, [0] = 5}; t[#t + 1] = 40; return #t, t[0], t.x, #'abc' .. 'd'") == [4, 5, 1, b"3d"]
    assert run("local function f(...) local a, b = ...; return b, a, select('#', ...) end; return f(1, 2, 3)") == [2, 1, 3]
    assert run("local function f() return 1, 2 end; local t = {f(), f()}; local a, b, c = f(); return #t, (f()), c") == [3, 1, None]
    # closures over a local table (newNetlist's pattern), method-style definitions, multiple assignment
    assert run("""
        local function mk() local o = { n = 0 }; function o.inc(k) local w = o.n; o.n = o.n + (k or 1); return w end; return o end
        local a, b = mk(), mk(); a.inc(3); a.inc(); b.inc()
        local x, y = 1, 2; x, y = y, x
        return a.n, b.n, x, y""") == [4, 1, 2, 1]
    assert run("local s = 0; for i = 1, 9, 2 do s = s + i end; for i = 3, 1, -1 do s = s * 2 end; return s") == [200]
    assert run("local i, n = 0, 0; while true do i = i + 1; if i > 5 then break elseif i % 2 == 0 then n = n + i end end; return n") == [6]
    assert run("return nil and 1, false or 'x', 1 and 2, nil == false, 1 == 1.0, 'a' < 'b', not nil") == \
        [None, b"x", 2, False, True, True, True]
    assert run("return string.pack('<i4i4', 1, -1), ('abc'):sub(2), ('hello'):sub(2, 3), ('AB'):byte(2), string.char(72, 105), "
               "string.rep('\\0', 3), table.concat({'a', 'b', 3}, ',')") == \
        [struct.pack("<ii", 1, -1), b"bc", b"el", 66, b"Hi", b"\0\0\0", b"a,b,3"]
    assert run("local t = {}; for k, v in pairs({a = 1, b = 2}) do t[#t + 1] = k end; local s = 0; for _, v in ipairs({4, 5}) do s = s + v end; return #t, s") == [2, 9]
    with pytest.raises(ml.LuaError, match="arithmetic"):
        run("return {} + 1")
    with pytest.raises(ml.LuaError, match="index a nil"):
        run("local t; return t.x")
    with pytest.raises(ml.LuaError):
        run("goto done")                        # outside the subset: refused, not guessed


# ---- the facade on a plaintext stand-in of the backend ---------------------------------------------------------------
ROW = 3                                            # ints per "sample" of the stand-in: the last one carries the bit


def gate_eval(op, a, b, c):
    if op == OPS["NAND"]: return 1 - (a & b)
    if op == OPS["AND"]: return a & b
    if op == OPS["OR"]: return a | b
    if op == OPS["NOR"]: return 1 - (a | b)
    if op == OPS["XOR"]: return a ^ b
    if op == OPS["XNOR"]: return 1 - (a ^ b)
    if op == OPS["ANDNY"]: return (1 - a) & b
    if op == OPS["ANDYN"]: return a & (1 - b)
    if op == OPS["ORNY"]: return (1 - a) | b
    if op == OPS["ORYN"]: return a | (1 - b)
    if op == OPS["MUX"]: return np.where(a == 1, b, c)
    if op == OPS["NOT"]: return 1 - a
    if op == OPS["COPY"]: return a
    if op == OPS["CONST0"]: return np.zeros_like(a)
    if op == OPS["MAJ"]: return ((a + b + c) >= 2).astype(a.dtype)
    if op == OPS["XOR3"]: return a ^ b ^ c
    return np.ones_like(a)


def run_packed(packed, bits):
    """bits [nWires][instances]; evaluates the packed netlist (5 int32 per gate) in order, in place"""
    g = np.frombuffer(packed, "<i4").reshape(-1, 5)
    z = np.zeros_like(bits[0])
    for op, i0, i1, i2, out in g:
        bits[out] = gate_eval(op, bits[i0] if i0 >= 0 else z, bits[i1] if i1 >= 0 else z, bits[i2] if i2 >= 0 else z)
    return len(g), int(sum(2 if op == OPS["MUX"] else (0 if OPS["NOT"] <= op <= OPS["CONST1"] else 1) for op, *_ in g))


class PlainBackend:
    """same signatures and string formats as the C binding's l_* entries, on plaintext 'samples'"""

    def __init__(self):
        self.calls = []

    def table(self):
        def circuit_run(gates, wires, n_wires, instances):
            self.calls.append(("circuitRun", len(gates) // 20, n_wires, instances))
            if len(wires) != n_wires * instances * ROW * 4:
                return None
            w = np.frombuffer(wires, "<i4").reshape(n_wires, instances, ROW).copy()
            bits = w[:, :, -1].copy()
            run_packed(gates, bits)
            w[:, :, -1] = bits
            return w.tobytes()

        def encrypt_bits(bits):
            self.calls.append(("encryptBits", bits))
            out = np.zeros((len(bits), ROW), "<i4")
            out[:, -1] = np.frombuffer(bits, np.uint8)
            return out.tobytes()
        def netlist_cost(gates, instances):
            """the binding's l_netlistCost on the Python twin of eoc_netlist_cost (tests/test_circuits_cpu.py compares the two)"""
            from eoc_tfhe_amd import Gate, circuits
            g = [Gate(*map(int, row)) for row in np.frombuffer(gates, "<i4").reshape(-1, 5)]
            self.calls.append(("netlistCost", len(g), instances))
            return circuits.netlist_cost(g, instances, 1024)
        def netlist_optimize(gates, outputs):
            """the binding's l_netlistOptimize on the Python twin of eoc_netlist_optimize"""
            from eoc_tfhe_amd import Gate, circuits
            g = [Gate(*map(int, row)) for row in np.frombuffer(gates, "<i4").reshape(-1, 5)]
            outs = [int(v) for v in np.frombuffer(outputs, "<i4")]
            self.calls.append(("netlistOptimize", len(g), len(outs)))
            return b"".join(struct.pack("<5i", x.op, x.in0, x.in1, x.in2, x.out) for x in circuits.optimize(g, outs))
        return ml.table_from({"sampleInts": lambda: ROW, "circuitRun": circuit_run, "encryptBits": encrypt_bits,
                              "netlistCost": netlist_cost, "netlistOptimize": netlist_optimize,
                              "gateNAND": lambda a, b, pk=None: b"NAND(" + a + b"," + b + b")",
                              "setDevices": lambda *d: len(d)})


@pytest.fixture()
def facade():
    it = ml.Interpreter()
    be = PlainBackend()
    tf = ml.LuaTable()
    tf.set(b"backend", be.table())
    it.set_global("Tfhe", tf)                       # ao-tfhe/tfhe.lua:1-3 creates this table; the text below is appended to it
    it.run(open(FACADE, "rb").read(), "tfhe_gates.lua")
    return it, tf, be


def call(it, tf, name, *args):
    return it.call(tf.get(name.encode()), list(args))


def planes_of(values, nbits, instances):
    """LSB-first bit planes [nbits][instances][ROW] as the facade's operand strings"""
    w = np.zeros((nbits, instances, ROW), "<i4")
    for i in range(nbits):
        w[i, :, -1] = (values >> i) & 1
    return w.tobytes()


def value_of(buf, instances):
    w = np.frombuffer(buf, "<i4").reshape(-1, instances, ROW)[:, :, -1]
    return sum(w[i].astype(np.int64) << i for i in range(w.shape[0]))


def test_facade_defines_its_functions_and_passes_through(facade):
    it, tf, be = facade
    for name in ("generateGateKey", "nand", "band", "bor", "bnot", "mux", "adderNetlist", "equalNetlist", "minMaxNetlist",
                 "newCircuit", "muxAdderNetlist", "majAdderNetlist", "majSubtractorNetlist", "majLessThanNetlist", "maj", "xor3", "prefixAdderNetlist", "prefixSubtractorNetlist", "wallaceMultiplierNetlist", "multiplierNetlistFor", "subtractorNetlistFor", "lessThanNetlist", "lessThanTreeNetlist", "adderNetlistFor", "lessThanNetlistFor",
                 "subtractorNetlist", "multiplierNetlist", "runNetlist", "addBitsBatch", "subtractBitsBatch",
                 "multiplyBitsBatch", "minMaxBitsBatch", "lessThanBitsBatch", "equalBits", "equalStrings", "encryptStringBits", "setDevices"):
        assert isinstance(tf.get(name.encode()), ml.LuaFunction), name
    assert call(it, tf, "nand", b"x", b"y") == [b"NAND(x,y)"]
    assert call(it, tf, "setDevices", 0, 1, 2) == [3]                          # varargs travel
    assert ml.to_python(tf.get(b"OP"))[b"MUX"] == 10


@pytest.mark.parametrize("nbits", [1, 2, 3, 5])
def test_netlist_builders_on_plaintext(facade, nbits):
    it, tf, be = facade
    vals = np.arange(1 << nbits)
    A, B = [x.ravel() for x in np.meshgrid(vals, vals)]                           # every input pair
    S = len(A)

    def evaluate(nl, a, b):
        packed = it.call(nl.get(b"packed"), [])[0]
        bits = np.zeros((nl.get(b"nWires"), S), np.int64)
        for i in range(nbits):
            bits[a + i], bits[b + i] = (A >> i) & 1, (B >> i) & 1
        return bits, run_packed(packed, bits)

    def word(bits, wires):
        return sum(bits[w] << i for i, w in enumerate(wires))

    nl, a, b, s = call(it, tf, "adderNetlist", nbits)
    bits, (ngates, boots) = evaluate(nl, a, b)
    assert np.array_equal(word(bits, ml.to_python(s)), A + B) and boots == max(2, 5 * nbits - 3)
    nl, a, b, s = call(it, tf, "adderNetlist", nbits, True)
    bits, (ngates, boots) = evaluate(nl, a, b)
    assert np.array_equal(word(bits, ml.to_python(s)), A + B) and boots == 5 * nbits    # BASELINE.md's 5 gates per bit
    nl, x, y, out = call(it, tf, "equalNetlist", nbits)
    bits, (ngates, boots) = evaluate(nl, x, y)
    assert np.array_equal(bits[out], (A == B).astype(np.int64)) and boots == 2 * nbits - 1
    nl, a, b, lt, mn, mx = call(it, tf, "minMaxNetlist", nbits)
    bits, _ = evaluate(nl, a, b)
    assert np.array_equal(bits[lt], (A < B).astype(np.int64))
    assert np.array_equal(word(bits, ml.to_python(mn)), np.minimum(A, B))
    assert np.array_equal(word(bits, ml.to_python(mx)), np.maximum(A, B))
    nl, a, b, diff, br = call(it, tf, "subtractorNetlist", nbits)
    bits, (ngates, boots) = evaluate(nl, a, b)
    assert np.array_equal(word(bits, ml.to_python(diff)), (A - B) % (1 << nbits)) and np.array_equal(bits[br], (A < B).astype(np.int64))
    assert boots == 2 + 4 * (nbits - 1)
    nl, a, b, prod = call(it, tf, "multiplierNetlist", nbits)
    bits, _ = evaluate(nl, a, b)
    prod = ml.to_python(prod)
    assert len(prod) == 2 * nbits and np.array_equal(word(bits, prod), A * B)
    # round 6: the forms picked by instance count -- gate for gate the Python circuit layer's netlists
    from eoc_tfhe_amd import circuits
    for lua_name, py in (("muxAdderNetlist", circuits.mux_carry_adder), ("prefixAdderNetlist", circuits.prefix_adder),
                         ("majAdderNetlist", circuits.maj_adder)):
        nl, a, b, s = call(it, tf, lua_name, nbits)
        bits, (ngates, boots) = evaluate(nl, a, b)
        assert np.array_equal(word(bits, ml.to_python(s)), A + B), lua_name
        pg = py(nbits)[0]
        assert (ngates, boots) == (len(pg), sum(2 if g.op == 10 else 0 if 11 <= g.op <= 14 else 1 for g in pg)), lua_name
    nl, a, b, prod = call(it, tf, "wallaceMultiplierNetlist", nbits)
    bits, (ngates, boots) = evaluate(nl, a, b)
    assert np.array_equal(word(bits, ml.to_python(prod)), A * B)
    pg = circuits.wallace_multiplier(nbits)[0]
    assert (ngates, boots) == (len(pg), sum(2 if g.op == 10 else 0 if 11 <= g.op <= 14 else 1 for g in pg))
    # ... gate for gate the Python builder's netlist (same wire numbering: the (level, wire) order of a column is part of it)
    assert it.call(nl.get(b"packed"), [])[0] == b"".join(struct.pack("<5i", g.op, g.in0, g.in1, g.in2, g.out) for g in pg)
    nl, a, b, diff, br = call(it, tf, "majSubtractorNetlist", nbits)
    bits, (ngates, boots) = evaluate(nl, a, b)
    assert np.array_equal(word(bits, ml.to_python(diff)), (A - B) % (1 << nbits)) and np.array_equal(bits[br], (A < B).astype(np.int64))
    assert boots == 2 * nbits
    nl, a, b, lt = call(it, tf, "majLessThanNetlist", nbits)
    bits, (ngates, boots) = evaluate(nl, a, b)
    assert np.array_equal(bits[lt], (A < B).astype(np.int64)) and boots == nbits
    nl, a, b, diff, br = call(it, tf, "prefixSubtractorNetlist", nbits)
    bits, (ngates, boots) = evaluate(nl, a, b)
    assert np.array_equal(word(bits, ml.to_python(diff)), (A - B) % (1 << nbits)) and np.array_equal(bits[br], (A < B).astype(np.int64))
    pg = circuits.prefix_subtractor(nbits)[0]
    assert (ngates, boots) == (len(pg), sum(2 if g.op == 10 else 0 if 11 <= g.op <= 14 else 1 for g in pg))
    nl, a, b, lt = call(it, tf, "lessThanTreeNetlist", nbits)
    bits, (ngates, boots) = evaluate(nl, a, b)
    assert np.array_equal(bits[lt], (A < B).astype(np.int64))
    pg = circuits.less_than_tree(nbits)[0]
    assert (ngates, boots) == (len(pg), sum(2 if g.op == 10 else 0 if 11 <= g.op <= 14 else 1 for g in pg))


def test_eight_bit_netlists_match_the_python_circuit_layer(facade):
    """the facade's netlists compute what eoc_tfhe_amd/circuits.py's compute, with the same bootstrap counts"""
    import eoc_tfhe_amd as eoc
    from eoc_tfhe_amd import circuits
    it, tf, be = facade
    rng = np.random.default_rng(4)
    A, B = rng.integers(0, 256, 200), rng.integers(0, 256, 200)
    for lua_name, largs, py in (("adderNetlist", (8, True), circuits.ripple_carry_adder(8, carry_in_zero=True)),
                                ("subtractorNetlist", (8,), circuits.subtractor(8)), ("multiplierNetlist", (8,), circuits.multiplier(8))):
        res = call(it, tf, lua_name, *largs)
        nl, a, b, outs = res[0], res[1], res[2], ml.to_python(res[3])
        bits = np.zeros((nl.get(b"nWires"), len(A)), np.int64)
        for i in range(8):
            bits[a + i], bits[b + i] = (A >> i) & 1, (B >> i) & 1
        _, boots = run_packed(it.call(nl.get(b"packed"), [])[0], bits)
        got = sum(bits[w] << i for i, w in enumerate(outs))
        want = {"adderNetlist": A + B, "subtractorNetlist": (A - B) % 256, "multiplierNetlist": A * B}[lua_name]
        assert np.array_equal(got, want), lua_name
        assert boots == eoc.circuit_bootstraps(py[0]), (lua_name, boots)


def test_forms_are_picked_by_instance_count(facade):
    """Tfhe.adderNetlistFor / lessThanNetlistFor ask the backend's level-cost estimate (eoc_netlist_cost): one instance or
    a handful take the log-depth forms (5 and 4 levels at 8 bits), thousands the forms with the fewest bootstraps"""
    from eoc_tfhe_amd import Gate, circuits
    it, tf, be = facade

    def shape_of(g):
        return sum(2 if x.op == 10 else 0 if 11 <= x.op <= 14 else 1 for x in g), circuits.bootstrap_depth(g)

    def shape(nl):
        return shape_of([Gate(*map(int, row)) for row in np.frombuffer(it.call(nl.get(b"packed"), [])[0], "<i4").reshape(-1, 5)])

    for inst, add_want, lt_want in ((1, (48, 5), (29, 4)), (8, (48, 5), (29, 4)), (4096, (16, 8), (8, 8))):
        assert shape(call(it, tf, "adderNetlistFor", 8, inst)[0]) == add_want, inst
        assert shape(call(it, tf, "lessThanNetlistFor", 8, inst)[0]) == lt_want, inst
    assert shape(call(it, tf, "multiplierNetlistFor", 8, 2)[0]) == (244, 11)
    assert shape(call(it, tf, "multiplierNetlistFor", 8, 4096)[0]) == (320, 40)
    assert shape(call(it, tf, "subtractorNetlistFor", 8, 2)[0]) == (48, 5)
    assert shape(call(it, tf, "subtractorNetlistFor", 8, 4096)[0]) == (16, 8)
    assert shape(call(it, tf, "minMaxNetlistFor", 8, 1)[0]) == (29 + 32, 5)
    assert shape(call(it, tf, "minMaxNetlistFor", 8, 4096)[0]) == (8 + 16 + 8, 10)      # MAJ chain, min by MUX, max = XOR3(a, b, min)
    A = np.array([200, 13, 255]); B = np.array([100, 250, 255])
    lo, hi, lt = call(it, tf, "minMaxBitsBatch", planes_of(A, 8, 3), planes_of(B, 8, 3), 8, 3)
    assert np.array_equal(value_of(lo, 3), np.minimum(A, B)) and np.array_equal(value_of(hi, 3), np.maximum(A, B))
    assert np.array_equal(value_of(lt, 3), (A < B).astype(np.int64))
    assert np.array_equal(value_of(call(it, tf, "lessThanBitsBatch", planes_of(A, 8, 3), planes_of(B, 8, 3), 8, 3)[0], 3),
                          (A < B).astype(np.int64))
    out = call(it, tf, "addBitsBatch", planes_of(A, 8, 3), planes_of(B, 8, 3), 8, 3)[0]
    assert np.array_equal(value_of(out, 3), A + B)
    # what runs is the picked form AFTER netlistOptimize (the prefix adder: 48 -> 40 bootstraps on the same 5 levels)
    ran = circuits.adder(8, 3)[0]
    assert be.calls[-1][:2] == ("circuitRun", len(ran)) and shape_of(ran) == (40, 5)


def test_deferred_circuit_records_gate_calls_and_runs_them_in_one_backend_call(facade):
    """Tfhe.newCircuit: the reference's call style (one ciphertext operation per Lua call) recorded on wire handles and
    evaluated by ONE circuitRun (after netlistOptimize) -- a 3-bit adder written gate by gate the textbook way, over 5
    instances from raw samples and over one instance from base64 strings"""
    import base64
    from eoc_tfhe_amd import circuits
    it, tf, be = facade
    S = 5
    A, B = np.array([3, 7, 0, 5, 6]), np.array([1, 7, 4, 2, 3])

    def build(c, xs, ys):
        cget = lambda name: c.get(name.encode())
        carry, outs = None, []
        for i in range(3):
            p = it.call(cget("xor"), [xs[i], ys[i]])[0]
            g = it.call(cget("band"), [xs[i], ys[i]])[0]
            if carry is None:
                outs.append(p)
                carry = g
            else:
                outs.append(it.call(cget("xor"), [p, carry])[0])
                carry = it.call(cget("bor"), [g, it.call(cget("band"), [p, carry])[0]])[0]
        return outs + [carry]

    c = call(it, tf, "newCircuit")[0]
    xs = [it.call(c.get(b"inputSamples"), [planes_of(A, 3, S)[i * S * ROW * 4:(i + 1) * S * ROW * 4]])[0] for i in range(3)]
    ys = [it.call(c.get(b"inputSamples"), [planes_of(B, 3, S)[i * S * ROW * 4:(i + 1) * S * ROW * 4]])[0] for i in range(3)]
    outs = build(c, xs, ys)
    assert it.call(c.get(b"gateCount"), []) == [2 + 5 * 2]
    n_calls = len(be.calls)
    res = ml.to_python(it.call(c.get(b"run"), [ml.table_from(outs)])[0])
    new_calls = [x[0] for x in be.calls[n_calls:]]
    assert new_calls == ["netlistOptimize", "circuitRun"], new_calls                      # ONE evaluation for twelve gate calls
    assert be.calls[-1][1] == 6                                    # rewritten: XOR + AND, then XOR3 + MAJ per bit
    assert np.array_equal(value_of(b"".join(res), S), A + B)
    # a second input with another instance count is refused
    assert it.call(c.get(b"inputSamples"), [planes_of(A[:2], 1, 2)]) == [None]
    # one instance, base64 strings in and out
    c1 = call(it, tf, "newCircuit")[0]

    def ct(bit):
        return base64.b64encode(np.array([0] * (ROW - 1) + [bit], "<i4").tobytes() + bytes(8))

    x, y, z = (it.call(c1.get(b"input"), [ct(v)])[0] for v in (1, 0, 1))
    m = it.call(c1.get(b"maj"), [x, y, z])[0]
    q = it.call(c1.get(b"xor3"), [x, it.call(c1.get(b"bnot"), [y])[0], it.call(c1.get(b"constant"), [1])[0]])[0]
    r = ml.to_python(it.call(c1.get(b"run"), [ml.table_from([m, q])])[0])
    bits = [int(np.frombuffer(base64.b64decode(v)[: ROW * 4], "<i4")[-1]) for v in r]
    assert bits == [1, 1 ^ 1 ^ 1]


def test_batch_functions_pack_and_slice_wires(facade):
    it, tf, be = facade
    rng = np.random.default_rng(9)
    S, nbits = 7, 4
    A, B = rng.integers(0, 16, S), rng.integers(0, 16, S)
    pa, pb = planes_of(A, nbits, S), planes_of(B, nbits, S)
    out = call(it, tf, "addBitsBatch", pa, pb, nbits, S)[0]
    assert len(out) == (nbits + 1) * S * ROW * 4 and np.array_equal(value_of(out, S), A + B)
    # the form is picked by the instance count: at 4 bits the XOR3 / MAJ adder (2 nbits gates) is as shallow as the prefix
    # form and cheaper
    assert be.calls[-1] == ("circuitRun", 2 * nbits, be.calls[-1][2], S)
    out = call(it, tf, "subtractBitsBatch", pa, pb, nbits, S)[0]
    v = value_of(out, S)
    assert np.array_equal(v & 15, (A - B) % 16) and np.array_equal(v >> 4, (A < B).astype(np.int64))
    out = call(it, tf, "multiplyBitsBatch", pa, pb, nbits, S)[0]
    assert np.array_equal(value_of(out, S), A * B)
    lo, hi, lt = call(it, tf, "minMaxBitsBatch", pa, pb, nbits, S)
    assert np.array_equal(value_of(lo, S), np.minimum(A, B)) and np.array_equal(value_of(hi, S), np.maximum(A, B))
    assert np.array_equal(value_of(lt, S), (A < B).astype(np.int64))
    # a backend that refuses (nil) comes through as nil, not as an error
    be2 = PlainBackend().table()
    be2.set(b"circuitRun", lambda *a: None)
    tf.set(b"backend", be2)
    assert call(it, tf, "addBitsBatch", pa, pb, nbits, S) == [None]


def test_strings_travel_lsb_first_and_compare(facade):
    it, tf, be = facade
    x = call(it, tf, "encryptStringBits", b"Hi")[0]
    assert be.calls[-1] == ("encryptBits", bytes([0, 0, 0, 1, 0, 0, 1, 0, 1, 0, 0, 1, 0, 1, 1, 0]))   # 'H' = 0x48, 'i' = 0x69
    y = call(it, tf, "encryptStringBits", b"Hi")[0]
    z = call(it, tf, "encryptStringBits", b"Hj")[0]
    eq = call(it, tf, "equalStrings", x, y)[0]
    ne = call(it, tf, "equalStrings", x, z)[0]
    assert len(eq) == ROW * 4 and value_of(eq, 1)[0] == 1 and value_of(ne, 1)[0] == 0


# ---- GPU: facade -> binding C (Lua C-API double) -> library ----------------------------------------------------------
@pytest.mark.gpu
def test_facade_on_the_real_binding_and_gpu(tmp_path, built_lib):
    from gpu_util import torch_cuda
    torch_cuda()
    import eoc_tfhe_amd as eoc
    import oracle_lib as ol
    from test_lua_binding import Lua
    eoc.gpu_shutdown()
    eoc.Tfhe.resetGateKey()
    lua = Lua(tmp_path)
    names = ("generateGateKey", "resetGateKey", "encryptBits", "decryptBits", "gateBatch", "circuitRun", "sampleInts", "keyMode",
             "engineCount", "gateNAND", "encryptBit", "decryptBit", "gateMUX", "gateNOT", "netlistOptimize", "netlistCost",
             "netlistDepth", "circuitBootstraps")
    backend = ml.table_from({n: (lambda *a, _n=n: lua.call(_n, *a)) for n in names})
    it = ml.Interpreter()
    tf = ml.LuaTable()
    tf.set(b"backend", backend)
    it.set_global("Tfhe", tf)
    it.run(open(FACADE, "rb").read(), "tfhe_gates.lua")
    try:
        tok = call(it, tf, "generateGateKey", 80, 9)[0]
        assert tok and call(it, tf, "keyMode") == [1]
        orc = ol.Oracle(0, 9)
        # strings: 2 x 3 characters -> 24 bit-ciphertexts each; XOR per bit + OR tree + NOT on the GPU
        x, y, z = (call(it, tf, "encryptStringBits", s)[0] for s in (b"abc", b"abc", b"abd"))
        assert len(x) == 24 * 501 * 4
        assert lua.call("decryptBits", call(it, tf, "equalStrings", x, y)[0]) == b"\x01"
        assert lua.call("decryptBits", call(it, tf, "equalStrings", x, z)[0]) == b"\x00"
        # 4-bit adder and min / max over 6 instances
        rng = np.random.default_rng(21)
        S, nbits = 6, 4
        A, B = rng.integers(0, 16, S), rng.integers(0, 16, S)

        def enc(vals):
            bits = np.stack([(vals >> i) & 1 for i in range(nbits)]).astype(np.uint8)     # [nbits][S]
            return lua.call("encryptBits", bits.tobytes())

        def dec(buf):
            bits = np.frombuffer(lua.call("decryptBits", buf), np.uint8).reshape(-1, S).astype(np.int64)
            return sum(bits[i] << i for i in range(bits.shape[0]))

        ea, eb = enc(A), enc(B)
        out = call(it, tf, "addBitsBatch", ea, eb, nbits, S)[0]
        assert np.array_equal(dec(out), A + B)
        lo, hi, lt = call(it, tf, "minMaxBitsBatch", ea, eb, nbits, S)
        assert np.array_equal(dec(lo), np.minimum(A, B)) and np.array_equal(dec(hi), np.maximum(A, B))
        assert np.array_equal(dec(lt), (A < B).astype(np.int64))
        assert np.array_equal(dec(call(it, tf, "lessThanBitsBatch", ea, eb, nbits, S)[0]), (A < B).astype(np.int64))
        # the adder's bytes against the oracle: the facade's netlist, gate by gate
        # (the form addBitsBatch picked for 6 instances: the XOR3 / MAJ adder at 4 bits)
        nl, a, b, s = call(it, tf, "adderNetlistFor", nbits, S)
        g = np.frombuffer(it.call(nl.get(b"packed"), [])[0], "<i4").reshape(-1, 5)
        assert (g[:, 0] == OPS["MAJ"]).sum() == nbits - 1 and (g[:, 0] == OPS["XOR3"]).sum() == nbits - 1
        wires = np.zeros((nl.get(b"nWires"), S, 501), np.int32)
        wires[a:a + nbits] = np.frombuffer(ea, np.int32).reshape(nbits, S, 501)
        wires[b:b + nbits] = np.frombuffer(eb, np.int32).reshape(nbits, S, 501)
        for op, i0, i1, i2, o in g:
            wires[o] = orc.gate_batch(int(op), wires[i0], wires[i1], wires[i2] if i2 >= 0 else None)
        want = np.concatenate([wires[w] for w in ml.to_python(s)]).tobytes()
        assert out == want
        # deferred gates through the real binding: MAJ / XOR3 / NOT / CONSTANT recorded on handles, ONE circuitRun on the GPU
        c = call(it, tf, "newCircuit")[0]
        e1, e0 = call(it, tf, "encryptBit", 1)[0], call(it, tf, "encryptBit", 0)[0]
        h = [it.call(c.get(b"input"), [x])[0] for x in (e1, e0, e1)]
        m = it.call(c.get(b"maj"), h)[0]
        q = it.call(c.get(b"xor3"), [h[0], it.call(c.get(b"bnot"), [h[1]])[0], it.call(c.get(b"constant"), [1])[0]])[0]
        r = ml.to_python(it.call(c.get(b"run"), [ml.table_from([m, q])])[0])
        assert [call(it, tf, "decryptBit", v)[0] for v in r] == [1, 1 ^ 1 ^ 1]
        # string-level circuits: base64 ciphertext strings in and out (the facade's own base64), one backend call each;
        # lessThanBits goes through the backend's netlistOptimize (NOT folding, MUX fusion, dead gates dropped)
        for av, bv in ((5, 6), (7, 2)):
            As = ml.table_from([call(it, tf, "encryptBit", (av >> i) & 1)[0] for i in range(3)])
            Bs = ml.table_from([call(it, tf, "encryptBit", (bv >> i) & 1)[0] for i in range(3)])
            ssum = ml.to_python(call(it, tf, "addBits", As, Bs)[0])
            assert sum(call(it, tf, "decryptBit", x)[0] << i for i, x in enumerate(ssum)) == av + bv
            assert call(it, tf, "decryptBit", call(it, tf, "lessThanBits", As, Bs)[0]) == [int(av < bv)]
            mn, mx = (ml.to_python(t) for t in call(it, tf, "minMaxBits", As, Bs))
            assert sum(call(it, tf, "decryptBit", x)[0] << i for i, x in enumerate(mn)) == min(av, bv)
            assert sum(call(it, tf, "decryptBit", x)[0] << i for i, x in enumerate(mx)) == max(av, bv)
    finally:
        lua.call("resetGateKey")
        lua.lib.ld_close(lua.S)
        eoc.gpu_shutdown()


def test_base64_and_string_level_circuits(facade):
    """the facade's own base64 (Lua 5.3 has none) against Python's, and addBits / lessThanBits / minMaxBits -- arrays of
    base64 ciphertext strings in and out, ONE backend call per circuit, like tfhe.js"""
    import base64
    it, tf, be = facade
    rng = np.random.default_rng(3)
    for n in (0, 1, 2, 3, 4, 5, 31, 100):
        raw = rng.integers(0, 256, n).astype(np.uint8).tobytes()
        enc = call(it, tf, "base64Encode", raw)[0]
        assert enc == base64.b64encode(raw)
        assert call(it, tf, "base64Decode", enc)[0] == raw
    assert call(it, tf, "base64Decode", b"QUJD!ignored")[0] == b"ABC"          # stops at the first non-alphabet byte

    def ct(bit):                                                              # export_lweSample_toStream bytes of the stand-in
        return base64.b64encode(struct.pack("<3i", 7, 7, int(bit)) + b"\0" * 8)

    def bit_of(s):
        return struct.unpack("<3i", base64.b64decode(s)[:12])[2]

    # netlistOptimize of the stand-in backend: identity (the real one is exercised on the GPU leg)
    be_t = tf.get(b"backend")
    be_t.set(b"netlistOptimize", lambda gates, outs: gates)
    for a, b in ((5, 3), (9, 12), (15, 15), (0, 1)):
        A = ml.table_from([ct((a >> i) & 1) for i in range(4)])
        B = ml.table_from([ct((b >> i) & 1) for i in range(4)])
        s = ml.to_python(call(it, tf, "addBits", A, B)[0])
        assert len(s) == 5 and sum(bit_of(x) << i for i, x in enumerate(s)) == a + b
        assert bit_of(call(it, tf, "lessThanBits", A, B)[0]) == int(a < b)
        mn, mx = (ml.to_python(t) for t in call(it, tf, "minMaxBits", A, B))
        assert sum(bit_of(x) << i for i, x in enumerate(mn)) == min(a, b)
        assert sum(bit_of(x) << i for i, x in enumerate(mx)) == max(a, b)
