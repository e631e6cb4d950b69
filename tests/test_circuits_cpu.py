"""Host-side circuit layer: plaintext semantics of the netlist builders and the
NOT-folding / MUX-fusion rewrites.  No GPU, no oracle: pure netlist logic."""
import numpy as np
import pytest

from eoc_tfhe_amd import OPS, Gate, circuit_bootstraps
from eoc_tfhe_amd import circuits as c


def _words(nbits):
    S = 1 << (2 * nbits)
    idx = np.arange(S)
    return idx & ((1 << nbits) - 1), idx >> nbits, S


def _load(w, wires, values):
    for i, wi in enumerate(wires):
        w[wi] = (values >> i) & 1


def _value(w, wires):
    return sum(w[wi].astype(np.int64) << i for i, wi in enumerate(wires))


@pytest.mark.parametrize("nbits", [1, 2, 4])
def test_adder_all_inputs(nbits):
    gates, nw, a, b, s = c.ripple_carry_adder(nbits)
    A, B, S = _words(nbits)
    w = np.zeros((nw, S), np.uint8)
    _load(w, a, A)
    _load(w, b, B)
    assert np.array_equal(_value(c.evaluate_plain(gates, w), s), A + B)


@pytest.mark.parametrize("nbits", [1, 2, 4])
def test_literal_baseline_adder_five_gates_per_bit(nbits):
    """BASELINE.md counts configs[2] at a uniform 5 bootstrapped gates per bit (40 per 8-bit pair, 163 840 for 4096
    pairs): the carry_in_zero variant is that netlist -- a full adder at bit 0 whose carry-in is bootsCONSTANT(0)"""
    gates, nw, a, b, s = c.ripple_carry_adder(nbits, carry_in_zero=True)
    assert circuit_bootstraps(gates) == 5 * nbits
    assert gates[0].op == OPS["CONST0"]
    A, B, S = _words(nbits)
    w = np.zeros((nw, S), np.uint8)
    _load(w, a, A)
    _load(w, b, B)
    assert np.array_equal(_value(c.evaluate_plain(gates, w), s), A + B)
    assert circuit_bootstraps(c.ripple_carry_adder(8, carry_in_zero=True)[0]) * 4096 == 163840
    assert circuit_bootstraps(c.ripple_carry_adder(8)[0]) == 37


@pytest.mark.parametrize("nbits", [1, 3, 5])
def test_less_than_and_min_max_all_inputs(nbits):
    A, B, S = _words(nbits)
    gates, nw, a, b, lt = c.less_than(nbits)
    assert circuit_bootstraps(gates) == 1 + 3 * (nbits - 1)
    w = np.zeros((nw, S), np.uint8)
    _load(w, a, A)
    _load(w, b, B)
    assert np.array_equal(c.evaluate_plain(gates, w)[lt], (A < B).astype(np.uint8))
    gates, nw, a, b, mn, mx = c.min_max(nbits)
    w = np.zeros((nw, S), np.uint8)
    _load(w, a, A)
    _load(w, b, B)
    r = c.evaluate_plain(gates, w)
    assert np.array_equal(_value(r, mn), np.minimum(A, B))
    assert np.array_equal(_value(r, mx), np.maximum(A, B))


def test_string_equal_plain():
    gates, nw, x, y, out = c.string_equal(2)
    rng = np.random.default_rng(5)
    S = 64
    X = rng.integers(0, 1 << 16, S)
    Y = np.where(rng.integers(0, 2, S) == 1, X, rng.integers(0, 1 << 16, S))
    w = np.zeros((nw, S), np.uint8)
    _load(w, x, X)
    _load(w, y, Y)
    assert np.array_equal(c.evaluate_plain(gates, w)[out], (X == Y).astype(np.uint8))
    assert circuit_bootstraps(gates) == 16 + 15


def test_fold_nots_every_op_and_polarity():
    two = ["NAND", "AND", "OR", "NOR", "XOR", "XNOR", "ANDNY", "ANDYN", "ORNY", "ORYN"]
    bits = np.array([[0, 0, 1, 1], [0, 1, 0, 1]], np.uint8)
    for name in two:
        for n0 in (0, 1):
            for n1 in (0, 1):
                gates, nxt = [], 2
                i0, i1 = 0, 1
                if n0:
                    gates.append(Gate(OPS["NOT"], 0, -1, -1, nxt)); i0 = nxt; nxt += 1
                if n1:
                    gates.append(Gate(OPS["NOT"], 1, -1, -1, nxt)); i1 = nxt; nxt += 1
                gates.append(Gate(OPS[name], i0, i1, -1, nxt)); out = nxt; nxt += 1
                opt = c.fold_nots(gates, [out])
                assert len(opt) == 1 and opt[0].in0 == 0 and opt[0].in1 == 1, (name, n0, n1)
                w = np.zeros((nxt, 4), np.uint8)
                w[:2] = bits
                assert np.array_equal(c.evaluate_plain(gates, w)[out], c.evaluate_plain(opt, w)[out]), (name, n0, n1)


def test_double_not_and_mux_selector():
    g = [Gate(OPS["NOT"], 0, -1, -1, 3), Gate(OPS["NOT"], 3, -1, -1, 4), Gate(OPS["MUX"], 3, 1, 2, 5)]
    opt = c.fold_nots(g, [4, 5])
    names = [c._NAMES[x.op] for x in opt]
    assert names == ["COPY", "MUX"] and (opt[1].in0, opt[1].in1, opt[1].in2) == (0, 2, 1)
    w = np.zeros((6, 8), np.uint8)
    for k in range(8):
        w[0, k], w[1, k], w[2, k] = k & 1, (k >> 1) & 1, (k >> 2) & 1
    a, b = c.evaluate_plain(g, w), c.evaluate_plain(opt, w)
    assert np.array_equal(a[4], b[4]) and np.array_equal(a[5], b[5])


def test_fuse_mux_word_select():
    # out_i = (s & x_i) | (~s & y_i) written the long way: 1 NOT + 3 gates per bit -> 1 MUX per bit
    nb = 4
    s, x, y = 0, list(range(1, 1 + nb)), list(range(1 + nb, 1 + 2 * nb))
    nxt = 1 + 2 * nb
    ns = nxt; nxt += 1
    gates = [Gate(OPS["NOT"], s, -1, -1, ns)]
    outs = []
    for i in range(nb):
        t0, t1, o = nxt, nxt + 1, nxt + 2
        nxt += 3
        gates += [Gate(OPS["AND"], s, x[i], -1, t0), Gate(OPS["AND"], ns, y[i], -1, t1), Gate(OPS["OR"], t0, t1, -1, o)]
        outs.append(o)
    opt = c.optimize(gates, outs)
    assert [c._NAMES[g.op] for g in opt] == ["MUX"] * nb
    assert circuit_bootstraps(gates) == 3 * nb and circuit_bootstraps(opt) == 2 * nb
    rng = np.random.default_rng(2)
    w = np.zeros((nxt, 200), np.uint8)
    w[: 1 + 2 * nb] = rng.integers(0, 2, (1 + 2 * nb, 200))
    a, b = c.evaluate_plain(gates, w), c.evaluate_plain(opt, w)
    for o in outs:
        assert np.array_equal(a[o], b[o])


def test_fuse_mux_keeps_shared_inner_wires():
    # the inner AND is read by a second gate, so it must stay
    gates = [Gate(OPS["AND"], 0, 1, -1, 3), Gate(OPS["ANDNY"], 0, 2, -1, 4), Gate(OPS["OR"], 3, 4, -1, 5),
             Gate(OPS["XOR"], 3, 2, -1, 6)]
    opt = c.optimize(gates, [5, 6])
    assert [c._NAMES[g.op] for g in opt] == ["AND", "ANDNY", "OR", "XOR"]


def test_random_netlists_rewrite_is_equivalent():
    rng = np.random.default_rng(11)
    names = list(c._SEM2) + ["NOT", "NOT", "NOT", "MUX", "COPY"]
    for trial in range(60):
        n_in, n_g = 4, int(rng.integers(5, 40))
        gates, avail = [], list(range(n_in))
        for k in range(n_g):
            name = names[int(rng.integers(0, len(names)))]
            pick = lambda: int(avail[int(rng.integers(0, len(avail)))])
            out = n_in + k
            if name in ("NOT", "COPY"):
                gates.append(Gate(OPS[name], pick(), -1, -1, out))
            elif name == "MUX":
                gates.append(Gate(OPS[name], pick(), pick(), pick(), out))
            else:
                gates.append(Gate(OPS[name], pick(), pick(), -1, out))
            avail.append(out)
        outs = [int(v) for v in rng.choice(avail[n_in:], size=min(3, n_g), replace=False)]
        opt = c.optimize(gates, outs)
        assert circuit_bootstraps(opt) <= circuit_bootstraps(gates)
        w = np.zeros((n_in + n_g, 16), np.uint8)
        for k in range(16):
            for i in range(n_in):
                w[i, k] = (k >> i) & 1
        a, b = c.evaluate_plain(gates, w), c.evaluate_plain(opt, w)
        for o in outs:
            assert np.array_equal(a[o], b[o]), (trial, o)


def test_rewrites_refuse_non_ssa():
    g = [Gate(OPS["AND"], 0, 1, -1, 2), Gate(OPS["OR"], 0, 1, -1, 2)]
    with pytest.raises(ValueError):
        c.fold_nots(g, [2])


def test_native_optimizer_matches_python(built_lib):
    """eoc_netlist_optimize (C ABI, host.cpp) rewrites exactly like circuits.optimize"""
    import eoc_tfhe_amd as eoc
    rng = np.random.default_rng(23)
    names = list(c._SEM2) + ["NOT", "NOT", "NOT", "MUX", "COPY", "CONST0", "CONST1"]
    as_t = lambda gs: [(g.op, g.in0, g.in1, g.in2, g.out) for g in gs]
    for trial in range(80):
        n_in, n_g = 4, int(rng.integers(1, 50))
        gates, avail = [], list(range(n_in))
        for k in range(n_g):
            name = names[int(rng.integers(0, len(names)))]
            pick = lambda: int(avail[int(rng.integers(0, len(avail)))])
            out = n_in + k
            if name in ("CONST0", "CONST1"):
                gates.append(Gate(OPS[name], -1, -1, -1, out))
            elif name in ("NOT", "COPY"):
                gates.append(Gate(OPS[name], pick(), -1, -1, out))
            elif name == "MUX":
                gates.append(Gate(OPS[name], pick(), pick(), pick(), out))
            else:
                gates.append(Gate(OPS[name], pick(), pick(), -1, out))
            avail.append(out)
        outs = [int(v) for v in rng.choice(avail[n_in:], size=min(3, n_g), replace=False)]
        w = np.zeros((n_in + n_g, 16), np.uint8)
        for k in range(16):
            for i in range(n_in):
                w[i, k] = (k >> i) & 1
        ref, opt = c.evaluate_plain(gates, w), c.evaluate_plain(c.optimize(gates, outs), w)
        assert all(np.array_equal(ref[o], opt[o]) for o in outs), trial
        assert as_t(eoc.netlist_optimize(gates, outs)) == as_t(c.optimize(gates, outs)), trial
    # the word-select example: 1 NOT + 3 gates per bit -> 1 MUX per bit
    gates = [Gate(OPS["NOT"], 0, -1, -1, 9)]
    outs = []
    for i in range(4):
        gates += [Gate(OPS["AND"], 0, 1 + i, -1, 10 + 3 * i), Gate(OPS["AND"], 9, 5 + i, -1, 11 + 3 * i),
                  Gate(OPS["OR"], 10 + 3 * i, 11 + 3 * i, -1, 12 + 3 * i)]
        outs.append(12 + 3 * i)
    assert [g.op for g in eoc.netlist_optimize(gates, outs)] == [OPS["MUX"]] * 4
    # not single-assignment / malformed -> error
    with pytest.raises(eoc.EocError):
        eoc.netlist_optimize([Gate(OPS["AND"], 0, 1, -1, 2), Gate(OPS["OR"], 0, 1, -1, 2)], [2])
    with pytest.raises(eoc.EocError):
        eoc.netlist_optimize([Gate(OPS["AND"], 3, 1, -1, 2), Gate(OPS["OR"], 0, 1, -1, 3)], [2])
    assert eoc.netlist_optimize([], []) == []


@pytest.mark.parametrize("nbits", [1, 2, 4, 5])
def test_subtractor_and_multiplier_all_inputs(nbits):
    """round 3: a - b (with the final borrow = a < b) and a * b (schoolbook: nbits^2 partial products + nbits - 1
    ripple rows), every input pair; the host-side optimiser (NOT folding, MUX fusion, dead gates) keeps their meaning"""
    A, B, S = _words(nbits)
    gates, nw, a, b, d, br = c.subtractor(nbits)
    assert circuit_bootstraps(gates) == 2 + 4 * (nbits - 1)
    w = np.zeros((nw, S), np.uint8)
    _load(w, a, A)
    _load(w, b, B)
    r = c.evaluate_plain(gates, w)
    assert np.array_equal(_value(r, d), (A - B) % (1 << nbits)) and np.array_equal(r[br], (A < B).astype(np.uint8))
    gates, nw, a, b, p = c.multiplier(nbits)
    w = np.zeros((nw, S), np.uint8)
    _load(w, a, A)
    _load(w, b, B)
    assert np.array_equal(_value(c.evaluate_plain(gates, w), p), A * B)
    opt = c.optimize(gates, p)
    assert circuit_bootstraps(opt) <= circuit_bootstraps(gates)
    assert np.array_equal(_value(c.evaluate_plain(opt, w), p), A * B)
