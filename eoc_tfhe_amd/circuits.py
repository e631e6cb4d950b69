"""Netlist builders for the circuits BASELINE.json names (they do not exist in ao-tfhe/tfhe.lua --
SURVEY.md 0.1 -- so this repo defines them, in the reference's style; INTEGRATION.md has the Lua text).

A netlist is a list of eoc_tfhe_amd.Gate over integer wire ids; wires [w][instance][n+1] hold LWE
samples; eoc_circuit_run(_device) levelises and batches every level over all instances.
"""
from . import OPS, Gate


def ripple_carry_adder(nbits=8, carry_in_zero=False):
    """a[0..nbits) + b[0..nbits) -> s[0..nbits], LSB first.  5 gates per bit (2 XOR, 2 AND, 1 OR).
    Default: bit 0 is a half adder (XOR + AND), 5*nbits - 3 bootstraps (37 for 8 bits).
    carry_in_zero=True: bit 0 is a full adder too, its carry-in the noiseless constant 0 (bootsCONSTANT, free) --
    the uniform 5 gates per bit BASELINE.md counts for configs[2]: 5*nbits = 40 bootstraps per 8-bit pair, 163 840 for
    4096 pairs.  Same sums; the literal configuration costs 3 bootstraps more per pair.
    Wires: a = 0..nbits-1, b = nbits..2nbits-1, sum = 2nbits..3nbits (nbits+1 wires), then temporaries.
    Returns (gates, n_wires, a_wires, b_wires, sum_wires)."""
    a = list(range(nbits))
    b = list(range(nbits, 2 * nbits))
    s = list(range(2 * nbits, 3 * nbits + 1))
    nxt = 3 * nbits + 1
    gates = []
    carry = None
    if carry_in_zero:
        carry = nxt; nxt += 1
        gates.append(Gate(OPS["CONST0"], -1, -1, -1, carry))
    for i in range(nbits):
        if carry is None:
            gates.append(Gate(OPS["XOR"], a[i], b[i], -1, s[i]))
            carry = nxt; nxt += 1
            gates.append(Gate(OPS["AND"], a[i], b[i], -1, carry))
        else:
            p, g, pc = nxt, nxt + 1, nxt + 2
            nxt += 3
            gates.append(Gate(OPS["XOR"], a[i], b[i], -1, p))
            gates.append(Gate(OPS["AND"], a[i], b[i], -1, g))
            gates.append(Gate(OPS["XOR"], p, carry, -1, s[i]))
            gates.append(Gate(OPS["AND"], p, carry, -1, pc))
            newc = s[nbits] if i == nbits - 1 else nxt
            if i != nbits - 1:
                nxt += 1
            gates.append(Gate(OPS["OR"], g, pc, -1, newc))
            carry = newc
    if nbits == 1 and not carry_in_zero:
        gates.append(Gate(OPS["COPY"], carry, -1, -1, s[1]))
    return gates, nxt, a, b, s


def string_equal(nbytes=32):
    """x[0..8nbytes) == y[0..8nbytes): XOR per bit, OR tree, final NOT (free).
    8nbytes XOR + (8nbytes - 1) OR bootstraps (= 511 for 32 bytes, BASELINE config 5).
    Returns (gates, n_wires, x_wires, y_wires, out_wire)."""
    nb = 8 * nbytes
    x = list(range(nb))
    y = list(range(nb, 2 * nb))
    nxt = 2 * nb
    gates = []
    level = []
    for i in range(nb):
        gates.append(Gate(OPS["XOR"], x[i], y[i], -1, nxt))
        level.append(nxt)
        nxt += 1
    while len(level) > 1:
        new = []
        for i in range(0, len(level) - 1, 2):
            gates.append(Gate(OPS["OR"], level[i], level[i + 1], -1, nxt))
            new.append(nxt)
            nxt += 1
        if len(level) % 2:
            new.append(level[-1])
        level = new
    out = nxt
    gates.append(Gate(OPS["NOT"], level[0], -1, -1, out))
    return gates, nxt + 1, x, y, out


# ---- more word-level circuits (comparison / min / max) ------------------------------

def less_than(nbits=8):
    """unsigned a < b, LSB first: lt_0 = (not a_0) and b_0; lt_i = MUX(a_i XNOR b_i, lt_{i-1}, b_i).
    1 + 3 (nbits - 1) bootstraps (XNOR = 1, MUX = 2).  Returns (gates, n_wires, a_wires, b_wires, out_wire)."""
    a = list(range(nbits))
    b = list(range(nbits, 2 * nbits))
    nxt = 2 * nbits
    gates = [Gate(OPS["ANDNY"], a[0], b[0], -1, nxt)]
    lt = nxt
    nxt += 1
    for i in range(1, nbits):
        eq, new = nxt, nxt + 1
        nxt += 2
        gates.append(Gate(OPS["XNOR"], a[i], b[i], -1, eq))
        gates.append(Gate(OPS["MUX"], eq, lt, b[i], new))
        lt = new
    return gates, nxt, a, b, lt


def min_max(nbits=8):
    """(min(a, b), max(a, b)) of two unsigned words: one comparator, then a MUX per output bit.
    Returns (gates, n_wires, a_wires, b_wires, min_wires, max_wires)."""
    gates, nxt, a, b, lt = less_than(nbits)
    mn = list(range(nxt, nxt + nbits))
    mx = list(range(nxt + nbits, nxt + 2 * nbits))
    for i in range(nbits):
        gates.append(Gate(OPS["MUX"], lt, a[i], b[i], mn[i]))
        gates.append(Gate(OPS["MUX"], lt, b[i], a[i], mx[i]))
    return gates, nxt + 2 * nbits, a, b, mn, mx


def subtractor(nbits=8):
    """a - b mod 2^nbits with the final borrow (= a < b), LSB first: p_i = a_i XOR b_i, d_i = p_i XOR br_i,
    br_{i+1} = MUX(p_i, b_i, br_i) (a_i != b_i: a borrow arises iff b_i = 1; equal bits pass the borrow on).
    Bit 0 has no incoming borrow: d_0 = p_0, br_1 = ANDNY(a_0, b_0).  2 + 4 (nbits - 1) bootstraps.
    Returns (gates, n_wires, a_wires, b_wires, diff_wires, borrow_wire)."""
    a = list(range(nbits))
    b = list(range(nbits, 2 * nbits))
    d = list(range(2 * nbits, 3 * nbits))
    nxt = 3 * nbits
    gates = [Gate(OPS["XOR"], a[0], b[0], -1, d[0]), Gate(OPS["ANDNY"], a[0], b[0], -1, nxt)]
    br = nxt
    nxt += 1
    for i in range(1, nbits):
        p, new = nxt, nxt + 1
        nxt += 2
        gates.append(Gate(OPS["XOR"], a[i], b[i], -1, p))
        gates.append(Gate(OPS["XOR"], p, br, -1, d[i]))
        gates.append(Gate(OPS["MUX"], p, b[i], br, new))
        br = new
    return gates, nxt, a, b, d, br


def multiplier(nbits=4):
    """a * b -> 2 nbits product bits, LSB first: nbits^2 AND partial products, then nbits - 1 shifted ripple-carry rows
    (row r adds the partial products a_j b_r at weight r + j into the running sum; the low bit of every running sum is
    final).  All partial products are one level; the rows chain through their carries.
    Returns (gates, n_wires, a_wires, b_wires, product_wires)."""
    a = list(range(nbits))
    b = list(range(nbits, 2 * nbits))
    nxt = 2 * nbits
    gates = []
    pp = [[0] * nbits for _ in range(nbits)]            # pp[r][j] = a_j AND b_r
    for r in range(nbits):
        for j in range(nbits):
            gates.append(Gate(OPS["AND"], a[j], b[r], -1, nxt))
            pp[r][j] = nxt
            nxt += 1
    prod = [pp[0][0]]
    acc = pp[0][1:]                                     # running sum above the bits already final (nbits - 1 wires)
    top = None                                          # its carry-out bit, None while it is known to be zero
    for r in range(1, nbits):
        row = pp[r]                                     # nbits wires, aligned with acc[0]
        new_acc, carry = [], None
        for j in range(nbits):
            x = acc[j] if j < len(acc) else top         # bit j of the running sum (None = constant 0)
            y = row[j]
            if x is None and carry is None:
                sbit, cout = y, None
            elif x is None or carry is None:            # half adder
                z = carry if x is None else x
                sbit, cout = nxt, nxt + 1
                nxt += 2
                gates.append(Gate(OPS["XOR"], z, y, -1, sbit))
                gates.append(Gate(OPS["AND"], z, y, -1, cout))
            else:                                       # full adder: 2 XOR + 2 AND + 1 OR
                p, g, pc = nxt, nxt + 1, nxt + 2
                sbit, cout = nxt + 3, nxt + 4
                nxt += 5
                gates.append(Gate(OPS["XOR"], x, y, -1, p))
                gates.append(Gate(OPS["AND"], x, y, -1, g))
                gates.append(Gate(OPS["XOR"], p, carry, -1, sbit))
                gates.append(Gate(OPS["AND"], p, carry, -1, pc))
                gates.append(Gate(OPS["OR"], g, pc, -1, cout))
            new_acc.append(sbit)
            carry = cout
        prod.append(new_acc[0])
        acc, top = new_acc[1:], carry
    prod += acc
    if top is not None:
        prod.append(top)
    else:
        z = nxt
        nxt += 1
        gates.append(Gate(OPS["CONST0"], -1, -1, -1, z))
        prod.append(z)
    assert len(prod) == 2 * nbits
    return gates, nxt, a, b, prod


# ---- plaintext semantics and netlist rewriting -----------------------------------------------------

_NAMES = {v: k for k, v in OPS.items()}
_SEM2 = {
    "NAND": lambda a, b: 1 - (a & b), "AND": lambda a, b: a & b, "OR": lambda a, b: a | b,
    "NOR": lambda a, b: 1 - (a | b), "XOR": lambda a, b: a ^ b, "XNOR": lambda a, b: 1 - (a ^ b),
    "ANDNY": lambda a, b: (1 - a) & b, "ANDYN": lambda a, b: a & (1 - b),
    "ORNY": lambda a, b: (1 - a) | b, "ORYN": lambda a, b: a | (1 - b),
}


def evaluate_plain(gates, wires):
    """Run a netlist on plaintext bits: wires is an integer array [n_wires][instances] (modified copy is
    returned).  The truth tables are those of the boots* gates."""
    import numpy as np
    w = np.array(wires, dtype=np.uint8, copy=True)
    for g in gates:
        name = _NAMES[g.op]
        if name == "NOT":
            w[g.out] = 1 - w[g.in0]
        elif name == "COPY":
            w[g.out] = w[g.in0]
        elif name in ("CONST0", "CONST1"):
            w[g.out] = 1 if name == "CONST1" else 0
        elif name == "MUX":
            w[g.out] = np.where(w[g.in0] == 1, w[g.in1], w[g.in2])
        else:
            w[g.out] = _SEM2[name](w[g.in0], w[g.in1])
    return w


def _table(name, n0=0, n1=0):
    f = _SEM2[name]
    return tuple(f(a ^ n0, b ^ n1) for a in (0, 1) for b in (0, 1))


_BY_TABLE = {_table(k): k for k in _SEM2}


def _check_ssa(gates):
    written = set()
    for g in gates:
        if g.out in written:
            raise ValueError("netlist rewriting needs single-assignment wires (wire %d is written twice)" % g.out)
        for i in (g.in0, g.in1, g.in2):
            if i >= 0 and i == g.out:
                raise ValueError("gate reads its own output wire %d" % g.out)
        written.add(g.out)
    for k, g in enumerate(gates):
        for i in (g.in0, g.in1, g.in2):
            if i >= 0 and i in written and not any(h.out == i for h in gates[:k]):
                raise ValueError("wire %d is read before it is written" % i)


def _uses(gates):
    u = {}
    for g in gates:
        for i in (g.in0, g.in1, g.in2):
            if i >= 0:
                u[i] = u.get(i, 0) + 1
    return u


def _drop_dead(gates, keep):
    keep = set(keep)
    while True:
        u = _uses(gates)
        live = [g for g in gates if g.out in keep or u.get(g.out, 0) > 0]
        if len(live) == len(gates):
            return live
        gates = live


def fold_nots(gates, outputs):
    """NOT is free, but a NOT in front of a bootstrapped gate is unnecessary altogether: the ten two-input
    boots* gates are closed under input negation (AND with a negated first input IS bootsANDNY, ...), a negated
    MUX selector swaps the branches, NOT(NOT x) is a COPY.  `outputs` are the wires the caller reads; NOT gates
    nobody reads afterwards are dropped.  Single-assignment netlists only."""
    _check_ssa(gates)
    src = {g.out: g for g in gates}

    def strip(wire):
        neg = 0
        while wire in src and _NAMES[src[wire].op] == "NOT":
            wire, neg = src[wire].in0, neg ^ 1
        return wire, neg

    out = []
    for g in gates:
        name = _NAMES[g.op]
        if name in _SEM2:
            (i0, n0), (i1, n1) = strip(g.in0), strip(g.in1)
            out.append(Gate(OPS[_BY_TABLE[_table(name, n0, n1)]], i0, i1, -1, g.out))
        elif name == "MUX":
            (s, ns) = strip(g.in0)
            b, c = (g.in2, g.in1) if ns else (g.in1, g.in2)
            out.append(Gate(g.op, s, b, c, g.out))
        elif name == "NOT":
            i0, n0 = strip(g.in0)
            out.append(Gate(OPS["COPY"] if n0 else OPS["NOT"], i0, -1, -1, g.out))
        else:
            out.append(Gate(g.op, g.in0, g.in1, g.in2, g.out))
    return _drop_dead(out, outputs)


def fuse_mux(gates, outputs):
    """OR(AND(s, b), ANDNY(s, c)) with single-use inner wires is bootsMUX(s, b, c): 2 blind rotations and
    one key switch instead of 3 + 3.  Run fold_nots first so that AND(NOT s, c) has become
    ANDNY(s, c).  Single-assignment netlists only."""
    _check_ssa(gates)
    src = {g.out: g for g in gates}
    uses = _uses(gates)
    keep = set(outputs)

    def as_sel(g):
        """(selector, data, polarity) for a gate computing sel&data (polarity 1) or (not sel)&data (0)"""
        name = _NAMES[g.op]
        if name == "AND":
            return [(g.in0, g.in1, 1), (g.in1, g.in0, 1)]
        if name == "ANDNY":
            return [(g.in0, g.in1, 0)]
        if name == "ANDYN":
            return [(g.in1, g.in0, 0)]
        return []

    out = []
    for g in gates:
        done = False
        if _NAMES[g.op] == "OR" and g.in0 in src and g.in1 in src:
            x, y = src[g.in0], src[g.in1]
            inner_ok = all(uses.get(t.out, 0) == 1 and t.out not in keep for t in (x, y))
            if inner_ok:
                for (s0, d0, p0) in as_sel(x):
                    for (s1, d1, p1) in as_sel(y):
                        if not done and s0 == s1 and p0 != p1:
                            b, c = (d0, d1) if p0 else (d1, d0)
                            out.append(Gate(OPS["MUX"], s0, b, c, g.out))
                            done = True
        if not done:
            out.append(Gate(g.op, g.in0, g.in1, g.in2, g.out))
    return _drop_dead(out, outputs)


def optimize(gates, outputs):
    """fold_nots, then fuse_mux; returns the rewritten netlist (same wire numbering, fewer gates)."""
    return fuse_mux(fold_nots(gates, outputs), outputs)
