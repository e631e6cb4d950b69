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


# ---- forms chosen by instance count: fewest bootstraps for wide batches, fewest levels for small ones -------------

def mux_carry_adder(nbits=8):
    """a + b -> nbits + 1 sum bits, LSB first, with the carry as ONE gate per bit: p_i = a_i XOR b_i,
    c_{i+1} = MUX(p_i, c_i, a_i) (inputs differ: the carry-in passes; inputs agree: either of them is the carry),
    s_i = p_i XOR c_i; bit 0 is a half adder.  2 + 4 (nbits - 1) bootstraps on nbits dependent levels (30 / 8 for 8 bits
    against the textbook form's 37 / 16): what fuse_carry makes of ripple_carry_adder, written directly.
    Same wire layout as ripple_carry_adder.  Returns (gates, n_wires, a_wires, b_wires, sum_wires)."""
    a = list(range(nbits))
    b = list(range(nbits, 2 * nbits))
    s = list(range(2 * nbits, 3 * nbits + 1))
    nxt = 3 * nbits + 1
    gates = [Gate(OPS["XOR"], a[0], b[0], -1, s[0])]
    carry = s[1] if nbits == 1 else nxt
    if nbits > 1:
        nxt += 1
    gates.append(Gate(OPS["AND"], a[0], b[0], -1, carry))
    for i in range(1, nbits):
        p = nxt
        nxt += 1
        gates.append(Gate(OPS["XOR"], a[i], b[i], -1, p))
        gates.append(Gate(OPS["XOR"], p, carry, -1, s[i]))
        newc = s[nbits] if i == nbits - 1 else nxt
        if i != nbits - 1:
            nxt += 1
        gates.append(Gate(OPS["MUX"], p, carry, a[i], newc))
        carry = newc
    return gates, nxt, a, b, s


def maj_adder(nbits=8):
    """a + b with the EXTENSION gates (include/eoc_tfhe_gpu.h): a full adder is s_i = XOR3(a_i, b_i, c_i) and
    c_{i+1} = MAJ(a_i, b_i, c_i) -- ONE bootstrap each, both on the level of c_i; bit 0 is a half adder (XOR, AND).
    2 nbits bootstraps on nbits levels: 16 / 8 at 8 bits against the MUX-carry form's 30 / 8 and the textbook form's 37 / 15;
    it is what optimize() makes of either.  Same wire layout as ripple_carry_adder.
    Returns (gates, n_wires, a_wires, b_wires, sum_wires)."""
    a = list(range(nbits))
    b = list(range(nbits, 2 * nbits))
    s = list(range(2 * nbits, 3 * nbits + 1))
    nxt = 3 * nbits + 1
    gates = [Gate(OPS["XOR"], a[0], b[0], -1, s[0])]
    carry = s[1] if nbits == 1 else nxt
    if nbits > 1:
        nxt += 1
    gates.append(Gate(OPS["AND"], a[0], b[0], -1, carry))
    for i in range(1, nbits):
        gates.append(Gate(OPS["XOR3"], a[i], b[i], carry, s[i]))
        newc = s[nbits] if i == nbits - 1 else nxt
        if i != nbits - 1:
            nxt += 1
        gates.append(Gate(OPS["MAJ"], a[i], b[i], carry, newc))
        carry = newc
    return gates, nxt, a, b, s


def maj_subtractor(nbits=8):
    """a - b mod 2^nbits and the final borrow with the extension gates: d_i = XOR3(a_i, b_i, br_i),
    br_{i+1} = MAJ(NOT a_i, b_i, br_i) (NOT is free); bit 0: d_0 = XOR, br_1 = ANDNY.  2 nbits bootstraps on nbits levels
    (16 / 8 against subtractor's 30 / 8).  Same wire layout as subtractor.
    Returns (gates, n_wires, a_wires, b_wires, diff_wires, borrow_wire)."""
    a = list(range(nbits))
    b = list(range(nbits, 2 * nbits))
    d = list(range(2 * nbits, 3 * nbits))
    nxt = 3 * nbits
    gates = [Gate(OPS["XOR"], a[0], b[0], -1, d[0]), Gate(OPS["ANDNY"], a[0], b[0], -1, nxt)]
    br = nxt
    nxt += 1
    for i in range(1, nbits):
        na, new = nxt, nxt + 1
        nxt += 2
        gates.append(Gate(OPS["NOT"], a[i], -1, -1, na))
        gates.append(Gate(OPS["XOR3"], a[i], b[i], br, d[i]))
        gates.append(Gate(OPS["MAJ"], na, b[i], br, new))
        br = new
    return gates, nxt, a, b, d, br


def maj_less_than(nbits=8):
    """unsigned a < b = the final borrow of a - b: lt_0 = ANDNY(a_0, b_0), lt_i = MAJ(NOT a_i, b_i, lt_{i-1}): ONE bootstrap
    per bit (8 on 8 levels against less_than's 22 on 8).  Returns (gates, n_wires, a_wires, b_wires, out_wire)."""
    a = list(range(nbits))
    b = list(range(nbits, 2 * nbits))
    nxt = 2 * nbits
    gates = [Gate(OPS["ANDNY"], a[0], b[0], -1, nxt)]
    lt = nxt
    nxt += 1
    for i in range(1, nbits):
        na, new = nxt, nxt + 1
        nxt += 2
        gates.append(Gate(OPS["NOT"], a[i], -1, -1, na))
        gates.append(Gate(OPS["MAJ"], na, b[i], lt, new))
        lt = new
    return gates, nxt, a, b, lt


def _prefix_cells(emit, a, b, sub, outs=None, top=None):
    """Sklansky parallel-prefix network over (generate, propagate) pairs on the wire lists a, b (LSB first), shared by
    prefix_adder (sub = False: G = a AND b, P = a XOR b, out_i = P_i XOR carry_i), prefix_subtractor (sub = True:
    G = (NOT a) AND b -- a borrow arises --, P = a XNOR b -- a borrow passes --, out_i = P_i XNOR borrow_i) and the final
    addition of wallace_multiplier.  emit(op, i0, i1, i2=-1, out=None) appends a gate and returns its output wire; outs /
    top name the output wires (fresh ones if None).  Returns (out wires, carry / borrow out)."""
    nbits = len(a)
    outs = list(outs) if outs is not None else [None] * nbits
    p_op, g_op, o_op = ("XNOR", "ANDNY", "XNOR") if sub else ("XOR", "AND", "XOR")
    outs[0] = emit("XOR", a[0], b[0], out=outs[0])          # bit 0 has no carry / borrow in
    if nbits == 1:
        return outs, emit(g_op, a[0], b[0], out=top)
    nlev = (nbits - 1).bit_length()
    # position 0 is never an upper operand: its P is not needed (the adder's would be the sum bit itself)
    P = [None] + [emit(p_op, a[i], b[i]) for i in range(1, nbits)]
    p_bit = list(P)
    # a generate wire only where the position serves as a LOWER operand while still a single bit (even positions)
    G = [emit(g_op, a[i], b[i]) if i % 2 == 0 and i + 1 < nbits or i == 0 else None for i in range(nbits)]
    single = [True] * nbits
    for k in range(nlev):
        newG, newP, newsingle = list(G), list(P), list(single)
        for i in range(nbits):
            if not (i >> k) & 1:
                continue
            j = ((i >> k) << k) - 1                     # last position of the block below
            final = i < (1 << (k + 1))                  # the result covers bits 0 .. i
            last = final and i == nbits - 1
            # a single bit as the upper operand needs no generate wire: where its inputs differ (P = 0) the carry is a_i,
            # the borrow b_i
            # (the borrow as NOT a_i, a free gate: the optimizer then turns the cell into MAJ(NOT a_i, b_i, G_lo))
            g_hi = (emit("NOT", a[i], -1) if sub else a[i]) if single[i] else G[i]
            newG[i] = emit("MUX", P[i], G[j], g_hi, out=top if last else None)
            # P of the combined group is read only by a later cell that uses position i (or a higher position of its
            # block) as the upper operand: never once the group starts at bit 0
            newP[i] = None if final else emit("AND", P[i], P[j])
            newsingle[i] = False
        G, P, single = newG, newP, newsingle
    for i in range(1, nbits):
        outs[i] = emit(o_op, p_bit[i], G[i - 1], out=outs[i])
    return outs, G[nbits - 1]


def _emitter(gates, first_free):
    state = {"nxt": first_free}

    def emit(op, i0, i1, i2=-1, out=None):
        if out is None:
            out = state["nxt"]
            state["nxt"] += 1
        gates.append(Gate(OPS[op], i0, i1, i2, out))
        return out
    return emit, state


def _prefix_network(nbits, sub):
    """wires: a = 0..n-1, b = n..2n-1, outputs 2n..3n-1, the carry / borrow out at 3n, then temporaries"""
    a = list(range(nbits))
    b = list(range(nbits, 2 * nbits))
    gates = []
    emit, state = _emitter(gates, 3 * nbits + 1)
    o, top = _prefix_cells(emit, a, b, sub, outs=list(range(2 * nbits, 3 * nbits)), top=3 * nbits)
    return gates, state["nxt"], a, b, o, top


def prefix_adder(nbits=8):
    """a + b in logarithmic depth: a Sklansky parallel-prefix network over (generate, propagate) pairs.  A group's G and P
    exclude each other, so the prefix cell (G, P) = (G_hi OR (P_hi AND G_lo), P_hi AND P_lo) is ONE MUX and one AND on the
    same level: G = MUX(P_hi, G_lo, G_hi).  A single bit as the upper operand needs no generate wire of its own
    (MUX(p_i, G_lo, a_i)), a prefix that already starts at bit 0 needs no P.  Levels: 1 (p_i, g_i) + ceil(log2 nbits)
    + 1 (sums) -- 5 for 8 bits, 6 for 16 -- at 48 bootstraps per 8-bit pair (mux_carry_adder: 30 on 8 levels).
    Same wire layout as ripple_carry_adder.  Returns (gates, n_wires, a_wires, b_wires, sum_wires)."""
    gates, n_wires, a, b, o, top = _prefix_network(nbits, False)
    return gates, n_wires, a, b, o + [top]


def prefix_subtractor(nbits=8):
    """a - b mod 2^nbits and the final borrow (= a < b) in logarithmic depth: the prefix network of prefix_adder over
    (a borrow arises, a borrow passes) = ((NOT a) AND b, a XNOR b); 48 bootstraps on 5 levels at 8 bits against subtractor's
    30 on 8.  Same wire layout as subtractor (difference 2n..3n-1).  Returns (gates, n_wires, a, b, diff_wires, borrow)."""
    return _prefix_network(nbits, True)


def wallace_multiplier(nbits=4, extension_gates=True):
    """a * b -> 2 nbits product bits in logarithmic depth: nbits^2 AND partial products in columns by weight, column
    compression by full adders until no column holds more than two wires, then ONE parallel-prefix addition of the two
    remaining rows.  A full adder is XOR3(x, y, z) + MAJ(x, y, z) -- 2 bootstraps on ONE level (extension gates, default)
    -- or, inside libtfhe's gate family, (x XOR y) XOR z + MUX(x XOR y, z, x): 4 bootstraps on 2 levels, the latest
    arriving wire as z.  8 bits: 11 levels (16 without the extension gates) against the row-by-row multiplier's 40.
    Returns (gates, n_wires, a_wires, b_wires, product_wires)."""
    a = list(range(nbits))
    b = list(range(nbits, 2 * nbits))
    gates = []
    emit, state = _emitter(gates, 2 * nbits)
    if nbits == 1:
        p0 = emit("AND", a[0], b[0])
        return gates + [Gate(OPS["CONST0"], -1, -1, -1, state["nxt"])], state["nxt"] + 1, a, b, [p0, state["nxt"]]
    cols = [[] for _ in range(2 * nbits)]               # column c: (level, wire) of weight 2^c
    for r in range(nbits):
        for j in range(nbits):
            cols[r + j].append((1, emit("AND", a[j], b[r])))

    def full_adder(x, y, z):                            # -> (level, sum wire), (level, carry wire)
        (lx, wx), (ly, wy), (lz, wz) = x, y, z
        if extension_gates:
            lv = max(lx, ly, lz) + 1
            return (lv, emit("XOR3", wx, wy, wz)), (lv, emit("MAJ", wx, wy, wz))
        p = emit("XOR", wx, wy)
        lv = max(max(lx, ly) + 1, lz) + 1
        return (lv, emit("XOR", p, wz)), (lv, emit("MUX", p, wz, wx))

    # Dadda's schedule: column heights come down through 9, 6, 4, 3, 2; in a layer every column is reduced to the target
    # height with as few adders as possible (a full adder removes two wires, a half adder one), counting the carries the
    # column below sends up in the same layer
    targets = [2]
    while targets[-1] * 3 // 2 < nbits:
        targets.append(targets[-1] * 3 // 2)
    for target in reversed(targets):
        new = [[] for _ in range(2 * nbits)]
        for c in range(2 * nbits):
            col = sorted(cols[c])                       # earliest wires first: the latest of a triple is its z
            i = 0
            while len(col) - i + len(new[c]) > target:
                if len(col) - i + len(new[c]) >= target + 2 and len(col) - i >= 3:
                    sm, cy = full_adder(col[i], col[i + 1], col[i + 2])
                    i += 3
                else:                                   # half adder: XOR + AND on one level
                    (lx, wx), (ly, wy) = col[i], col[i + 1]
                    lv = max(lx, ly) + 1
                    sm, cy = (lv, emit("XOR", wx, wy)), (lv, emit("AND", wx, wy))
                    i += 2
                new[c].append(sm)
                new[c + 1].append(cy)
            new[c].extend(col[i:])
        cols = new
    prod = []
    c0 = 0
    while c0 < 2 * nbits and len(cols[c0]) <= 1:        # low columns that are already final
        prod.append(cols[c0][0][1] if cols[c0] else None)
        c0 += 1
    hi = 2 * nbits - 1
    while hi >= c0 and not cols[hi]:
        hi -= 1
    if hi >= c0:
        zero = None
        xs, ys = [], []
        for c in range(c0, hi + 1):
            col = sorted(cols[c])
            xs.append(col[0][1])
            if len(col) > 1:
                ys.append(col[1][1])
            else:                                       # a lone wire inside the addition: + constant 0 (folded by optimize)
                if zero is None:
                    zero = emit("CONST0", -1, -1)
                ys.append(zero)
        outs, top = _prefix_cells(emit, xs, ys, False)
        prod += outs + [top]
    prod = prod[: 2 * nbits]
    while len(prod) < 2 * nbits:
        prod.append(None)
    for k, w in enumerate(prod):
        if w is None:
            prod[k] = emit("CONST0", -1, -1)
    return gates, state["nxt"], a, b, prod


def less_than_tree(nbits=8):
    """unsigned a < b in logarithmic depth: a balanced tree over (LT, EQ) pairs of bit ranges, upper half first:
    LT = MUX(EQ_hi, LT_lo, LT_hi), EQ = EQ_hi AND EQ_lo (only where a parent still needs it).  A single bit as the upper
    operand needs no LT wire (MUX(a_i XNOR b_i, LT_lo, NOT a_i): where the bits differ NOT a_i is b_i).  1 + ceil(log2 nbits) levels (4 for 8 bits) at 29
    bootstraps, against less_than's 8 levels at 22.  Returns (gates, n_wires, a_wires, b_wires, out_wire)."""
    a = list(range(nbits))
    b = list(range(nbits, 2 * nbits))
    state = {"nxt": 2 * nbits}
    gates = []

    def emit(op, i0, i1, i2=-1):
        out = state["nxt"]
        state["nxt"] += 1
        gates.append(Gate(OPS[op], i0, i1, i2, out))
        return out

    def build(lo, hi, need_lt, need_eq):
        """(LT wire or None, EQ wire or None) of bits lo .. hi - 1; a single bit with need_lt False hands back b_i as the
        stand-in its parent's MUX takes"""
        if hi - lo == 1:
            eq = emit("XNOR", a[lo], b[lo]) if need_eq else None
            lt = emit("ANDNY", a[lo], b[lo]) if need_lt else None
            return lt, eq
        mid = lo + (hi - lo + 1) // 2                    # lower part [lo, mid), upper part [mid, hi)
        up_single = hi - mid == 1
        lt_lo, eq_lo = build(lo, mid, True, need_eq)
        lt_hi, eq_hi = build(mid, hi, not up_single, True)
        # a single bit as the upper operand: on the branch the MUX takes where a_i != b_i, NOT a_i equals b_i (= a_i < b_i);
        # NOT is free, and written this way the optimizer turns the cell into MAJ(NOT a_i, b_i, LT_lo): one bootstrap
        lt = emit("MUX", eq_hi, lt_lo, emit("NOT", a[mid], -1) if up_single else lt_hi) if need_lt else None
        eq = emit("AND", eq_hi, eq_lo) if need_eq else None
        return lt, eq

    lt, _ = build(0, nbits, True, False)
    return gates, state["nxt"], a, b, lt


ADDER_FORMS = {"ripple": lambda n: ripple_carry_adder(n), "mux": mux_carry_adder, "maj": maj_adder, "prefix": prefix_adder}
LESS_THAN_FORMS = {"ripple": less_than, "maj": maj_less_than, "tree": less_than_tree}
SUBTRACTOR_FORMS = {"ripple": lambda n: subtractor(n), "maj": maj_subtractor, "prefix": prefix_subtractor}
MULTIPLIER_FORMS = {"rows": lambda n: _optimized(multiplier(n)), "wallace": lambda n: _optimized(wallace_multiplier(n))}


def _optimized(built):
    """the builder's result with its netlist run through optimize (outputs = every wire group behind the operands)"""
    outs = []
    for part in built[4:]:
        outs += list(part) if isinstance(part, (list, tuple)) else [part]
    return (optimize(built[0], outs),) + tuple(built[1:])


def pick_form(forms, nbits, instances, resident_jobs=1024, optimized=False):
    """the form of lowest netlist_cost for this many instances (ties: fewest bootstraps): depth decides below a quarter
    of the resident set, bootstraps decide above it.  optimized: every candidate goes through optimize first and is
    priced -- and returned -- as rewritten (the prefix adder 48 -> 40 bootstraps, the tree comparator 29 -> 24).
    Returns (name, builder result)."""
    best = None
    for name, build in forms.items():
        r = _optimized(build(nbits)) if optimized else build(nbits)
        boots = sum(_boots(g) for g in r[0])
        key = (netlist_cost(r[0], instances, resident_jobs), boots)
        if best is None or key < best[0]:
            best = (key, name, r)
    return best[1], best[2]


def adder(nbits=8, instances=1, resident_jobs=1024):
    """the adder form to run for `instances` input pairs, through optimize: maj_adder (fewest bootstraps) for wide batches,
    prefix_adder (fewest levels) for small ones.  ripple_carry_adder is never chosen (it is the textbook form BASELINE
    configs[2] is timed on, kept as written), nor mux_carry_adder (the optimizer turns both into maj_adder's netlist)"""
    forms = {k: v for k, v in ADDER_FORMS.items() if k in ("maj", "prefix")}
    return pick_form(forms, nbits, instances, resident_jobs, True)[1]


def less_than_for(nbits=8, instances=1, resident_jobs=1024):
    return pick_form(LESS_THAN_FORMS, nbits, instances, resident_jobs, True)[1]


def multiplier_for(nbits=8, instances=1, resident_jobs=1024):
    """the multiplier form for this many instances (both forms after optimize): column compression + one prefix addition
    for small batches, the row-by-row form (fewest bootstraps) for wide ones"""
    return pick_form(MULTIPLIER_FORMS, nbits, instances, resident_jobs)[1]


def subtractor_for(nbits=8, instances=1, resident_jobs=1024):
    return pick_form(SUBTRACTOR_FORMS, nbits, instances, resident_jobs, True)[1]


def min_max_on(built_less_than, xor3_select=False):
    """(min, max) behind a comparator: min_i = MUX(lt, a_i, b_i) and max_i = MUX(lt, b_i, a_i) -- or, xor3_select,
    max_i = XOR3(a_i, b_i, min_i) (min_i XOR max_i = a_i XOR b_i): ONE bootstrap instead of the MUX's two, one level later.
    Returns (gates, n_wires, a_wires, b_wires, min_wires, max_wires)."""
    gates, nxt, a, b, lt = built_less_than
    gates, nbits = list(gates), len(a)
    mn = list(range(nxt, nxt + nbits))
    mx = list(range(nxt + nbits, nxt + 2 * nbits))
    for i in range(nbits):
        gates.append(Gate(OPS["MUX"], lt, a[i], b[i], mn[i]))
        gates.append(Gate(OPS["XOR3"], a[i], b[i], mn[i], mx[i]) if xor3_select else Gate(OPS["MUX"], lt, b[i], a[i], mx[i]))
    return gates, nxt + 2 * nbits, a, b, mn, mx


def min_max_for(nbits=8, instances=1, resident_jobs=1024):
    """(min, max) for this many instances, through optimize: every comparator form with both ways of selecting the maximum,
    the cheapest by netlist_cost -- tree comparator + two MUXes per bit for small batches (8 bits: 56 bootstraps on 5 levels),
    MAJ chain + MUX + XOR3 per bit for wide ones (32 on 10; with two MUXes 40 on 9).
    Returns (gates, n_wires, a_wires, b_wires, min_wires, max_wires)."""
    best = None
    for build in LESS_THAN_FORMS.values():
        for xor3_select in (False, True):
            r = _optimized(min_max_on(build(nbits), xor3_select))
            key = (netlist_cost(r[0], instances, resident_jobs), sum(_boots(g) for g in r[0]))
            if best is None or key < best[0]:
                best = (key, r)
    return best[1]


# ---- plaintext semantics and netlist rewriting -----------------------------------------------------

_NAMES = {v: k for k, v in OPS.items()}
_SEM2 = {
    "NAND": lambda a, b: 1 - (a & b), "AND": lambda a, b: a & b, "OR": lambda a, b: a | b,
    "NOR": lambda a, b: 1 - (a | b), "XOR": lambda a, b: a ^ b, "XNOR": lambda a, b: 1 - (a ^ b),
    "ANDNY": lambda a, b: (1 - a) & b, "ANDYN": lambda a, b: a & (1 - b),
    "ORNY": lambda a, b: (1 - a) | b, "ORYN": lambda a, b: a | (1 - b),
}


_SEM3 = {"MAJ": lambda a, b, c: ((a + b + c) >= 2) * 1, "XOR3": lambda a, b, c: a ^ b ^ c}
_FREE = ("NOT", "COPY", "CONST0", "CONST1")


def _inputs(g):
    """the wires a gate reads (unused slots are ignored whatever they hold)"""
    name = _NAMES[g.op]
    if name in ("CONST0", "CONST1"):
        return []
    if name in ("NOT", "COPY"):
        return [g.in0]
    return [g.in0, g.in1, g.in2] if name in ("MUX", "MAJ", "XOR3") else [g.in0, g.in1]


def _boots(g):
    name = _NAMES[g.op]
    return 2 if name == "MUX" else 0 if name in _FREE else 1


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
        elif name in _SEM3:
            w[g.out] = _SEM3[name](w[g.in0].astype(np.int64), w[g.in1].astype(np.int64), w[g.in2].astype(np.int64))
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
        for i in _inputs(g):
            if i == g.out:
                raise ValueError("gate reads its own output wire %d" % g.out)
        written.add(g.out)
    seen = set()
    for g in gates:
        for i in _inputs(g):
            if i in written and i not in seen:
                raise ValueError("wire %d is read before it is written" % i)
        seen.add(g.out)


def _uses(gates):
    u = {}
    for g in gates:
        for i in _inputs(g):
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


_MIRROR = {"ANDYN": "ANDNY", "ORYN": "ORNY"}            # ANDYN(a, b) = ANDNY(b, a), ORYN(a, b) = ORNY(b, a)
_SYMMETRIC = ("NAND", "AND", "OR", "NOR", "XOR", "XNOR")


def merge_duplicates(gates, outputs):
    """a gate that repeats an earlier one -- same opcode on the same wires, operand order aside where the gate is symmetric --
    becomes a COPY of it (a naive builder computes a XOR b once for the sum and once for the carry), and a gate that reads
    one wire twice is no gate: AND(x, x) = x, XOR(x, x) = 0, NAND(x, x) = NOT x, MUX(s, b, b) = b, MUX(s, s, c) = OR(s, c),
    MUX(s, b, s) = AND(s, b), MAJ(x, x, y) = x, XOR3(x, x, y) = y.  The COPYs are free and the later passes look through
    them.  Single-assignment netlists only."""
    _check_ssa(gates)
    rep, seen, out = {}, {}, []

    def unary(kind, wire, o):
        return Gate(OPS[kind], wire, -1, -1, o)

    for g in gates:
        name = _NAMES[g.op]
        i0, i1, i2 = (rep.get(w, w) if w >= 0 else w for w in (g.in0, g.in1, g.in2))
        new = None
        if name in _SEM2:
            if i0 == i1:
                f0, f1 = _SEM2[name](0, 0), _SEM2[name](1, 1)
                new = (unary("COPY", i0, g.out) if (f0, f1) == (0, 1) else unary("NOT", i0, g.out) if (f0, f1) == (1, 0)
                       else Gate(OPS["CONST1" if f0 else "CONST0"], -1, -1, -1, g.out))
            else:
                if name in _MIRROR:
                    name, i0, i1 = _MIRROR[name], i1, i0
                elif name in _SYMMETRIC and i1 < i0:
                    i0, i1 = i1, i0
                key = (name, i0, i1)
        elif name == "MUX":
            if i1 == i2:
                new = unary("COPY", i1, g.out)
            elif i0 == i1:
                name, i0, i1, i2 = "OR", min(i0, i2), max(i0, i2), -1
                key = (name, i0, i1)
            elif i0 == i2:
                name, i0, i1, i2 = "AND", min(i0, i1), max(i0, i1), -1
                key = (name, i0, i1)
            else:
                key = (name, i0, i1, i2)
        elif name in _SEM3:
            a, b, c = sorted((i0, i1, i2))
            if a == b or b == c:
                new = unary("COPY", b if name == "MAJ" else (c if a == b else a), g.out)
            else:
                i0, i1, i2 = a, b, c
                key = (name, a, b, c)
        else:                                           # NOT / COPY / CONSTANT: free, left to fold_nots / fold_constants
            new = Gate(g.op, i0 if name in ("NOT", "COPY") else -1, -1, -1, g.out)
        if new is None:
            if key in seen:
                new = unary("COPY", seen[key], g.out)
                rep[g.out] = seen[key]
            else:
                seen[key] = g.out
                new = Gate(OPS[name], i0, i1, i2 if name in ("MUX", "MAJ", "XOR3") else -1, g.out)
        elif _NAMES[new.op] == "COPY":
            rep[g.out] = new.in0
        out.append(new)
    return _drop_dead(out, outputs)


def fold_constants(gates, outputs):
    """Constant propagation: bootsCONSTANT wires (and what follows from them) are folded into their readers -- a
    two-input gate with one known input is that constant, a COPY or a NOT of the other input (free); MUX with a known
    selector is a COPY, with one known branch a two-input gate (MUX(s, 0, c) = ANDNY(s, c), MUX(s, 1, c) = OR(s, c),
    MUX(s, b, 0) = AND(s, b), MUX(s, b, 1) = ORNY(s, b): one bootstrap instead of two), with two known branches a
    COPY / NOT of the selector or a constant.  Single-assignment netlists only."""
    _check_ssa(gates)
    const = {}

    def unary(wire, neg, out):
        """gate computing `wire` (negated if neg) into out; wire known -> a constant"""
        if wire in const:
            v = const[wire] ^ neg
            const[out] = v
            return Gate(OPS["CONST1"] if v else OPS["CONST0"], -1, -1, -1, out)
        return Gate(OPS["NOT"] if neg else OPS["COPY"], wire, -1, -1, out)

    def constant(v, out):
        const[out] = v
        return Gate(OPS["CONST1"] if v else OPS["CONST0"], -1, -1, -1, out)

    res = []
    for g in gates:
        name = _NAMES[g.op]
        if name in ("CONST0", "CONST1"):
            res.append(constant(1 if name == "CONST1" else 0, g.out))
        elif name in ("NOT", "COPY"):
            res.append(unary(g.in0, 1 if name == "NOT" else 0, g.out))
        elif name in _SEM3:
            # MAJ / XOR3 with known inputs: all three -> a constant; two -> the third, a constant (MAJ of two equal) or its
            # negation; one -> a two-input gate (MAJ(x, y, 0) = AND, MAJ(x, y, 1) = OR, XOR3(x, y, 0) = XOR, XOR3(x, y, 1) = XNOR)
            ins = [g.in0, g.in1, g.in2]
            unk = [i for i in ins if i not in const]
            ones = sum(const[i] for i in ins if i in const)
            nk = 3 - len(unk)
            maj = name == "MAJ"
            if nk == 3:
                res.append(constant(int(ones >= 2) if maj else ones & 1, g.out))
            elif nk == 2:
                res.append(constant(int(ones == 2), g.out) if maj and ones != 1 else unary(unk[0], 0 if maj else ones & 1, g.out))
            elif nk == 1:
                op2 = ("OR" if ones else "AND") if maj else ("XNOR" if ones else "XOR")
                res.append(Gate(OPS[op2], unk[0], unk[1], -1, g.out))
            else:
                res.append(Gate(g.op, g.in0, g.in1, g.in2, g.out))
        elif name == "MUX":
            s, b, c = g.in0, g.in1, g.in2
            kb, kc = const.get(b), const.get(c)
            if s in const:
                res.append(unary(b if const[s] else c, 0, g.out))
            elif kb is None and kc is None:
                res.append(Gate(g.op, s, b, c, g.out))
            elif kb is not None and kc is not None:
                res.append(constant(kb, g.out) if kb == kc else unary(s, 0 if kb else 1, g.out))
            elif kb is not None:
                res.append(Gate(OPS["OR"] if kb else OPS["ANDNY"], s, c, -1, g.out))
            else:
                res.append(Gate(OPS["ORNY"] if kc else OPS["AND"], s, b, -1, g.out))
        else:
            f = _SEM2[name]
            ka, kb = const.get(g.in0), const.get(g.in1)
            if ka is not None and kb is not None:
                res.append(constant(f(ka, kb), g.out))
            elif ka is not None or kb is not None:
                r0, r1 = (f(ka, 0), f(ka, 1)) if ka is not None else (f(0, kb), f(1, kb))
                other = g.in1 if ka is not None else g.in0
                res.append(constant(r0, g.out) if r0 == r1 else unary(other, 0 if r1 else 1, g.out))
            else:
                res.append(Gate(g.op, g.in0, g.in1, -1, g.out))
    return _drop_dead(res, outputs)


def fold_nots(gates, outputs):
    """NOT and COPY are free, but in front of a bootstrapped gate they are unnecessary altogether: the ten two-input
    boots* gates are closed under input negation (AND with a negated first input IS bootsANDNY, ...), a negated
    MUX selector swaps the branches, NOT(NOT x) is a COPY, and every reader looks through COPY.  `outputs` are the
    wires the caller reads; NOT / COPY gates nobody reads afterwards are dropped.  Single-assignment netlists only."""
    _check_ssa(gates)
    src = {g.out: g for g in gates}

    def strip(wire):
        neg = 0
        while wire in src and _NAMES[src[wire].op] in ("NOT", "COPY"):
            neg ^= 1 if _NAMES[src[wire].op] == "NOT" else 0
            wire = src[wire].in0
        return wire, neg

    out = []
    for g in gates:
        name = _NAMES[g.op]
        if name in _SEM2:
            (i0, n0), (i1, n1) = strip(g.in0), strip(g.in1)
            out.append(Gate(OPS[_BY_TABLE[_table(name, n0, n1)]], i0, i1, -1, g.out))
        elif name == "MUX":
            (s, ns) = strip(g.in0)
            (b, nb), (c, nc) = strip(g.in1), strip(g.in2)
            if nb:
                b = g.in1                               # a negated branch stays behind its (free) NOT
            if nc:
                c = g.in2
            if ns:
                b, c = c, b
            out.append(Gate(g.op, s, b, c, g.out))
        elif name in _SEM3:                             # inputs look through COPY; a negated input stays behind its NOT
            ins = []
            for w in (g.in0, g.in1, g.in2):
                sw, neg = strip(w)
                ins.append(w if neg else sw)
            out.append(Gate(g.op, ins[0], ins[1], ins[2], g.out))
        elif name in ("NOT", "COPY"):
            i0, n0 = strip(g.in0)
            n0 ^= 1 if name == "NOT" else 0
            out.append(Gate(OPS["NOT"] if n0 else OPS["COPY"], i0, -1, -1, g.out))
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
            inner_ok = all(uses.get(t.out, 0) == 1 and t.out not in keep for t in (x, y)) and x.out != y.out
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


def fuse_carry(gates, outputs, extension_gates=False):
    """The carry of a textbook full adder, OR(AND(a, b), AND(XOR(a, b), c)), is MUX(XOR(a, b), c, a): where the inputs
    differ the carry-in passes, where they agree either of them is the carry.  The wire of AND(XOR(a, b), c) must be
    single-use and no output; the XOR wire stays (the sum bit reads it too), AND(a, b) stays if something else reads it.  3 bootstraps on 2 dependent levels become 2 bootstraps on
    ONE level: the literal 8-bit ripple-carry adder goes from 40 bootstraps / 17 levels to 32 / 9 (30 / 8 once the
    constant carry-in is folded).  Single-assignment netlists only."""
    _check_ssa(gates)
    src = {g.out: g for g in gates}
    uses = _uses(gates)
    keep = set(outputs)
    AND, XOR = OPS["AND"], OPS["XOR"]
    out = []
    for g in gates:
        m = None
        if _NAMES[g.op] == "OR" and g.in0 in src and g.in1 in src and g.in0 != g.in1:
            for x, y in ((src[g.in0], src[g.in1]), (src[g.in1], src[g.in0])):      # x = a AND b, y = p AND c
                if m is not None or x.op != AND or y.op != AND or x.in0 == x.in1:
                    continue
                if uses.get(y.out, 0) != 1 or y.out in keep:       # a AND b may have other readers: it then simply stays
                    continue
                for p, c in ((y.in0, y.in1), (y.in1, y.in0)):
                    q = src.get(p)
                    if m is None and q is not None and q.op == XOR and \
                            ((q.in0 == x.in0 and q.in1 == x.in1) or (q.in0 == x.in1 and q.in1 == x.in0)):
                        # the carry is the MAJORITY of (a, b, c): one bootstrap as the extension gate, two as libtfhe's MUX
                        m = Gate(OPS["MAJ"], x.in0, x.in1, c, g.out) if extension_gates else Gate(OPS["MUX"], p, c, x.in0, g.out)
        out.append(m if m is not None else Gate(g.op, g.in0, g.in1, g.in2, g.out))
    return _drop_dead(out, outputs)


def fuse_maj(gates, outputs):
    """extension gates only: a MUX whose selector is XOR(x, y) or XNOR(x, y) and one of whose branches is -- or, on that
    branch, equals -- x or y is a MAJORITY.  With d the branch taken where x and y differ and o the other one:
      o in {x, y}, or o = AND(x, y) / OR(x, y) (= x where they agree):   MAJ(x, y, d)   (the carry written as one MUX)
      d = NOT z, z in {x, y} (= the other input where they differ):      MAJ(d, other, o)
      d in {x, y}:   MAJ(NOT other, d, o) (a borrow / comparator step, MUX(XNOR(a, b), lt, b) = MAJ(NOT a, b, lt)) -- the NOT
                     takes the selector's own wire when this MUX is its only reader and it is no output
      d = ANDNY / ORNY(u, v) (= v where they differ) or ANDYN / ORYN(u, v) (= u) over the selector's inputs, read by this MUX
                     alone and no output (a tree comparator's LT_hi):  MAJ(NOT other, w, o), the NOT on d's own wire"""
    _check_ssa(gates)
    src = {g.out: g for g in gates}
    uses = _uses(gates)
    keep = set(outputs)
    out, pos = [], {}
    for g in gates:
        m = None
        q = src.get(g.in0) if _NAMES[g.op] == "MUX" else None
        if q is not None and _NAMES[q.op] in ("XOR", "XNOR") and q.in0 != q.in1:
            x, y = q.in0, q.in1
            d, o = (g.in1, g.in2) if _NAMES[q.op] == "XOR" else (g.in2, g.in1)
            hd, ho = src.get(d), src.get(o)
            dn, on = (_NAMES[hd.op] if hd is not None else None), (_NAMES[ho.op] if ho is not None else None)
            if o in (x, y) or (on in ("AND", "OR") and {ho.in0, ho.in1} == {x, y}):
                m = Gate(OPS["MAJ"], x, y, d, g.out)
            elif dn == "NOT" and hd.in0 in (x, y):
                m = Gate(OPS["MAJ"], d, y if hd.in0 == x else x, o, g.out)
            elif d in (x, y) and uses.get(q.out, 0) == 1 and q.out not in keep and q.out in pos:
                out[pos[q.out]] = Gate(OPS["NOT"], y if d == x else x, -1, -1, q.out)
                m = Gate(OPS["MAJ"], q.out, d, o, g.out)
            elif dn in ("ANDNY", "ORNY", "ANDYN", "ORYN") and {hd.in0, hd.in1} == {x, y} and uses.get(d, 0) == 1 \
                    and d not in keep and d in pos:
                w = hd.in1 if dn in ("ANDNY", "ORNY") else hd.in0
                out[pos[d]] = Gate(OPS["NOT"], y if w == x else x, -1, -1, d)
                m = Gate(OPS["MAJ"], d, w, o, g.out)
        out.append(m if m is not None else Gate(g.op, g.in0, g.in1, g.in2, g.out))
        pos[g.out] = len(out) - 1
    return _drop_dead(out, outputs)


def fuse_xor3(gates, outputs):
    """extension gates only: XOR(XOR(a, b), c) is XOR3(a, b, c) when that lets the inner wire die -- it is no output and its
    only other reader, if any, is a MUX that fuse_maj turns into MAJ(NOT ., ., .) on the inner wire itself (a subtractor's
    difference bit next to its borrow).  One bootstrap on one level instead of two on two."""
    _check_ssa(gates)
    src = {g.out: g for g in gates}
    uses = _uses(gates)
    keep = set(outputs)
    sel_of, sel_count = {}, {}
    for g in gates:
        if _NAMES[g.op] == "MUX":
            sel_of[g.in0] = g
            sel_count[g.in0] = sel_count.get(g.in0, 0) + 1

    def inner_dies(q):
        if q.out in keep:
            return False
        if uses.get(q.out, 0) == 1:
            return True
        if uses.get(q.out, 0) != 2 or sel_count.get(q.out, 0) != 1:
            return False
        m = sel_of[q.out]                               # q = XOR(x, y): the branch taken where x and y differ is in1
        return m.in1 != q.out and m.in2 != q.out and m.in1 in (q.in0, q.in1) and m.in2 not in (q.in0, q.in1)

    out = []
    for g in gates:
        m = None
        if _NAMES[g.op] == "XOR" and g.in0 != g.in1:
            for p, c in ((g.in0, g.in1), (g.in1, g.in0)):
                q = src.get(p)
                if m is None and q is not None and _NAMES[q.op] == "XOR" and q.in0 != q.in1 and inner_dies(q):
                    m = Gate(OPS["XOR3"], q.in0, q.in1, c, g.out)
        out.append(m if m is not None else Gate(g.op, g.in0, g.in1, g.in2, g.out))
    return _drop_dead(out, outputs)


def _as_tuples(gates):
    return [(g.op, g.in0, g.in1, g.in2, g.out) for g in gates]


def _normalized(gates):
    """unused input slots as -1 (the passes compare gates field by field)"""
    out = []
    for g in gates:
        ins = _inputs(g) + [-1, -1, -1]
        out.append(Gate(g.op, ins[0], ins[1], ins[2], g.out))
    return out


def optimize(gates, outputs, extension_gates=True):
    """merge_duplicates, fold_constants, fold_nots, fuse_mux, fuse_carry (and, with the extension gates, fuse_maj and fuse_xor3), repeated
    until nothing changes; returns the rewritten netlist (same wire numbering, never more bootstraps, never more levels).
    With the extension gates (default) a textbook full adder becomes XOR3 + MAJ -- the literal 8-bit ripple-carry adder 40
    bootstraps / 17 levels -> 16 / 8; extension_gates=False stays inside libtfhe's boots* family (carry as MUX: 30 / 8).
    eoc_netlist_optimize(_ex) (csrc/host.cpp) is the native twin: same passes, same order, same result."""
    # merging repeated gates can take a single-use wire away from a later pattern (two sums sharing one a XOR b): both
    # pipelines run and the better result is kept -- fewest bootstraps, then fewest levels, then fewest gates; merged on ties
    best = None
    for merge in (True, False):
        cur = _normalized(gates)
        for _ in range(8):
            nxt = merge_duplicates(cur, outputs) if merge else cur
            nxt = fold_nots(fold_constants(nxt, outputs), outputs)
            nxt = fuse_carry(fuse_mux(nxt, outputs), outputs, extension_gates)
            if extension_gates:
                nxt = fuse_xor3(fuse_maj(nxt, outputs), outputs)
            if _as_tuples(nxt) == _as_tuples(cur):
                break
            cur = nxt
        key = (sum(_boots(g) for g in cur), bootstrap_depth(cur), len(cur))
        if best is None or key < best[0]:
            best = (key, cur)
    return best[1]


# ---- levels and the level-cost estimate (what picks a circuit form for an instance count) ------------------------

def levels(gates):
    """level of every gate as eoc_circuit_run_device assigns it (eoc_levelise, csrc/host.cpp; 1-based).  A level is two
    time slots -- 2L - 1: its pre-pass (free gates), 2L: its bootstrapped gates -- and a gate takes the earliest slot of
    its kind after the writers of its inputs (RAW), the readers of its output's old value (WAR) and that output's last
    writer (WAW): a free gate costs no level."""
    wr, rd, lev = {}, {}, []
    for g in gates:
        ins = _inputs(g)
        t = max([wr.get(g.out, 0), rd.get(g.out, 0)] + [wr.get(i, 0) for i in ins]) + 1
        if (t & 1 == 1) != (_boots(g) == 0):
            t += 1
        for i in ins:
            rd[i] = max(rd.get(i, 0), t)
        wr[g.out] = t
        lev.append((t + 1) // 2)
    return lev


def bootstrap_depth(gates):
    """dependent levels that hold at least one blind rotation (free gates ride along)"""
    lev = levels(gates)
    return len({lv for g, lv in zip(gates, lev) if _boots(g)})


def netlist_cost(gates, instances, resident_jobs=1024):
    """Estimated run time of a netlist over `instances` instances, in units of 0.1 ms on one MI355X (Set A): a level
    holding J = instances x jobs blind rotations runs as J // R full launches (3.0 ms each, R = resident_jobs, 4 per
    compute unit) and one partly filled launch of J % R rotations costing 1.4 ms + 1.6 ms x max(J % R, R / 4) / R --
    the measured 1.8 / 2.3 / 3.0 ms at 256 / 512 / 1024 gates (profiles/r05_narrow_gate.txt): below a quarter of the
    resident set a level costs the same whatever its width, so DEPTH is the cost of a small batch and the number of
    BOOTSTRAPS that of a large one.  Native twin: eoc_netlist_cost."""
    R = max(4, int(resident_jobs))
    jobs = {}
    for g, lv in zip(gates, levels(gates)):
        w = _boots(g)
        if w:
            jobs[lv] = jobs.get(lv, 0) + w
    cost = 0
    for lv in sorted(jobs):
        J = jobs[lv] * int(instances)
        full, rem = divmod(J, R)
        cost += 30 * full
        if rem:
            cost += 14 + (16 * max(rem, R // 4) + R - 1) // R
    return cost


def noise_margin(gates, n_inputs_var, v_br, v_ks, v_modswitch):
    """Smallest decision margin, in standard deviations, over every blind rotation of a netlist -- the noise budget a
    rewrite must not exhaust.  Variances in torus units: fresh inputs carry n_inputs_var (sigma_ks^2); a two-input gate's
    output V_BR + V_KS, a MUX's 2 V_BR + V_KS (two rotations summed before ONE key switch), NOT / COPY pass their input's
    on, constants have none.  The phase a blind rotation sees: AND / OR family  +-1/8 + a + b  (margin 1/8 to the nearest
    decision boundary, variance V_a + V_b), XOR / XNOR  +-1/4 + 2 (a + b)  (margin 1/4, variance 4 (V_a + V_b)), MUX two
    AND-type rotations (selector + one branch each); every rotation adds the mod-switch rounding v_modswitch =
    (1 + |s|) / (48 N^2).  Returns (min margin in sigmas, index of the gate that has it)."""
    var = {}
    worst, where = float("inf"), -1
    for k, g in enumerate(gates):
        name = _NAMES[g.op]
        v = lambda w: var.get(w, n_inputs_var)
        if name in ("CONST0", "CONST1"):
            var[g.out] = 0.0
            continue
        if name in ("NOT", "COPY"):
            var[g.out] = v(g.in0)
            continue
        if name == "MUX":
            rot = [(0.125, v(g.in0) + v(g.in1)), (0.125, v(g.in0) + v(g.in2))]
            var[g.out] = 2 * v_br + v_ks
        elif name in ("XOR", "XNOR"):
            rot = [(0.25, 4 * (v(g.in0) + v(g.in1)))]
            var[g.out] = v_br + v_ks
        elif name == "XOR3":                            # -2 (a + b + c): phases +-1/4
            rot = [(0.25, 4 * (v(g.in0) + v(g.in1) + v(g.in2)))]
            var[g.out] = v_br + v_ks
        elif name == "MAJ":                             # a + b + c: phases +-1/8, +-3/8
            rot = [(0.125, v(g.in0) + v(g.in1) + v(g.in2))]
            var[g.out] = v_br + v_ks
        else:
            rot = [(0.125, v(g.in0) + v(g.in1))]
            var[g.out] = v_br + v_ks
        for margin, vin in rot:
            s = margin / (vin + v_modswitch) ** 0.5
            if s < worst:
                worst, where = s, k
    return worst, where

