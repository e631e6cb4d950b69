"""Netlist builders for the circuits BASELINE.json names (they do not exist in ao-tfhe/tfhe.lua --
SURVEY.md 0.1 -- so this repo defines them, in the reference's style; INTEGRATION.md has the Lua text).

A netlist is a list of eoc_tfhe_amd.Gate over integer wire ids; wires [w][instance][n+1] hold LWE
samples; eoc_circuit_run(_device) levelises and batches every level over all instances.
"""
from . import OPS, Gate


def ripple_carry_adder(nbits=8):
    """a[0..nbits) + b[0..nbits) -> s[0..nbits], LSB first.  5 gates per bit (2 XOR, 2 AND, 1 OR) except
    bit 0 (XOR + AND): 5*nbits - 3 bootstraps; BASELINE config 3 counts the uniform 5/bit = 40.
    Wires: a = 0..nbits-1, b = nbits..2nbits-1, sum = 2nbits..3nbits (nbits+1 wires), then temporaries.
    Returns (gates, n_wires, a_wires, b_wires, sum_wires)."""
    a = list(range(nbits))
    b = list(range(nbits, 2 * nbits))
    s = list(range(2 * nbits, 3 * nbits + 1))
    nxt = 3 * nbits + 1
    gates = []
    carry = None
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
    if nbits == 1:
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
