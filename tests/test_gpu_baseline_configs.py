"""BASELINE.json configs 3, 4, 5 at their FULL sizes on one GPU, checked through size-independent properties
(decrypted results vs plaintext) plus an oracle spot-check on a slice.  Config 2 at full size lives in
test_gpu_parity.py; config 1 is the oracle's golden test (CPU)."""
import numpy as np
import pytest

import oracle_lib as ol
from gpu_util import sync, to_dev, torch_cuda

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eoc(built_lib):
    torch_cuda()
    import eoc_tfhe_amd
    return eoc_tfhe_amd


@pytest.fixture(scope="module")
def rig(eoc):
    p = eoc.default_params(0)          # Set A, key seed 1
    sk = eoc.SecretKey(p, 1)
    eng = eoc.Engine(p)
    eng.load_cloud_key(sk)
    return p, sk, eng


def _enc_planes(sk, bits, seed0):
    """bits [S, nbits] -> wires [nbits][S][n+1], one encryption stream per bit plane"""
    return np.stack([sk.encrypt_bits(bits[:, i], seed0 + i, 0) for i in range(bits.shape[1])])


def _oracle_netlist(orc, gates, wires):
    """gate-by-gate evaluation of a netlist by the oracle on host wires [n_wires][S][n+1]"""
    w = wires.copy()
    for g in gates:
        i0 = w[g.in0] if g.in0 >= 0 else np.zeros_like(w[g.out])
        w[g.out] = orc.gate_batch(g.op, i0, None if g.in1 < 0 else w[g.in1], None if g.in2 < 0 else w[g.in2])
    return w


ORACLE_INSTANCES = 16      # instances of configs 3 and 5 whose EVERY written wire is compared with the oracle


def test_config3_adder_4096_pairs(eoc, rig):
    """8-bit ripple-carry add over 4096 input pairs (operands from seed 3, LSB first)."""
    from eoc_tfhe_amd import circuits
    p, sk, eng = rig
    torch = torch_cuda()
    # the literal configuration: 5 bootstrapped gates per bit = 40 per pair = 163 840 bootstraps; the
    # 37-gate form with a half adder at bit 0 is covered by tests/test_gpu_circuits.py and tests/test_gpu_cloud_server.py
    gates, n_wires, aw, bw, sw = circuits.ripple_carry_adder(8, carry_in_zero=True)
    assert eoc.circuit_bootstraps(gates) == 40
    S = 4096
    rng = np.random.default_rng(3)
    A, B = rng.integers(0, 256, S), rng.integers(0, 256, S)
    wires = torch.zeros((n_wires, S, p.n + 1), dtype=torch.int32, device="cuda")
    abits = ((A[:, None] >> np.arange(8)) & 1).astype(np.uint8)
    bbits = ((B[:, None] >> np.arange(8)) & 1).astype(np.uint8)
    wires[aw[0]: aw[0] + 8] = to_dev(_enc_planes(sk, abits, 1000))
    wires[bw[0]: bw[0] + 8] = to_dev(_enc_planes(sk, bbits, 2000))
    inputs = wires[:, :ORACLE_INSTANCES].cpu().numpy()
    before = eng.stats()["bootstraps"]
    eng.circuit_run_device(gates, wires.data_ptr(), n_wires, S)
    sync()
    assert eng.stats()["bootstraps"] - before == 40 * S == 163840
    sums = wires[sw[0]: sw[0] + 9].cpu().numpy()
    total = sum(sk.decrypt_bits(sums[i]).astype(np.int64) << i for i in range(9))
    assert np.array_equal(total, A + B)
    # ciphertexts vs oracle: every wire the netlist writes, first 16 instances (640 bootstraps)
    want = _oracle_netlist(ol.Oracle(0, 1), gates, inputs)
    got = wires[:, :ORACLE_INSTANCES].cpu().numpy()
    for g in gates:
        assert np.array_equal(got[g.out], want[g.out]), f"wire {g.out} (op {g.op})"


def _run_adder(eoc, rig, gates, n_wires, aw, bw, sw, S, seed):
    """encrypt S random 8-bit pairs, run the netlist on the GPU; returns (A, B, wires tensor, inputs of the first 16)"""
    p, sk, eng = rig
    torch = torch_cuda()
    rng = np.random.default_rng(seed)
    A, B = rng.integers(0, 256, S), rng.integers(0, 256, S)
    wires = torch.zeros((n_wires, S, p.n + 1), dtype=torch.int32, device="cuda")
    wires[aw[0]: aw[0] + 8] = to_dev(_enc_planes(sk, ((A[:, None] >> np.arange(8)) & 1).astype(np.uint8), 1000))
    wires[bw[0]: bw[0] + 8] = to_dev(_enc_planes(sk, ((B[:, None] >> np.arange(8)) & 1).astype(np.uint8), 2000))
    inputs = wires[:, :ORACLE_INSTANCES].cpu().numpy()
    before = eng.stats()["bootstraps"]
    eng.circuit_run_device(gates, wires.data_ptr(), n_wires, S)
    sync()
    assert eng.stats()["bootstraps"] - before == eoc.circuit_bootstraps(gates) * S
    sums = wires[sw[0]: sw[0] + 9].cpu().numpy()
    total = sum(sk.decrypt_bits(sums[i]).astype(np.int64) << i for i in range(9))
    assert np.array_equal(total, A + B)
    return wires, inputs


@pytest.mark.parametrize("S", [4096, 8])
def test_config3_adder_rewritten_and_log_depth_forms(eoc, rig, S):
    """VERDICT r5 task 1: BASELINE configs[2]'s literal 40-gate netlist through eoc_netlist_optimize -- with the extension
    gates (a full adder = XOR3 + MAJ: 16 bootstraps on 8 levels) and inside libtfhe's gate family (the carry as MUX +
    constant folding: 30 on 8) --, the XOR3 / MAJ and MUX-carry adders written directly and the parallel-prefix adder
    (48 bootstraps on 5 levels), over 4096 pairs and over 8: all sums decrypt, and EVERY wire the rewritten netlist writes
    equals the oracle's evaluation of that rewritten netlist on the first 16 (or all 8) instances, bit for bit"""
    from eoc_tfhe_amd import circuits
    lit, n_wires, aw, bw, sw = circuits.ripple_carry_adder(8, carry_in_zero=True)
    as_t = lambda gs: [(g.op, g.in0, g.in1, g.in2, g.out) for g in gs]
    opt = eoc.netlist_optimize(lit, sw)
    boots_only = eoc.netlist_optimize(lit, sw, extension_gates=False)
    assert as_t(opt) == as_t(circuits.optimize(lit, sw)) and as_t(boots_only) == as_t(circuits.optimize(lit, sw, False))
    assert eoc.circuit_bootstraps(opt) == 16 and eoc.netlist_levels(opt)[2] == 8
    assert eoc.circuit_bootstraps(boots_only) == 30 and eoc.netlist_levels(boots_only)[2] == 8
    forms = [("optimized literal", (opt, n_wires, aw, bw, sw)), ("mux carry", circuits.mux_carry_adder(8)),
             ("prefix", circuits.prefix_adder(8)), ("optimized literal, boots* gates only", (boots_only, n_wires, aw, bw, sw)),
             ("maj", circuits.maj_adder(8))]
    assert eoc.netlist_levels(forms[2][1][0])[2] == 5
    orc = ol.Oracle(0, 1)
    for name, (gates, nw, a, b, s) in forms:
        wires, inputs = _run_adder(eoc, rig, gates, nw, a, b, s, S, 3)
        k = min(S, ORACLE_INSTANCES)
        want = _oracle_netlist(orc, gates, inputs[:, :k])
        got = wires[:, :k].cpu().numpy()
        for g in gates:
            assert np.array_equal(got[g.out], want[g.out]), f"{name}: wire {g.out} (op {g.op})"
    # the chooser: 8 instances take the prefix form, 4096 the XOR3 / MAJ form (the engine's own resident set)
    assert circuits.pick_form({k: v for k, v in circuits.ADDER_FORMS.items() if k != "ripple"}, 8, S,
                              rig[2].resident_jobs() // 2)[0] == ("prefix" if S == 8 else "maj")


@pytest.mark.parametrize("S", [1024, 8])
def test_less_than_tree_and_ripple_bit_exact(eoc, rig, S):
    """the log-depth comparator (29 bootstraps on 4 levels), the ripple one (22 on 8) and the MAJ chain (8 on 8: the borrow
    of a - b, one bootstrap per bit): decrypt to a < b, every written wire of the first 16 instances equals the oracle's"""
    from eoc_tfhe_amd import circuits
    p, sk, eng = rig
    torch = torch_cuda()
    orc = ol.Oracle(0, 1)
    rng = np.random.default_rng(17)
    A, B = rng.integers(0, 256, S), rng.integers(0, 256, S)
    B[::5] = A[::5]                                             # equal operands: a < b is false through every EQ
    for build in (circuits.less_than_tree, circuits.less_than, circuits.maj_less_than):
        gates, n_wires, aw, bw, lt = build(8)
        wires = torch.zeros((n_wires, S, p.n + 1), dtype=torch.int32, device="cuda")
        wires[aw[0]: aw[0] + 8] = to_dev(_enc_planes(sk, ((A[:, None] >> np.arange(8)) & 1).astype(np.uint8), 6000))
        wires[bw[0]: bw[0] + 8] = to_dev(_enc_planes(sk, ((B[:, None] >> np.arange(8)) & 1).astype(np.uint8), 7000))
        k = min(S, ORACLE_INSTANCES)
        inputs = wires[:, :k].cpu().numpy()
        eng.circuit_run_device(gates, wires.data_ptr(), n_wires, S)
        sync()
        assert np.array_equal(sk.decrypt_bits(wires[lt].cpu().numpy()), (A < B).astype(np.uint8)), build.__name__
        want = _oracle_netlist(orc, gates, inputs)
        got = wires[:, :k].cpu().numpy()
        for g in gates:
            assert np.array_equal(got[g.out], want[g.out]), f"{build.__name__}: wire {g.out} (op {g.op})"


def test_config5_string_equality_1024x32(eoc, rig):
    """ASCII-string equality (per-bit XOR + OR tree + NOT) on 1024 pairs of 32-byte strings, half equal."""
    from eoc_tfhe_amd import circuits
    p, sk, eng = rig
    torch = torch_cuda()
    gates, n_wires, xw, yw, out = circuits.string_equal(32)
    assert eoc.circuit_bootstraps(gates) == 511
    S = 1024
    rng = np.random.default_rng(5)
    X = rng.integers(32, 127, (S, 32)).astype(np.uint8)
    Y = X.copy()
    diff = np.arange(S) % 2 == 1
    pos = rng.integers(0, 32, S)
    Y[diff, pos[diff]] ^= (1 << rng.integers(0, 7, S)[diff]).astype(np.uint8)
    xb = np.unpackbits(X, axis=1, bitorder="little")
    yb = np.unpackbits(Y, axis=1, bitorder="little")
    wires = torch.zeros((n_wires, S, p.n + 1), dtype=torch.int32, device="cuda")
    wires[xw[0]: xw[0] + 256] = to_dev(_enc_planes(sk, xb, 3000))
    wires[yw[0]: yw[0] + 256] = to_dev(_enc_planes(sk, yb, 4000))
    inputs = wires[:, :ORACLE_INSTANCES].cpu().numpy()
    eng.circuit_run_device(gates, wires.data_ptr(), n_wires, S)
    sync()
    got = sk.decrypt_bits(wires[out].cpu().numpy())

    assert np.array_equal(got, (~diff).astype(np.uint8))
    # ciphertexts vs oracle: every wire of the 511-bootstrap circuit, first 16 pairs (8 equal, 8 different)
    want = _oracle_netlist(ol.Oracle(0, 1), gates, inputs)
    gotw = wires[:, :ORACLE_INSTANCES].cpu().numpy()
    for g in gates:
        assert np.array_equal(gotw[g.out], want[g.out]), f"wire {g.out} (op {g.op})"


def test_config4_mixed_gates_all_eight_shards(eoc, rig):
    """1M mixed gates {NAND, XOR, MUX} sharded over 8 GPUs = 131072 gates per GPU (eoc_tfhe_amd.distributed.shard),
    op stream from seed 4.  One GPU evaluates ALL EIGHT blocks one after the other, exactly as the eight ranks would:
    every gate of the 2^20 is decrypt-checked, and a slice of every block (all three opcodes) is compared with the
    oracle bit for bit."""
    from eoc_tfhe_amd.distributed import config3_block
    p, sk, eng = rig
    torch = torch_cuda()
    total = 1 << 20
    orc = ol.Oracle(0, 1)
    covered = 0
    for rank in range(8):
        lo, hi, ops, _ = config3_block(total, rank, 8)    # the blocks bench.py --gpus 8 times, one per rank
        cnt = hi - lo
        covered += cnt                     # arbitrary opcode order: the engine groups equal opcodes on the device
        b = [np.random.default_rng(40 + 10 * rank + k).integers(0, 2, cnt).astype(np.uint8) for k in range(3)]
        c = [to_dev(sk.encrypt_bits(b[k], 5000 + k, lo)) for k in range(3)]
        out = torch.empty_like(c[0])
        before = eng.stats()["bootstraps"]
        eng.gate_batch_device(0, c[0].data_ptr(), c[1].data_ptr(), c[2].data_ptr(), out.data_ptr(), cnt, ops=ops)
        sync()
        n_mux = int((ops == eoc.OPS["MUX"]).sum())
        assert eng.stats()["bootstraps"] - before == cnt + n_mux      # MUX = 2 blind rotations
        got = out.cpu().numpy()
        want = np.where(ops == eoc.OPS["NAND"], 1 - (b[0] & b[1]),
                        np.where(ops == eoc.OPS["XOR"], b[0] ^ b[1], np.where(b[0] == 1, b[1], b[2])))
        assert np.array_equal(sk.decrypt_bits(got), want), f"shard {rank}"
        idx = np.concatenate([np.flatnonzero(ops == o)[:4] for o in np.unique(ops)])
        h = [x.cpu().numpy()[idx] for x in c]
        assert np.array_equal(got[idx], orc.gate_batch(0, h[0], h[1], h[2], ops=ops[idx])), f"shard {rank}"
        del c, out
    assert covered == total


def test_single_launch_262144_gates(eoc, rig):
    """one launch far beyond the headline batch (2^18 gates: 131072 workgroups, grid dimensions > 65535):
    every gate decrypts correctly"""
    p, sk, eng = rig
    torch = torch_cuda()
    cnt = 1 << 18
    rng = np.random.default_rng(18)
    b0, b1 = rng.integers(0, 2, cnt).astype(np.uint8), rng.integers(0, 2, cnt).astype(np.uint8)
    c0, c1 = to_dev(sk.encrypt_bits(b0, 6000, 0)), to_dev(sk.encrypt_bits(b1, 6001, 0))
    out = torch.empty_like(c0)
    eng.gate_batch_device(eoc.OPS["XNOR"], c0.data_ptr(), c1.data_ptr(), None, out.data_ptr(), cnt)
    sync()
    assert np.array_equal(sk.decrypt_bits(out.cpu().numpy()), 1 - (b0 ^ b1))
