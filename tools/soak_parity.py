"""Randomised parity soak (GPU box): random parameter set / LWE dimension / batch width / opcode mix / netlist through the
device API, the host-buffer API and the asynchronous submit/wait API, every result compared bit for bit with the CPU
oracle.

  python tools/soak_parity.py [seconds] [seed]      free-running, prints one line per case and a final tally
  soak(...) below                                   what tests/test_gpu_soak.py calls with a fixed seed: a bounded,
                                                    repeatable slice that visits all five kinds

Kinds: uniform (one opcode, device pointers), mixed (opcode per row), circuit (random netlist with hazards), optimized
(round 6: a random SINGLE-ASSIGNMENT netlist with constants, NOT / COPY chains, MUXes and textbook full-adder carries sent
through eoc_netlist_optimize; the rewritten netlist runs on the GPU and must equal the oracle's evaluation of that rewritten
netlist bit for bit, and its outputs must DECRYPT to what the ORIGINAL netlist computes on the plaintext bits), host
(a FRESH global context whose first call is a host-buffer batch: the path that grows the workspace -- where round 3's
NULL-stream memset race lived), async (fresh context, three submissions two deep on pinned buffers)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

KINDS = ("uniform", "mixed", "circuit", "optimized", "host", "async")


def soak(budget_s=120.0, seed=1, kinds=KINDS, max_contexts=None, cases_per_context=4, round_robin=False, log=print):
    """Returns (cases, mismatches, per_kind_counts).  Stops at the time budget or after max_contexts key contexts.
    round_robin: kinds are visited in order instead of drawn at random (a short slice then covers every kind)."""
    import torch
    import eoc_tfhe_amd as eoc
    import oracle_lib as ol
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda", 0)
    BOOT = [eoc.OPS[k] for k in ("NAND", "AND", "OR", "NOR", "XOR", "XNOR", "ANDNY", "ANDYN", "ORNY", "ORYN", "MUX", "MAJ", "XOR3")]
    THREE = (eoc.OPS["MUX"], eoc.OPS["MAJ"], eoc.OPS["XOR3"])
    FREE = [eoc.OPS[k] for k in ("NOT", "COPY", "CONST0", "CONST1")]
    t_end = time.time() + budget_s
    cases = bad = contexts = 0
    per_kind = {k: 0 for k in kinds}
    while time.time() < t_end and (max_contexts is None or contexts < max_contexts):
        contexts += 1
        pset = int(rng.integers(0, 2))
        n = int(rng.integers(4, 33))
        kseed = int(rng.integers(1, 1 << 30))
        p = eoc.default_params(pset)
        p.n = n
        sk = eoc.SecretKey(p, kseed)
        eng = eoc.Engine(p)
        eng.load_cloud_key(sk)
        orc = ol.Oracle(pset, kseed, n_override=n)
        for _ in range(cases_per_context):
            if time.time() >= t_end:
                break
            kind = kinds[cases % len(kinds)] if round_robin else str(rng.choice(list(kinds)))
            # widths around every launch-shape seam: the pair kernel's resident set (1024), the wide kernel's (2048), a full
            # wide launch + a pair-kernel remainder (2049 ... 3072), + a wide remainder (3073 ...)
            count = int(rng.choice([1, 2, 63, 64, 65, 127, 1023, 1024, 1025, 2047, 2048, 2049, 3073, int(rng.integers(1, 2600)),
                                    int(rng.integers(2600, 5200))]))
            c = [sk.encrypt_bits(rng.integers(0, 2, count).astype(np.uint8), int(rng.integers(1, 1 << 30)), 0) for _ in range(3)]
            if kind == "circuit":
                S = int(rng.choice([1, 3, 17, 64, 130]))
                n_wires, n_gates = 12, int(rng.integers(3, 25))
                gates = []
                for _g in range(n_gates):
                    op = int(rng.choice(BOOT + FREE))
                    i0, i1, i2, o = (int(x) for x in rng.integers(0, n_wires, 4))
                    ni = (0 if op in (eoc.OPS["CONST0"], eoc.OPS["CONST1"]) else 1 if op in (eoc.OPS["NOT"], eoc.OPS["COPY"])
                          else 3 if op in THREE else 2)
                    gates.append(eoc.Gate(op, i0 if ni >= 1 else -1, i1 if ni >= 2 else -1, i2 if ni >= 3 else -1, o))
                wires = np.stack([sk.encrypt_bits(rng.integers(0, 2, S).astype(np.uint8), int(rng.integers(1, 1 << 30)), 0)
                                  for _w in range(n_wires)])
                want = wires.copy()
                for g in gates:
                    i0 = want[g.in0] if g.in0 >= 0 else np.zeros_like(want[g.out])
                    want[g.out] = orc.gate_batch(g.op, i0, None if g.in1 < 0 else want[g.in1], None if g.in2 < 0 else want[g.in2])
                d = torch.from_numpy(wires).to(dev)
                eng.circuit_run_device(gates, d.data_ptr(), n_wires, S)
                torch.cuda.synchronize()
                got = d.cpu().numpy()
                desc = f"circuit gates={n_gates} S={S}"
            elif kind == "optimized":
                from eoc_tfhe_amd import circuits
                S = int(rng.choice([1, 5, 33, 130]))
                n_in, n_gates = 5, int(rng.integers(6, 40))
                gates, avail = [], list(range(n_in))
                pick = lambda: int(avail[int(rng.integers(0, len(avail)))])
                while len(gates) < n_gates:
                    out = n_in + len(gates)
                    r = int(rng.integers(0, 20))
                    if r == 0 and n_gates - len(gates) >= 4:       # a textbook full-adder carry: the rewrite's pattern
                        x, y, z = pick(), pick(), pick()
                        gates += [eoc.Gate(eoc.OPS["XOR"], x, y, -1, out), eoc.Gate(eoc.OPS["AND"], x, y, -1, out + 1),
                                  eoc.Gate(eoc.OPS["AND"], out, z, -1, out + 2), eoc.Gate(eoc.OPS["OR"], out + 1, out + 2, -1, out + 3)]
                        avail += [out, out + 3]
                        continue
                    if r == 1 and n_gates - len(gates) >= 3:       # a borrow / comparator step, sometimes with its difference bit
                        x, y, z = pick(), pick(), pick()
                        sel = "XOR" if rng.integers(0, 2) else "XNOR"
                        differ, same = (x if rng.integers(0, 2) else y, z) if rng.integers(0, 2) else (z, x if rng.integers(0, 2) else y)
                        br = (differ, same) if sel == "XOR" else (same, differ)
                        gates += [eoc.Gate(eoc.OPS[sel], x, y, -1, out), eoc.Gate(eoc.OPS["MUX"], out, br[0], br[1], out + 1)]
                        avail.append(out + 1)
                        if rng.integers(0, 2):
                            gates.append(eoc.Gate(eoc.OPS["XOR"], out, z, -1, out + 2))
                            avail.append(out + 2)
                        if rng.integers(0, 4) == 0:
                            avail.append(out)                      # the selector stays visible: the NOT cannot take its wire
                        continue
                    if r == 3 and n_gates - len(gates) >= 3:       # a MUX branch that equals a selector input ON that branch
                        x, y, z = pick(), pick(), pick()
                        sel = "XOR" if rng.integers(0, 2) else "XNOR"
                        kind_b = str(rng.choice(["NOT", "ANDNY", "ANDYN", "ORNY", "ORYN", "AND", "OR"]))
                        if kind_b == "NOT":
                            gates += [eoc.Gate(eoc.OPS[sel], x, y, -1, out), eoc.Gate(eoc.OPS["NOT"], x, -1, -1, out + 1)]
                        else:
                            gates += [eoc.Gate(eoc.OPS[sel], x, y, -1, out), eoc.Gate(eoc.OPS[kind_b], x, y, -1, out + 1)]
                        on_differ = kind_b not in ("AND", "OR")                         # the branch the stand-in belongs on
                        br = (out + 1, z) if (sel == "XOR") == on_differ else (z, out + 1)
                        gates.append(eoc.Gate(eoc.OPS["MUX"], out, br[0], br[1], out + 2))
                        avail += [out + 2] + ([out] if rng.integers(0, 2) else []) + ([out + 1] if rng.integers(0, 4) == 0 else [])
                        continue
                    if r == 2 and gates:                           # an earlier gate again, operands swapped where it has two
                        e = gates[int(rng.integers(0, len(gates)))]
                        two = e.in1 >= 0 and e.in2 < 0 and e.op in (eoc.OPS["AND"], eoc.OPS["OR"], eoc.OPS["XOR"], eoc.OPS["XNOR"],
                                                                     eoc.OPS["NAND"], eoc.OPS["NOR"])
                        gates.append(eoc.Gate(e.op, e.in1 if two else e.in0, e.in0 if two else e.in1, e.in2, out))
                        avail.append(out)
                        continue
                    op = int(rng.choice(BOOT + FREE + [eoc.OPS["NOT"], eoc.OPS["MUX"]]))
                    ni = (0 if op in (eoc.OPS["CONST0"], eoc.OPS["CONST1"]) else 1 if op in (eoc.OPS["NOT"], eoc.OPS["COPY"])
                          else 3 if op in THREE else 2)
                    gates.append(eoc.Gate(op, pick() if ni >= 1 else -1, pick() if ni >= 2 else -1, pick() if ni >= 3 else -1, out))
                    avail.append(out)
                n_wires = n_in + len(gates)
                outs = [int(v) for v in rng.choice(avail[n_in:], size=min(4, len(avail) - n_in), replace=False)]
                opt = eoc.netlist_optimize(gates, outs, extension_gates=bool(rng.integers(0, 2)))
                bits = rng.integers(0, 2, (n_in, S)).astype(np.uint8)
                wires = np.zeros((n_wires, S, n + 1), np.int32)
                for w_ in range(n_in):
                    wires[w_] = sk.encrypt_bits(bits[w_], int(rng.integers(1, 1 << 30)), 0)
                want = wires.copy()
                for g in opt:
                    i0 = want[g.in0] if g.in0 >= 0 else np.zeros_like(want[g.out])
                    want[g.out] = orc.gate_batch(g.op, i0, None if g.in1 < 0 else want[g.in1], None if g.in2 < 0 else want[g.in2])
                d = torch.from_numpy(wires).to(dev)
                if opt:
                    eng.circuit_run_device(opt, d.data_ptr(), n_wires, S)
                torch.cuda.synchronize()
                got = d.cpu().numpy()
                plain = np.zeros((n_wires, S), np.uint8)
                plain[:n_in] = bits
                plain = circuits.evaluate_plain(gates, plain)
                for o in outs:                                      # the rewritten netlist computes what the original does
                    if not np.array_equal(sk.decrypt_bits(got[o]), plain[o]):
                        got = got.copy()
                        got[o, 0, 0] ^= 1                           # reported as a mismatch below
                desc = f"optimized gates={len(gates)}->{len(opt)} boots={eoc.circuit_bootstraps(gates)}->{eoc.circuit_bootstraps(opt)} S={S}"
            else:
                ops = None
                op = int(rng.choice(BOOT))
                if kind == "mixed":
                    ops = rng.choice(np.array(BOOT + FREE, np.uint8), count)
                    op = 0
                want = orc.gate_batch(op, c[0], c[1], c[2], ops=ops)
                if kind == "async":
                    # three submissions kept two deep in flight on pinned buffers (eoc_gate_batch_submit / _wait)
                    eoc.gpu_shutdown()
                    eoc.gpu_init(p, devices=[0] * int(rng.integers(1, 4)))
                    eoc.upload_cloud_key(sk)
                    pins = [[eoc.PinnedArray(c[0].shape) for _ in range(4)] for _b in range(3)]
                    wants, tks = [], []
                    for b, pb in enumerate(pins):
                        for k in range(3):
                            pb[k].array[:] = np.roll(c[k], b, axis=0)
                        wants.append(orc.gate_batch(op, pb[0].array, pb[1].array, pb[2].array, ops=ops))
                        if b >= 2:
                            eoc.gate_batch_wait(tks[b - 2])
                        tks.append(eoc.gate_batch_submit(op, pb[0].array, pb[1].array, pb[2].array, ops=ops, out=pb[3].array))
                    for t in tks:
                        eoc.gate_batch_wait(t)
                    got = np.concatenate([pb[3].array for pb in pins])
                    want = np.concatenate(wants)
                    for pb in pins:
                        for a in pb:
                            a.free()
                    eoc.gpu_shutdown()
                elif kind == "host":
                    eoc.gpu_shutdown()
                    eoc.gpu_init(p, devices=[0] * int(rng.integers(1, 4)))
                    eoc.upload_cloud_key(sk)
                    got = eoc.gate_batch(op, c[0], c[1], c[2], ops=ops)
                    eoc.gpu_shutdown()
                else:
                    d = [torch.from_numpy(x).to(dev) for x in c]
                    out = torch.empty_like(d[0])
                    eng.gate_batch_device(op, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), out.data_ptr(), count, ops=ops)
                    torch.cuda.synchronize()
                    got = out.cpu().numpy()
                desc = f"{kind} op={op} count={count}"
            ok = bool(np.array_equal(got, want))
            cases += 1
            per_kind[kind] += 1
            bad += 0 if ok else 1
            log(f"[{cases}] set={'AB'[pset]} n={n} seed={kseed} {desc}: {'ok' if ok else 'MISMATCH'}")
        eng.close()
    return cases, bad, per_kind


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    kinds = tuple(os.environ.get("SOAK_KINDS", ",".join(KINDS)).split(","))
    cases, bad, per_kind = soak(budget, seed, kinds, log=lambda s: print(s, flush=True))
    print(f"soak: {cases} cases, {bad} mismatches, per kind {per_kind}")
    sys.exit(1 if bad else 0)
