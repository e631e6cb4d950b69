"""Measured run time of every circuit form against the level-cost estimate that picks between them (eoc_netlist_cost).
For each word-level operation (8-bit add, subtract, less-than, multiply) and each instance count S, every form is run on
device-resident wires (median of 5 calls), decrypt-checked, and printed beside its estimate (0.1 ms units -> ms); the form
the facades would pick (lowest estimate) is marked `*`, the measured-fastest `<`.  Usage (GPU box): python tools/form_sweep.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eoc_tfhe_amd as eoc  # noqa: E402
from eoc_tfhe_amd import circuits as c  # noqa: E402

p = eoc.default_params(0)
sk = eoc.SecretKey(p, 1)
eng = eoc.Engine(p)
eng.load_cloud_key(sk)
R = eng.resident_jobs() // 2                      # the pair kernel's resident set: what eoc_netlist_cost calls R
COUNTS = [int(x) for x in os.environ.get("COUNTS", "1,8,64,256,512,1024,2048,4096").split(",")]
rng = np.random.default_rng(9)

opt = c._optimized                                # what the facades run: the picked form through eoc_netlist_optimize
OPS = [
    ("add8", {"ripple as written": lambda: c.ripple_carry_adder(8, carry_in_zero=True), "mux-carry": lambda: c.mux_carry_adder(8),
              "xor3/maj": lambda: c.maj_adder(8), "prefix": lambda: c.prefix_adder(8),
              "prefix (optimized)": lambda: opt(c.prefix_adder(8))}, lambda A, B: A + B, 8),
    ("sub8", {"ripple": lambda: c.subtractor(8)[:5], "ripple (optimized)": lambda: opt(c.subtractor(8))[:5],
              "xor3/maj": lambda: c.maj_subtractor(8)[:5], "prefix": lambda: c.prefix_subtractor(8)[:5],
              "prefix (optimized)": lambda: opt(c.prefix_subtractor(8))[:5]},
     lambda A, B: (A - B) % 256, 8),
    ("lt8", {"ripple": lambda: c.less_than(8), "ripple (optimized)": lambda: opt(c.less_than(8)), "maj": lambda: c.maj_less_than(8),
             "tree": lambda: c.less_than_tree(8), "tree (optimized)": lambda: opt(c.less_than_tree(8))},
     lambda A, B: (A < B).astype(np.int64), 8),
    # min / max: built[4] = min wires (checked below), built[5] = max wires (checked through CHECK_MAX)
    ("minmax8", {"chain + 2 MUX as written": lambda: c.min_max(8),
                 "maj + 2 MUX": lambda: opt(c.min_max_on(c.maj_less_than(8))),
                 "maj + MUX + XOR3": lambda: opt(c.min_max_on(c.maj_less_than(8), True)),
                 "tree + 2 MUX": lambda: opt(c.min_max_on(c.less_than_tree(8))),
                 "tree + MUX + XOR3": lambda: opt(c.min_max_on(c.less_than_tree(8), True))}, lambda A, B: np.minimum(A, B), 8),
    ("mul8", {"rows as written": lambda: c.multiplier(8), "rows (optimized)": lambda: c.MULTIPLIER_FORMS["rows"](8),
              "columns (optimized)": lambda: c.MULTIPLIER_FORMS["wallace"](8)}, lambda A, B: A * B, 8),
]
# the forms the facades choose between (the others are shown for comparison)
CANDIDATES = {"add8": ("xor3/maj", "prefix (optimized)"), "sub8": ("xor3/maj", "prefix (optimized)"), "lt8": ("maj", "tree (optimized)"),
              "minmax8": ("maj + 2 MUX", "maj + MUX + XOR3", "tree + 2 MUX", "tree + MUX + XOR3"),
              "mul8": ("rows (optimized)", "columns (optimized)")}
worst = 0.0
agree = total = 0
for name, forms, truth, nbits in OPS:
    print(f"== {name}")
    for S in COUNTS:
        if name == "mul8" and S > 1024:
            continue
        A, B = rng.integers(0, 1 << nbits, S), rng.integers(0, 1 << nbits, S)
        row = {}
        for fname, build in forms.items():
            built = build()
            gates, n_wires, aw, bw, outw = built[0], built[1], built[2], built[3], built[4]
            outw = outw if isinstance(outw, list) else [outw]
            wires = torch.zeros((n_wires, S, p.n + 1), dtype=torch.int32, device="cuda")
            for i in range(nbits):
                wires[aw[i]] = torch.from_numpy(sk.encrypt_bits(((A >> i) & 1).astype(np.uint8), 100 + i, 0)).cuda()
                wires[bw[i]] = torch.from_numpy(sk.encrypt_bits(((B >> i) & 1).astype(np.uint8), 200 + i, 0)).cuda()
            ts = []
            for rep in range(6):
                t0 = time.perf_counter()
                eng.circuit_run_device(gates, wires.data_ptr(), n_wires, S)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            ms = float(np.median(ts[1:])) * 1e3
            got = sum(sk.decrypt_bits(wires[w].cpu().numpy()).astype(np.int64) << i for i, w in enumerate(outw))
            ok = bool(np.array_equal(got, truth(A, B)))
            if name == "minmax8":
                top = sum(sk.decrypt_bits(wires[w].cpu().numpy()).astype(np.int64) << i for i, w in enumerate(built[5]))
                ok = ok and bool(np.array_equal(top, np.maximum(A, B)))
            est = eoc.netlist_cost(gates, S, R) / 10.0
            row[fname] = (ms, est, ok, eoc.circuit_bootstraps(gates), eoc.netlist_levels(gates)[2])
        cand = CANDIDATES[name]
        picked = min(cand, key=lambda k: (row[k][1], row[k][3]))
        fastest = min(cand, key=lambda k: row[k][0])
        total += 1
        agree += picked == fastest or row[picked][0] <= 1.03 * row[fastest][0]
        cells = []
        for k, (ms, est, ok, boots, depth) in row.items():
            worst = max(worst, abs(ms / est - 1))
            cells.append(f"{k} [{boots}/{depth}] {ms:8.2f} ms (est {est:7.1f}){'*' if k == picked else ' '}{'<' if k == fastest else ' '}{'' if ok else ' WRONG'}")
        print(f"  S={S:5d}  " + " | ".join(cells), flush=True)
print(f"picked form (* : lowest estimate among the facades' candidates) = measured-fastest candidate (<, or within 3 % of it) in "
      f"{agree} of {total} cases; worst |measured / estimate - 1| over all forms = {worst:.3f}")
eng.close()
