#!/usr/bin/env python3
"""Cycle shares of the blind-rotate step's segments from the -DEOC_STAMPS diagnostic build.
Usage (GPU box):  python -m eoc_tfhe_amd.build --stamps && EOC_TFHE_LIB=eoc_tfhe_amd/libeoc_tfhe_gpu_stamps.so python tools/stamps.py [A|B]
Read SHARES, not lengths: the stamps' waits forbid overlaps the real kernel has."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
import eoc_tfhe_amd as eoc  # noqa

pset = {"A": 0, "B": 1}[sys.argv[1] if len(sys.argv) > 1 else "A"]
G = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
p = eoc.default_params(pset)
sk = eoc.SecretKey(p, 1)
eng = eoc.Engine(p)
eng.load_cloud_key(sk)
L = eoc.lib()
L.eoc_dbg_stamps.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
WIDE = os.environ.get("EOC_TFHE_BR_WIDE") == "1" or G > 1024     # one wave per ciphertext (k_blind_rotate_wide)
waves = G if WIDE else G * 2
bits = np.random.default_rng(0).integers(0, 2, G)
c0 = torch.from_numpy(sk.encrypt_bits(bits, 2, 0)).cuda()
c1 = torch.from_numpy(sk.encrypt_bits(bits, 3, 0)).cuda()
out = torch.empty_like(c0)
for _ in range(2):
    eng.gate_batch_device(0, c0.data_ptr(), c1.data_ptr(), None, out.data_ptr(), G)
torch.cuda.synchronize()
assert L.eoc_dbg_stamps(eng.h, waves, None) == 0
eng.gate_batch_device(0, c0.data_ptr(), c1.data_ptr(), None, out.data_ptr(), G)
buf = np.zeros((waves, 16), np.uint64)
assert L.eoc_dbg_stamps(eng.h, waves, buf.ctypes.data) == 0
wide_names = {0: "rotation + digit words, both polynomials", 1: "forward pair, polynomial 0", 2: "forward pair, polynomial 1",
              3: "chains (key rows streamed by bin block)", 8: "inverse pair", 9: "untwist + round + acc update",
              15: "loop top (bara load)"}
names = {0: "rotate-diff (acc reads)", 1: "BK loads issue + digits + twist", 2: "forward FFT", 3: "MAC (waits BK)",
         4: "xchg write", 5: "barrier A", 6: "xchg read + add", 7: "barrier B", 8: "inverse FFT",
         9: "untwist + round + acc update", 15: "loop top (bara load)"}
if WIDE:
    names = wide_names
    if pset == 1:
        names = {0: "rotation + digit words + parking", 1: "forward pair (0,1) (0,2)", 2: "forward pair (0,3) (1,1)",
                 3: "chain phase A (5 rows)", 4: "single forward (1,2)", 5: "chain phase B (2 rows)", 6: "single forward (1,3)",
                 7: "chain phase C (5 rows)", 8: "fetch + inverse pair", 9: "untwist + round + acc update", 15: "loop top (bara load)"}
meta = buf[:, 11:15].copy()
buf[:, 11:15] = 0
tot = buf.sum(axis=1).astype(np.float64)
print(f"waves={waves} steps={p.n}  mean cycles/step/wave = {tot.mean() / p.n:.0f}")
for k in sorted(names):
    v = buf[:, k].astype(np.float64)
    print(f"  [{k:2d}] {names[k]:34s} {v.mean() / p.n:9.1f} cyc/step  {100 * v.sum() / tot.sum():5.1f} %")

# per-workgroup loop durations: a launch ends with its slowest workgroup
dur = (meta[:, 0].astype(np.int64) - meta[:, 1].astype(np.int64)).astype(np.float64)   # exit - entry
t0 = meta[:, 1].astype(np.int64)
wg = dur.reshape(-1, 2).max(axis=1)        # per workgroup (two waves: a pair, or two independent ciphertexts)
start = (t0.reshape(-1, 2).min(axis=1) - t0.min()).astype(np.float64)
end = start + wg
print(f"workgroup loop duration [cycles]: min {wg.min():.0f}  mean {wg.mean():.0f}  max {wg.max():.0f}  (max/mean = {wg.max() / wg.mean():.3f})")
print(f"loop entry spread: {start.max():.0f} cycles;  last exit at {end.max():.0f};  mean exit at {end.mean():.0f}")
xcc = (meta[:, 2] & 0xF).reshape(-1, 2)[:, 0]
hw = meta[:, 3].reshape(-1, 2)[:, 0]
cu = (hw >> 8) & 0xF
sh = (hw >> 12) & 0x1
se = (hw >> 13) & 0x7
for x in sorted(set(xcc.tolist())):
    m = xcc == x
    print(f"  XCC {x}: {m.sum():4d} workgroups  mean {wg[m].mean():.0f}  max {wg[m].max():.0f}  distinct (se,sh,cu) = {len(set(zip(se[m].tolist(), sh[m].tolist(), cu[m].tolist())))}")
key = list(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
from collections import Counter
cnt = Counter(key)
print("workgroups per CU histogram:", sorted(Counter(cnt.values()).items()))
pairs = {}
for k, d in zip(key, wg.tolist()):
    pairs.setdefault(k, []).append(d)
two = np.array([sorted(v) for v in pairs.values() if len(v) == 2])
if len(two):
    print(f"per CU with two workgroups: faster mean {two[:, 0].mean():.0f}  slower mean {two[:, 1].mean():.0f}  "
          f"ratio {(two[:, 1] / two[:, 0]).mean():.3f}")
idx = np.arange(len(wg))
print(f"first half of the grid mean {wg[idx < len(wg) // 2].mean():.0f}   second half mean {wg[idx >= len(wg) // 2].mean():.0f}")
print("deciles:", " ".join(f"{q:.0f}" for q in np.percentile(wg, [0, 10, 25, 50, 75, 90, 100])))
