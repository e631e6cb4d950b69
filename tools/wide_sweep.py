"""Pair kernel against the one-wave-per-ciphertext kernel over launch widths (device-pointer API, NAND; PSET=0 Set A, 1 Set B;
PARTS="1,2,3": also the forced wide kernel with the blind rotation cut into that many consecutive launches).
Usage (GPU box): python tools/wide_sweep.py [duty ...]   -- each duty is an EOC_TFHE_PRIO_DUTY value for the wide kernel
(default: the built-in alternation).  Prints ms per call and bootstraps/s; every result is decrypt-checked and the two
kernels' outputs are compared bit for bit."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eoc_tfhe_amd as eoc  # noqa: E402

p = eoc.default_params(int(os.environ.get("PSET", "0")))
sk = eoc.SecretKey(p, 1)
WIDTHS = [int(x) for x in os.environ.get("WIDTHS", "1024,1280,1536,2048,3072,4096,6144,8192,16384").split(",")]
G = max(WIDTHS)
bits = np.random.default_rng(0).integers(0, 2, G).astype(np.uint8)
c0 = torch.from_numpy(sk.encrypt_bits(bits, 2, 0)).cuda()
c1 = torch.from_numpy(sk.encrypt_bits(bits, 3, 0)).cuda()
ref = {}
modes = [("pair", {"EOC_TFHE_BR_WIDE": "0"}), ("auto (shipped policy)", {})]
duties = sys.argv[1:] or [""]
for d in duties:
    env = {"EOC_TFHE_BR_WIDE": "1"}
    if d:
        env["EOC_TFHE_PRIO_DUTY"] = d
    modes.append((f"wide duty={d or 'default'}", env))
for parts in [x for x in os.environ.get("PARTS", "").split(",") if x]:
    modes.append((f"wide parts={parts}", {"EOC_TFHE_BR_WIDE": "1", "EOC_TFHE_BR_PARTS": parts}))
    modes.append((f"pair parts={parts}", {"EOC_TFHE_BR_WIDE": "0", "EOC_TFHE_BR_PARTS": parts}))
for name, env in modes:
    for k in ("EOC_TFHE_BR_WIDE", "EOC_TFHE_PRIO_DUTY", "EOC_TFHE_BR_PARTS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    eng = eoc.Engine(p)
    eng.load_cloud_key(sk)
    eng.set_profiling(True)
    line = []
    for cnt in WIDTHS:
        out = torch.empty_like(c0[:cnt])
        for _ in range(6):
            eng.gate_batch_device(0, c0.data_ptr(), c1.data_ptr(), None, out.data_ptr(), cnt)
        torch.cuda.synchronize()
        eng.kernel_times(reset=True)
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.gate_batch_device(0, c0.data_ptr(), c1.data_ptr(), None, out.data_ptr(), cnt)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        kt = eng.kernel_times(reset=True)
        o = out.cpu().numpy()
        ok = np.array_equal(sk.decrypt_bits(o), 1 - bits[:cnt])
        if name == "pair":
            ref[cnt] = o
        same = np.array_equal(o, ref[cnt])
        line.append(f"{cnt}: {dt * 1e3:.3f} ms {cnt / dt / 1e3:.1f}k{'' if ok else ' WRONG'}{'' if same else ' DIFFERS'}")
    print(f"{name:24s} " + " | ".join(line), flush=True)
    eng.close()

# ---- mixed column (round 6): `cnt` rows of NAND / XOR / MUX in arbitrary order -- the two-input block and the MUX run
# as ONE pooled blind rotation (shipped) against one level per group (EOC_TFHE_NO_POOL=1); bootstraps = rows + MUX rows
if os.environ.get("MIXED", "1") == "1" and p.l == 2:
    c2 = torch.from_numpy(sk.encrypt_bits(bits, 4, 0)).cuda()
    mops_all = np.random.default_rng(4).choice(np.array([0, 4, 10], np.uint8), G)
    mref = {}
    for name, env in (("mixed, levels per group", {"EOC_TFHE_NO_POOL": "1"}), ("mixed, one pool (shipped)", {})):
        for k in ("EOC_TFHE_BR_WIDE", "EOC_TFHE_PRIO_DUTY", "EOC_TFHE_BR_PARTS", "EOC_TFHE_NO_POOL"):
            os.environ.pop(k, None)
        os.environ.update(env)
        eng = eoc.Engine(p)
        eng.load_cloud_key(sk)
        line = []
        for cnt in WIDTHS:
            mops = mops_all[:cnt]
            boots = cnt + int((mops == 10).sum())
            out = torch.empty_like(c0[:cnt])
            for _ in range(4):
                eng.gate_batch_device(0, c0.data_ptr(), c1.data_ptr(), c2.data_ptr(), out.data_ptr(), cnt, ops=mops)
            torch.cuda.synchronize()
            reps = 5
            t0 = time.perf_counter()
            for _ in range(reps):
                eng.gate_batch_device(0, c0.data_ptr(), c1.data_ptr(), c2.data_ptr(), out.data_ptr(), cnt, ops=mops)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            o = out.cpu().numpy()
            b = bits[:cnt]
            want = np.where(mops == 0, 1 - (b & b), np.where(mops == 4, b ^ b, b))       # all three operands carry `bits`
            ok = np.array_equal(sk.decrypt_bits(o), want)
            if cnt not in mref:
                mref[cnt] = o
            same = np.array_equal(o, mref[cnt])
            line.append(f"{cnt}: {dt * 1e3:.3f} ms {boots / dt / 1e3:.1f}k{'' if ok else ' WRONG'}{'' if same else ' DIFFERS'}")
        print(f"{name:24s} " + " | ".join(line), flush=True)
        eng.close()
    os.environ.pop("EOC_TFHE_NO_POOL", None)
