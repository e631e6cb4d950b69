#!/usr/bin/env python3
"""The in-library multi-GPU path on every visible device: ONE process, ONE global key (the reference's shape,
eoc-tfhe-run.cpp:38-40), one engine per GPU behind eoc_gate_batch; keys replicated by the library's own RCCL broadcast.
Prints one JSON line.  bench.py runs it as a child process at N = 1 when more than one GPU is visible (on a one-GPU box it
has nothing to add); usable on its own:  python tools/in_library_multi.py [--devices 0,1] [--steps 10] [--pset A]"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--devices", default="all")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--gates-per-device", type=int, default=1024)
    ap.add_argument("--pset", default="A", choices=["A", "B"])
    args = ap.parse_args()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import eoc_tfhe_amd as eoc
    ndev = eoc.lib().eoc_device_count()
    devices = list(range(ndev)) if args.devices == "all" else [int(x) for x in args.devices.split(",")]
    p = eoc.default_params({"A": 0, "B": 1}[args.pset])
    sk = eoc.SecretKey(p, 1)
    eoc.gpu_init(p, devices=devices)
    t0 = time.perf_counter()
    eoc.upload_cloud_key(sk)
    t_upload = time.perf_counter() - t0
    st = eoc.stats_multi()
    G = args.gates_per_device * len(devices)
    rng = np.random.default_rng(77)
    b0, b1 = rng.integers(0, 2, G).astype(np.uint8), rng.integers(0, 2, G).astype(np.uint8)
    pin = [eoc.PinnedArray((G, p.n + 1)) for _ in range(3)]
    pin[0].array[:] = sk.encrypt_bits(b0, 2, 0)
    pin[1].array[:] = sk.encrypt_bits(b1, 3, 0)
    for _ in range(6):
        eoc.gate_batch(eoc.OPS["NAND"], pin[0].array, pin[1].array, out=pin[2].array)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eoc.gate_batch(eoc.OPS["NAND"], pin[0].array, pin[1].array, out=pin[2].array)
    dt = time.perf_counter() - t0
    ok = bool(np.array_equal(sk.decrypt_bits(pin[2].array), 1 - (b0 & b1)))
    after = eoc.stats_multi()
    res = {"devices": devices, "engines": len(devices), "gates_per_call": G, "calls": args.steps,
           "gates_per_s": round(G * args.steps / dt, 1), "ms_per_call": round(dt / args.steps * 1e3, 4),
           "timed_region": "synchronous eoc_gate_batch calls on pinned buffers, first H2D to last D2H, blocks of 1024 gates per engine",
           "key_broadcast_method": st["key_broadcast_method"], "key_broadcast_s": round(st["key_broadcast_s"], 4),
           "key_upload_total_s": round(t_upload, 3), "rccl_origin": st["rccl_origin"],
           "bootstraps_per_engine": [e["bootstraps"] for e in after["engines"]],
           "worker_wakeups": after["worker_wakeups"], "decrypt_ok": ok, "param_set": args.pset}
    for a in pin:
        a.free()
    eoc.gpu_shutdown()
    os.write(real_stdout, (json.dumps(res) + "\n").encode())
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
