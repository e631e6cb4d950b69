"""Latency of small batches on the device-pointer API (Set A): milliseconds per call and per-gate rate for 1 ... 2048 gates.
A blind rotation is n = 500 strictly sequential steps, so a call never takes less than one ciphertext's 500 steps;
throughput comes from width.  Usage (GPU box): python tools/latency_probe.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eoc_tfhe_amd as eoc  # noqa: E402

p = eoc.default_params(0)
sk = eoc.SecretKey(p, 1)
eng = eoc.Engine(p)
eng.load_cloud_key(sk)
G = 2048
bits = np.random.default_rng(0).integers(0, 2, G).astype(np.uint8)
c0 = torch.from_numpy(sk.encrypt_bits(bits, 2, 0)).cuda()
c1 = torch.from_numpy(sk.encrypt_bits(bits, 3, 0)).cuda()
out = torch.empty_like(c0)
for cnt in (1, 2, 8, 32, 128, 256, 512, 768, 1024, 2048):
    for _ in range(12):
        eng.gate_batch_device(0, c0.data_ptr(), c1.data_ptr(), None, out.data_ptr(), cnt)
    torch.cuda.synchronize()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.gate_batch_device(0, c0.data_ptr(), c1.data_ptr(), None, out.data_ptr(), cnt)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{cnt:5d} gates: {dt * 1e3:7.3f} ms per call  {cnt / dt:10.0f} gates/s  {dt / 500 * 1e6:6.2f} us per blind-rotate step")
