"""Diagnostic: per-step GPU time of the headline workload (1024 bootsNAND, Set A) over a long run that starts from an idle
device, after idle gaps of several lengths -- shows the device's clock ramp.  Usage: python tools/clock_ramp.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eoc_tfhe_amd as eoc  # noqa: E402

p = eoc.default_params(0)  # Set A
eng = eoc.Engine(p, device=0)
sk = eoc.SecretKey(p, 1)
eng.load_cloud_key(sk)
G = 1024
rng = np.random.default_rng(1)
dev = torch.device("cuda", 0)
d0 = torch.from_numpy(sk.encrypt_bits(rng.integers(0, 2, G).astype(np.uint8), 2, 0)).to(dev)
d1 = torch.from_numpy(sk.encrypt_bits(rng.integers(0, 2, G).astype(np.uint8), 3, 0)).to(dev)
out = torch.empty_like(d0)
st = torch.cuda.current_stream().cuda_stream


def run(nsteps):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(nsteps + 1)]
    ev[0].record()
    for i in range(nsteps):
        eng.gate_batch_device(eoc.OPS["NAND"], d0.data_ptr(), d1.data_ptr(), None, out.data_ptr(), G, stream=st)
        ev[i + 1].record()
    torch.cuda.synchronize()
    return [ev[i].elapsed_time(ev[i + 1]) for i in range(nsteps)]


for idle in (2.0, 0.001, 0.005, 0.02, 0.1):
    time.sleep(idle)
    t = run(400 if idle == 2.0 else 40)
    pick = [0, 1, 2, 4, 7, 10, 15, 20, 30, 39] + ([60, 100, 150, 200, 300, 399] if len(t) > 40 else [])
    print(f"idle {idle * 1e3:7.1f} ms -> step ms: " + " ".join(f"[{i}]{t[i]:.3f}" for i in pick), flush=True)
