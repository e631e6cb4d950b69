import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import eoc_tfhe_amd as eoc
p = eoc.default_params(0); sk = eoc.SecretKey(p, 1)
eoc.gpu_init(p, devices=[0]); eoc.upload_cloud_key(sk)
G = 1024
rng = np.random.default_rng(0)
b0 = rng.integers(0, 2, G).astype(np.uint8); b1 = rng.integers(0, 2, G).astype(np.uint8)
c0 = sk.encrypt_bits(b0, 2, 0); c1 = sk.encrypt_bits(b1, 3, 0)
pin = [eoc.PinnedArray(c0.shape) for _ in range(4)]
pin[0].array[:] = c0; pin[1].array[:] = c1
eng = eoc.Engine.borrow_global(0)
for mode in ("sync", "pipelined"):
    for rep in range(2):
        eng.set_profiling(True); eng.kernel_times(reset=True)
        t0 = time.perf_counter()
        if mode == "sync":
            for k in range(30): eoc.gate_batch(0, pin[0].array, pin[1].array, out=pin[2].array)
        else:
            tk = []
            for k in range(30):
                if k >= 2: eoc.gate_batch_wait(tk[k - 2])
                tk.append(eoc.gate_batch_submit(0, pin[0].array, pin[1].array, out=pin[2 + (k & 1)].array))
            for t in tk[-2:]: eoc.gate_batch_wait(t)
        dt = (time.perf_counter() - t0) / 30
        kt = eng.kernel_times(reset=True); eng.set_profiling(False)
        print(mode, rep, f"{dt*1e3:.3f} ms/call", {k: round(v["ms"] / max(1, v["launches"]), 4) for k, v in kt.items()}, flush=True)
ok = np.array_equal(sk.decrypt_bits(pin[2].array), 1 - (b0 & b1))
print("ok", ok)
