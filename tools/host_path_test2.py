import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import eoc_tfhe_amd as eoc
p = eoc.default_params(0); sk = eoc.SecretKey(p, 1)
rng = np.random.default_rng(0)
G = 1024
b0, b1 = rng.integers(0,2,G).astype(np.uint8), rng.integers(0,2,G).astype(np.uint8)
c0, c1 = sk.encrypt_bits(b0, 2, 0), sk.encrypt_bits(b1, 3, 0)
eng = eoc.Engine(p); eng.load_cloud_key(sk)
d0, d1 = torch.from_numpy(c0).cuda(), torch.from_numpy(c1).cuda(); o = torch.empty_like(d0)
def resident(reps):
    eng.gate_batch_device(0, d0.data_ptr(), d1.data_ptr(), None, o.data_ptr(), G); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): eng.gate_batch_device(0, d0.data_ptr(), d1.data_ptr(), None, o.data_ptr(), G)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
def resident_sync(reps):
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.gate_batch_device(0, d0.data_ptr(), d1.data_ptr(), None, o.data_ptr(), G); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
print(f"resident, queued: {resident(20)*1e3:.3f} ms   resident, synchronised every call: {resident_sync(20)*1e3:.3f} ms")
eoc.gpu_init(p, device=0); eoc.upload_cloud_key(sk)
pin = [eoc.PinnedArray(c0.shape) for _ in range(3)]
pin[0].array[:] = c0; pin[1].array[:] = c1
hout = np.empty_like(c0)
L = eoc.lib()
def raw(a, b, out, reps):
    L.eoc_gate_batch(0, None, a.ctypes.data, b.ctypes.data, None, out.ctypes.data, G)
    t0 = time.perf_counter()
    for _ in range(reps): L.eoc_gate_batch(0, None, a.ctypes.data, b.ctypes.data, None, out.ctypes.data, G)
    return (time.perf_counter() - t0) / reps
for rnd in range(3):
    os.environ.pop("EOC_TFHE_NO_ZERO_COPY", None)
    tz = raw(pin[0].array, pin[1].array, pin[2].array, 20)
    os.environ["EOC_TFHE_NO_ZERO_COPY"] = "1"
    td = raw(pin[0].array, pin[1].array, pin[2].array, 20)
    tp = raw(c0, c1, hout, 20)
    print(f"round {rnd}: pinned zero-copy {tz*1e3:.3f} ms   pinned DMA copies {td*1e3:.3f} ms   pageable {tp*1e3:.3f} ms")
