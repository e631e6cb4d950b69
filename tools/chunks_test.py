import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import eoc_tfhe_amd as eoc
p = eoc.default_params(0); sk = eoc.SecretKey(p, 1)
G = 1024
rng = np.random.default_rng(0)
b0, b1 = rng.integers(0,2,G).astype(np.uint8), rng.integers(0,2,G).astype(np.uint8)
c0, c1 = sk.encrypt_bits(b0, 2, 0), sk.encrypt_bits(b1, 3, 0)
for chunks in ("1",):
    os.environ["EOC_TFHE_HOST_CHUNKS"] = chunks
    eoc.gpu_init(p, device=0); eoc.upload_cloud_key(sk)
    pin = [eoc.PinnedArray(c0.shape) for _ in range(3)]
    pin[0].array[:] = c0; pin[1].array[:] = c1
    eoc.gate_batch(0, pin[0].array, pin[1].array, out=pin[2].array)
    t0 = time.perf_counter()
    for _ in range(10): eoc.gate_batch(0, pin[0].array, pin[1].array, out=pin[2].array)
    dt = (time.perf_counter() - t0) / 10
    ok = np.array_equal(sk.decrypt_bits(pin[2].array), 1 - (b0 & b1))
    print(f"chunks={chunks}: {dt*1e3:.3f} ms/call  {G/dt:.0f} gates/s ok={ok}")
    for a in pin: a.free()
    eoc.gpu_shutdown()
# two engines on two torch streams, device API
e1, e2 = eoc.Engine(p), eoc.Engine(p)
e1.load_cloud_key(sk); e2.load_cloud_key(sk)
d0, d1 = torch.from_numpy(c0).cuda(), torch.from_numpy(c1).cuda()
o = torch.empty_like(d0)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def both():
    e1.gate_batch_device(0, d0.data_ptr(), d1.data_ptr(), None, o.data_ptr(), 512, stream=s1.cuda_stream)
    e2.gate_batch_device(0, d0[512:].data_ptr(), d1[512:].data_ptr(), None, o[512:].data_ptr(), 512, stream=s2.cuda_stream)
both(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): both()
torch.cuda.synchronize()
print(f"two engines x 512 on two streams: {(time.perf_counter()-t0)/10*1e3:.3f} ms")
def one():
    e1.gate_batch_device(0, d0.data_ptr(), d1.data_ptr(), None, o.data_ptr(), 1024, stream=s1.cuda_stream)
one(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): one()
torch.cuda.synchronize()
print(f"one engine x 1024: {(time.perf_counter()-t0)/10*1e3:.3f} ms")
def half():
    e1.gate_batch_device(0, d0.data_ptr(), d1.data_ptr(), None, o.data_ptr(), 512, stream=s1.cuda_stream)
half(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): half()
torch.cuda.synchronize()
print(f"one engine x 512: {(time.perf_counter()-t0)/10*1e3:.3f} ms")
# multi-round launches: priority duty forced on / off
from eoc_tfhe_amd import circuits
for duty in ("-1", "11", "8"):
    os.environ["EOC_TFHE_PRIO_DUTY"] = duty
    e3 = eoc.Engine(p); e3.load_cloud_key(sk)
    for G2 in (2048, 4096, 16384):
        bb0 = rng.integers(0,2,G2).astype(np.uint8)
        x0 = torch.from_numpy(sk.encrypt_bits(bb0, 5, 0)).cuda(); x1 = torch.from_numpy(sk.encrypt_bits(bb0, 6, 0)).cuda()
        oo = torch.empty_like(x0)
        e3.gate_batch_device(0, x0.data_ptr(), x1.data_ptr(), None, oo.data_ptr(), G2); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): e3.gate_batch_device(0, x0.data_ptr(), x1.data_ptr(), None, oo.data_ptr(), G2)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print(f"duty {duty}: {G2} gates {dt*1e3:.2f} ms  {G2/dt:.0f} gates/s")
    e3.close()
