import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import eoc_tfhe_amd as eoc
p = eoc.default_params(0); sk = eoc.SecretKey(p, 1)
rng = np.random.default_rng(0)
eng = eoc.Engine(p); eng.load_cloud_key(sk)
eoc.gpu_init(p, device=0); eoc.upload_cloud_key(sk)
for G in (1024, 4096, 16384):
    b0, b1 = rng.integers(0,2,G).astype(np.uint8), rng.integers(0,2,G).astype(np.uint8)
    c0, c1 = sk.encrypt_bits(b0, 2, 0), sk.encrypt_bits(b1, 3, 0)
    d0, d1 = torch.from_numpy(c0).cuda(), torch.from_numpy(c1).cuda(); o = torch.empty_like(d0)
    reps = 10 if G <= 4096 else 3
    eng.gate_batch_device(0, d0.data_ptr(), d1.data_ptr(), None, o.data_ptr(), G); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): eng.gate_batch_device(0, d0.data_ptr(), d1.data_ptr(), None, o.data_ptr(), G)
    torch.cuda.synchronize()
    t_res = (time.perf_counter() - t0) / reps
    pin = [eoc.PinnedArray(c0.shape) for _ in range(3)]
    pin[0].array[:] = c0; pin[1].array[:] = c1
    eoc.gate_batch(0, pin[0].array, pin[1].array, out=pin[2].array)
    t0 = time.perf_counter()
    for _ in range(reps): eoc.gate_batch(0, pin[0].array, pin[1].array, out=pin[2].array)
    t_pin = (time.perf_counter() - t0) / reps
    ok = np.array_equal(pin[2].array, o.cpu().numpy())
    hout = np.empty_like(c0)
    eoc.gate_batch(0, c0, c1, out=hout)
    t0 = time.perf_counter()
    for _ in range(reps): eoc.gate_batch(0, c0, c1, out=hout)
    t_pg = (time.perf_counter() - t0) / reps
    print(f"{G}: resident {t_res*1e3:.3f} ms ({G/t_res/1e3:.1f}k)  pinned {t_pin*1e3:.3f} ms ratio {t_res/t_pin:.3f}  pageable {t_pg*1e3:.3f} ms ratio {t_res/t_pg:.3f}  ok={ok and np.array_equal(hout, pin[2].array)}")
    for a in pin: a.free()
