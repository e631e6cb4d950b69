import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import eoc_tfhe_amd as eoc
p = eoc.default_params(0); sk = eoc.SecretKey(p, 1)
rng = np.random.default_rng(0)
for mode in sys.argv[1:]:
    os.environ["EOC_TFHE_BR_SLICE"] = mode
    e3 = eoc.Engine(p); e3.load_cloud_key(sk)
    line = []
    for G2 in (1536, 2048, 4096, 16384):
        bb0 = rng.integers(0,2,G2).astype(np.uint8)
        x0 = torch.from_numpy(sk.encrypt_bits(bb0, 5, 0)).cuda(); x1 = torch.from_numpy(sk.encrypt_bits(bb0, 6, 0)).cuda()
        oo = torch.empty_like(x0)
        e3.gate_batch_device(0, x0.data_ptr(), x1.data_ptr(), None, oo.data_ptr(), G2); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): e3.gate_batch_device(0, x0.data_ptr(), x1.data_ptr(), None, oo.data_ptr(), G2)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        ok = np.array_equal(sk.decrypt_bits(oo.cpu().numpy()), 1 - bb0)
        line.append(f"{G2}: {G2/dt/1e3:.1f}k{'' if ok else ' WRONG'}")
    print(f"br_slice {mode}: " + "  ".join(line))
    e3.close()
