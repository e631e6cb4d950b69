"""Worker for tests/test_gpu_distributed.py: world_size-2 run of the N > 1 path on the HIP ENGINE.  Two ranks (gloo
control plane) share GPU 0; rank 0 builds the key images, eoc_tfhe_amd.distributed.replicate_cloud_key broadcasts them,
every rank evaluates its contiguous shard on its own engine, results are gathered and rank 0 compares with the oracle."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import eoc_tfhe_amd as eoc  # noqa: E402
from eoc_tfhe_amd import distributed as D  # noqa: E402


def main():
    out_path, total = sys.argv[1], int(sys.argv[2])
    rank, world, _, dist = D.init(backend="gloo")
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    p = eoc.default_params(0)
    p.n = 40
    sk = eoc.SecretKey(p, 9, with_cloud_key=(rank == 0))
    eng = eoc.Engine(p, device=0)
    # gloo moves host tensors: the images travel through the host here (RCCL moves them device to device in bench.py)
    bkfft = torch.empty(eng.bkfft_bytes // 8, dtype=torch.float64, device=dev)
    ksk = torch.empty(eng.ksk_dev_bytes // 4, dtype=torch.int32, device=dev)
    if rank == 0:
        eng.build_cloud_key_device(sk, bkfft.data_ptr(), ksk.data_ptr())
    hb, hk = bkfft.cpu(), ksk.cpu()
    secs = D.broadcast_key_images(dist, [hb, hk], src=0)
    if rank != 0:
        bkfft.copy_(hb)
        ksk.copy_(hk)
        eng.set_cloud_key_device(bkfft.data_ptr(), ksk.data_ptr())
    bits0 = np.random.default_rng(1).integers(0, 2, total).astype(np.uint8)
    bits1 = np.random.default_rng(2).integers(0, 2, total).astype(np.uint8)
    c0, c1 = sk.encrypt_bits(bits0, 2, 0), sk.encrypt_bits(bits1, 3, 0)
    lo, hi = D.shard(total, rank, world)
    d0, d1 = torch.from_numpy(c0[lo:hi]).to(dev), torch.from_numpy(c1[lo:hi]).to(dev)
    dout = torch.empty_like(d0)
    if hi > lo:
        eng.gate_batch_device(eoc.OPS["NAND"], d0.data_ptr(), d1.data_ptr(), None, dout.data_ptr(), hi - lo)
    torch.cuda.synchronize()
    full = D.gather_blocks(dist, dout.cpu(), total, rank, world).numpy()
    if rank == 0:
        import oracle_lib as ol
        orc = ol.Oracle(0, 9, n_override=40)
        ref = orc.gate_batch(ol.OPS["NAND"], c0, c1)
        ok = bool(np.array_equal(full, ref)) and bool(np.array_equal(sk.decrypt_bits(full), 1 - (bits0 & bits1)))
        json.dump({"ok": ok, "world": world, "blocks": [D.shard(total, r, world) for r in range(world)],
                   "broadcast_s": secs, "bootstraps_rank0": eng.stats()["bootstraps"]}, open(out_path, "w"))
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
