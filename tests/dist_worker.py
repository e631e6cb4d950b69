"""Worker for tests/test_distributed_cpu.py: world_size-2 rehearsal of the multi-GPU path on CPU.

Same plumbing as bench.py --gpus N (eoc_tfhe_amd.distributed): rank 0 makes the cloud key, the key
images are broadcast, every rank evaluates its contiguous shard of independent gates, results are
gathered.  Without a GPU the shard is evaluated by the oracle (the checker stands in for the
engine here -- this test is about sharding and the collective, not about the kernels)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import ctypes as C  # noqa: E402

import eoc_tfhe_amd as eoc  # noqa: E402
import oracle_lib as ol  # noqa: E402
from eoc_tfhe_amd import distributed as D  # noqa: E402


def main():
    out_path = sys.argv[1]
    total = int(sys.argv[2])
    rank, world, _, dist = D.init(backend="gloo")
    p = eoc.default_params(0)
    p.n = 16
    sk = eoc.SecretKey(p, 5, with_cloud_key=(rank == 0))
    o = ol.Oracle(0, 5, n_override=16, with_bk=False)
    bk = torch.zeros((p.n, 2 * p.l, 2, 1024), dtype=torch.int32)
    ksk = torch.zeros((1024 * p.ks_t * 3, p.n + 1), dtype=torch.int32)
    if rank == 0:
        bk.copy_(torch.from_numpy(sk.bk.copy()))
        ksk.copy_(torch.from_numpy(sk.ksk.copy()))
    secs = D.broadcast_key_images(dist, [bk, ksk], src=0)
    o.bk, o.ksk = bk.numpy(), ksk.numpy()
    o.bkfft = np.zeros((p.n, 2 * p.l, 2, 1024), np.float64)
    o.L.orc_bk_to_fft(C.byref(o.p), o.bk, o.bkfft)
    # same inputs on every rank (seeded), each evaluates only its block
    bits0 = np.random.default_rng(1).integers(0, 2, total)
    bits1 = np.random.default_rng(2).integers(0, 2, total)
    c0, c1 = o.encrypt_bits(bits0, 2, 0), o.encrypt_bits(bits1, 3, 0)
    lo, hi = D.shard(total, rank, world)
    local = o.gate_batch(ol.OPS["NAND"], c0[lo:hi], c1[lo:hi]) if hi > lo else np.zeros((0, p.n + 1), np.int32)
    full = D.gather_blocks(dist, torch.from_numpy(local), total, rank, world).numpy()
    if rank == 0:
        ref = o.gate_batch(ol.OPS["NAND"], c0, c1)
        ok = bool(np.array_equal(full, ref)) and bool(np.array_equal(o.decrypt_bits(full), 1 - (bits0 & bits1)))
        json.dump({"ok": ok, "world": world, "blocks": [D.shard(total, r, world) for r in range(world)],
                   "broadcast_s": secs}, open(out_path, "w"))
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
