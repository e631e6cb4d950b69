"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL on ROCm).

The path shards naturally: every gate / circuit instance is independent given the
cloud key, so ranks take contiguous blocks of instances and the only collective is the one-time
broadcast of the two key images (BK-FFT, KSK) from rank 0 over xGMI.  No steady-state collective.
The same functions run under the "gloo" backend on CPU tensors (tests/test_distributed_cpu.py).
"""
import os


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment.  Returns (rank, world, local_rank, dist)."""
    rank, world, local_rank = env_rank()
    if world == 1:
        return rank, world, local_rank, None
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for RCCL (no effect once HIP is up)
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend, device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend)
    return rank, world, local_rank, dist


def shard(total, rank, world):
    """Contiguous block [lo, hi) of `total` independent instances owned by `rank`; blocks differ by
    at most one instance and cover [0, total) exactly once."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


# ---- BASELINE.json configs[3]: 2^20 mixed gates, op ~ uniform{NAND, XOR, MUX} from seed 4, contiguous blocks -----------
CONFIG3_OPCODES = (0, 4, 10)  # NAND, XOR, MUX (include/eoc_tfhe_gpu.h)


def config3_ops(total=1 << 20):
    """the op stream of SURVEY.md 8(d) config 4 / BASELINE configs[3]: one opcode per gate, seed 4"""
    import numpy as np
    return np.random.default_rng(4).choice(np.array(CONFIG3_OPCODES, np.uint8), total)


def config3_block(total, rank, world):
    """(lo, hi, opcodes of this rank's block, bootstraps of the WHOLE stream): what bench.py times at N > 1 and
    tests/test_gpu_baseline_configs.py evaluates shard by shard.  MUX = 2 bootstraps."""
    import numpy as np
    ops = config3_ops(total)
    lo, hi = shard(total, rank, world)
    return lo, hi, np.ascontiguousarray(ops[lo:hi]), int(total + (ops == 10).sum())


def broadcast_key_images(dist, tensors, src=0):
    """One broadcast per key image (BK-FFT, KSK) from `src`; returns seconds spent."""
    import time
    import torch
    if dist is None:
        return 0.0
    if tensors and tensors[0].is_cuda:
        torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for t in tensors:
        dist.broadcast(t, src=src)
    if tensors and tensors[0].is_cuda:
        torch.cuda.synchronize()
    return time.perf_counter() - t0


def gather_blocks(dist, local, total, rank, world):
    """All-gather ragged contiguous blocks (rows) back into one [total, ...] tensor (used by tests and
    by hosts that want the whole result on every rank; the benchmark never needs it)."""
    import torch
    if dist is None:
        return local
    sizes = [shard(total, r, world) for r in range(world)]
    maxlen = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((maxlen,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    outs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad)
    return torch.cat([outs[r][: hi - lo] for r, (lo, hi) in enumerate(sizes)], dim=0)


def replicate_cloud_key(eng, sk, dist, rank, device):
    """Multi-GPU key set-up: rank 0 builds the two device images (BK transformed on its GPU,
    KSK padded) into torch tensors, one RCCL broadcast per image sends them over xGMI, every other rank's
    engine adopts its copy.  Keys are replicated, never sharded: every gate needs all of both.
    Returns (bkfft_tensor, ksk_tensor, broadcast_seconds); the tensors own the memory and must stay alive."""
    import torch
    bkfft = torch.empty(eng.bkfft_bytes // 8, dtype=torch.float64, device=device)
    ksk = torch.empty(eng.ksk_dev_bytes // 4, dtype=torch.int32, device=device)
    if rank == 0:
        eng.build_cloud_key_device(sk, bkfft.data_ptr(), ksk.data_ptr())
    secs = broadcast_key_images(dist, [bkfft, ksk], src=0)
    if rank != 0:
        eng.set_cloud_key_device(bkfft.data_ptr(), ksk.data_ptr())
    return bkfft, ksk, secs
