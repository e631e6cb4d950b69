"""Device-buffer plumbing for the GPU tests: torch owns the HBM allocations, the C ABI gets raw pointers."""
import numpy as np


def torch_cuda():
    import torch  # imported before libeoc_tfhe_gpu.so so both share one HIP runtime
    assert torch.cuda.is_available(), "GPU test on a box without a GPU"
    return torch


def to_dev(a):
    torch = torch_cuda()
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def dev_empty(shape, dtype):
    torch = torch_cuda()
    return torch.empty(shape, dtype=dtype, device="cuda")


def sync():
    torch_cuda().cuda.synchronize()
