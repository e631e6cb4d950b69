"""The HIP path against the COMMITTED golden bytes (tests/golden/golden_arrays.npz, golden.json): this module never
loads liboracle.so.  The other GPU tests compare with the live oracle, the CPU tests pin the oracle to the fixtures;
this closes the triangle, so that GPU and oracle cannot drift together unnoticed.

Fixtures are the oracle's own outputs (parity unpinned): key seed 1, the four input combinations of
(c0, c1, c2), Set A and Set B at their full sizes."""
import hashlib
import json
import os

import numpy as np
import pytest

from gpu_util import dev_empty, sync, to_dev, torch_cuda

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
N = 1024


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def eoc(built_lib):
    torch_cuda()
    import eoc_tfhe_amd
    return eoc_tfhe_amd


@pytest.fixture(scope="module")
def golden():
    g = json.load(open(os.path.join(HERE, "golden", "golden.json")))
    z = np.load(os.path.join(HERE, "golden", "golden_arrays.npz"))    # allow_pickle=False (default)
    return g, z


@pytest.mark.parametrize("pset,name", [(0, "A"), (1, "B")])
def test_gates_equal_committed_bytes(eoc, golden, pset, name):
    g, z = golden
    torch = torch_cuda()
    p = eoc.default_params(pset)
    sk = eoc.SecretKey(p, g[name]["key_seed"])
    assert sha(sk.bk) == g[name]["bk_sha"] and sha(sk.ksk) == g[name]["ksk_sha"]     # product keygen == fixture
    eng = eoc.Engine(p)
    eng.load_cloud_key(sk)
    c = [to_dev(z[f"{name}_c{k}"]) for k in range(3)]
    out = torch.empty_like(c[0])

    def run(op, a, b, m):
        eng.gate_batch_device(eoc.OPS[op], a.data_ptr(), None if b is None else b.data_ptr(),
                              None if m is None else m.data_ptr(), out.data_ptr(), a.shape[0])
        sync()
        return out.cpu().numpy()

    assert np.array_equal(run("NAND", c[0], c[1], None), z[f"{name}_NAND_out"])
    assert np.array_equal(run("MUX", c[0], c[1], c[2]), z[f"{name}_MUX_out"])
    assert np.array_equal(run("MAJ", c[0], c[1], c[2]), z[f"{name}_MAJ_out"])     # the extension gates (round 6)
    assert np.array_equal(run("XOR3", c[0], c[1], c[2]), z[f"{name}_XOR3_out"])
    for opn, e in g[name]["gates"].items():       # every opcode: SHA-256 of the four output samples + decrypted bits
        got = run(opn, c[0], None if opn == "NOT" else c[1], c[2] if opn in ("MUX", "MAJ", "XOR3") else None)
        assert sha(got) == e["sha"], opn
        assert sk.decrypt_bits(got).tolist() == e["bits"], opn
    # the blind rotation alone: extracted sample of NAND(1, 1), all 1025 words
    t = (np.int64(0) - z[f"{name}_c0"][3].astype(np.int64) - z[f"{name}_c1"][3].astype(np.int64))
    t[-1] += 1 << 29
    t = (t & 0xFFFFFFFF).astype(np.uint32).view(np.int32)[None, :]
    assert sha(t[0]) == g[name]["nand_11_t_sha"]
    d_u = dev_empty((1, N + 1), torch.int32)
    eng.blind_rotate_device(to_dev(t).data_ptr(), d_u.data_ptr(), 1)
    sync()
    assert np.array_equal(d_u.cpu().numpy()[0], z[f"{name}_nand_11_u"])
    eng.close()


def test_transforms_equal_committed_bytes(eoc, golden):
    g, z = golden
    torch = torch_cuda()
    p = eoc.default_params(0)
    p.n = 8
    eng = eoc.Engine(p)
    polys = np.stack([z["fft_small_in"], z["fft_big_in"]])
    d_s = dev_empty((2, N), torch.float64)
    eng.fft_fwd_device(to_dev(polys).data_ptr(), d_s.data_ptr(), 2)
    sync()
    spec = d_s.cpu().numpy()
    assert np.array_equal(spec[0], z["fft_small_spec"])
    assert sha(spec[0]) == g["fft"]["small_spec_sha"] and sha(spec[1]) == g["fft"]["big_spec_sha"]
    prod = (spec[0].view(np.complex128) * spec[1].view(np.complex128)).view(np.float64)[None, :]
    d_o = dev_empty((1, N), torch.int32)
    eng.fft_inv_device(to_dev(prod).data_ptr(), d_o.data_ptr(), 1)
    sync()
    assert np.array_equal(d_o.cpu().numpy()[0], z["fft_prod_inv"])
    eng.close()
