#!/usr/bin/env python3
"""Generate tests/golden/*.json|npz from the CPU oracle (run in the build container):

    python tests/golden/make_golden.py

PARITY UNPINNED: the reference holds no golden vector for gate bootstrapping and upstream libtfhe
is absent, so these vectors pin THIS repo's oracle (canonical transform v3, PRNG v1)
against regressions and pin the GPU path to it; they are not outputs of the reference.
Fixtures are data only: seeds, inputs, expected outputs, SHA-256 of the large arrays.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as ol  # noqa: E402

N = 1024


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    L = ol.lib()
    g = {"version": {"transform": "canonical-v3", "prng": "splitmix-ctr-v1"}}
    # PRNG / scalar KATs
    g["prng"] = {
        "stream_key(1,3,5)": str(L.orc_stream_key(1, 3, 5)),
        "u64(key(1,3,5),0..3)": [str(L.orc_rng_u64(L.orc_stream_key(1, 3, 5), i)) for i in range(4)],
        "gaussian32(key(9,5,0),ctr=10,mu=0,sigma=2.44e-5)x4":
            [int(L.orc_gaussian32(L.orc_stream_key(9, 5, 0), 10 + 2 * i, 0, 2.44e-5)) for i in range(4)],
    }
    g["modswitch"] = {
        "to(1,8)": L.orc_modswitch_to_torus32(1, 8), "to(-1,8)": L.orc_modswitch_to_torus32(-1, 8),
        "to(1,4)": L.orc_modswitch_to_torus32(1, 4),
        "to(42,2^31-1)": L.orc_modswitch_to_torus32(42, 2**31 - 1),
        "from(to(42,M),M)": L.orc_modswitch_from_torus32(L.orc_modswitch_to_torus32(42, 2**31 - 1), 2**31 - 1),
        "from(-1,2048)": L.orc_modswitch_from_torus32(-1, 2048),
        "from(2^31-2^20,2048)": L.orc_modswitch_from_torus32(2**31 - 2**20, 2048),
        "from(2^20-1,2048)": L.orc_modswitch_from_torus32(2**20 - 1, 2048),
        "from(2^20,2048)": L.orc_modswitch_from_torus32(2**20, 2048),
    }
    # transform KATs
    rng = np.random.default_rng(12345)
    small = rng.integers(-512, 512, N).astype(np.int32)
    big = rng.integers(-2**31, 2**31, N).astype(np.int32)
    fs, fb = ol.fft_fwd(small), ol.fft_fwd(big)
    prod = (fs.view(np.complex128) * fb.view(np.complex128)).view(np.float64)
    inv = ol.fft_inv(prod)
    arrays = {"fft_small_in": small, "fft_big_in": big, "fft_small_spec": fs, "fft_prod_inv": inv}
    g["fft"] = {"small_spec_sha": sha(fs), "big_spec_sha": sha(fb), "prod_inv_sha": sha(inv),
                "delta_spec_first4": ol.fft_fwd(np.eye(1, N, 1, dtype=np.int32)[0])[:8].tolist()}
    # keys + gates per parameter set
    for pset, name in ((0, "A"), (1, "B")):
        o = ol.Oracle(pset, 1)
        bits0, bits1, bits2 = np.array([0, 0, 1, 1]), np.array([0, 1, 0, 1]), np.array([1, 0, 0, 1])
        c0, c1, c2 = o.encrypt_bits(bits0, 2, 0), o.encrypt_bits(bits1, 2, 100), o.encrypt_bits(bits2, 2, 200)
        e = {"n": o.n, "l": o.l, "key_seed": 1, "enc_seed": 2,
             "lwe_key_sha": sha(o.lwe_key), "tlwe_key_sha": sha(o.tlwe_key), "bk_sha": sha(o.bk),
             "ksk_sha": sha(o.ksk), "bkfft_sha": sha(o.bkfft + 0.0),  # +0.0 canonicalises -0
             "c0_sha": sha(c0), "c1_sha": sha(c1), "gates": {}}
        # MAJ and XOR3 (round 6) are this repo's extension gates, not libtfhe functions: same bootstrap, three-input linear stage
        for opn in ["NAND", "AND", "OR", "NOR", "XOR", "XNOR", "ANDNY", "ANDYN", "ORNY", "ORYN", "MUX", "NOT", "MAJ", "XOR3"]:
            out = o.gate_batch(ol.OPS[opn], c0, c1 if opn != "NOT" else None, c2 if opn in ("MUX", "MAJ", "XOR3") else None)
            e["gates"][opn] = {"sha": sha(out), "bits": o.decrypt_bits(out).tolist()}
            if opn in ("NAND", "MUX", "MAJ", "XOR3"):
                arrays[f"{name}_{opn}_out"] = out
        t = o.gate_linear(ol.OPS["NAND"], c0[3], c1[3])
        u = o.blind_rotate_extract(t)
        e["nand_11_t_sha"], e["nand_11_u_sha"] = sha(t), sha(u)
        arrays[f"{name}_nand_11_u"] = u
        arrays[f"{name}_c0"], arrays[f"{name}_c1"], arrays[f"{name}_c2"] = c0, c1, c2
        g[name] = e
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(g, f, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(HERE, "golden_arrays.npz"), **arrays)
    print("wrote golden.json, golden_arrays.npz")


if __name__ == "__main__":
    main()
