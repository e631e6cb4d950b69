"""CPU tests of the product's host side: the C-ABI library loads and exports every declared symbol,
client-side functions agree with the oracle bit for bit, the hot path refuses to run without a GPU."""
import base64
import os
import re

import numpy as np
import pytest

import oracle_lib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = 1024


@pytest.fixture(scope="module")
def eoc(built_lib):
    import eoc_tfhe_amd
    return eoc_tfhe_amd


def test_library_exports_every_declared_symbol(eoc):
    syms = eoc.abi_symbols()
    assert len(syms) >= 50 and "eoc_gate_batch_device" in syms and "gateNAND" in syms
    lib = eoc.lib()
    assert [s for s in syms if not hasattr(lib, s)] == []


def test_twiddle_table_is_pinned_and_the_oracle_builds_from_a_copy_of_it():
    """one committed copy of the generated table (eoc_tfhe_amd/csrc/canon_twiddles.h); oracle/Makefile copies it next to
    the oracle's source, so the two sides of every parity test read the same constants; the values are checked against
    cos / sin and pinned by their SHA-256"""
    import hashlib
    import subprocess
    a = open(os.path.join(ROOT, "eoc_tfhe_amd", "csrc", "canon_twiddles.h")).read()
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "canon_twiddles.h"])
    assert open(os.path.join(ROOT, "oracle", "canon_twiddles.h")).read() == a
    tracked = subprocess.run(["git", "-C", ROOT, "ls-files", "oracle/canon_twiddles.h"], capture_output=True, text=True)
    assert tracked.returncode != 0 or tracked.stdout.strip() == ""      # generated, never committed
    vals = re.findall(r"\{(\S+), (\S+)\}", a)
    assert len(vals) == 1024
    tab = np.array([[float.fromhex(c), float.fromhex(s)] for c, s in vals])
    k = np.arange(1024)
    assert np.abs(tab[:, 0] - np.cos(np.pi * k / 1024)).max() < 4e-16
    assert np.abs(tab[:, 1] - np.sin(np.pi * k / 1024)).max() < 4e-16
    assert tab[0].tolist() == [1.0, 0.0] and tab[512].tolist() == [0.0, 1.0] and tab[256, 0] == tab[256, 1]
    assert hashlib.sha256(tab.tobytes()).hexdigest() == \
        "e2258706aa14d5f4346fdd6433ab56676613dc50847456f5f4d3c847fc3218fe"


def test_params(eoc):
    a, b = eoc.default_params(0), eoc.default_params(1)
    assert (a.n, a.l, a.Bgbit, a.ks_t, a.ks_basebit) == (500, 2, 10, 8, 2)
    assert (b.n, b.l, b.Bgbit, b.ks_t, b.ks_basebit) == (630, 3, 7, 8, 2)
    assert a.ks_stdev == 2.44e-5 and b.ks_stdev == 2.0**-15 and b.bk_stdev == 2.0**-25
    import ctypes as C
    p = eoc.Params()
    L = eoc.lib()
    assert L.eoc_params_for_lambda(80, C.byref(p)) == 0 and p.n == 500
    assert L.eoc_params_for_lambda(110, C.byref(p)) == 0 and p.n == 630   # what lambda=110 really selects
    assert L.eoc_params_for_lambda(128, C.byref(p)) == 0 and p.n == 630   # the reference's minimum_lambda
    assert L.eoc_params_for_lambda(129, C.byref(p)) < 0 and L.eoc_params_for_lambda(0, C.byref(p)) < 0


@pytest.mark.parametrize("pset,n", [(0, 16), (1, 12), (0, None)])
def test_keygen_encrypt_decrypt_match_oracle(eoc, pset, n):
    p = eoc.default_params(pset)
    if n:
        p.n = n
    sk = eoc.SecretKey(p, 42)
    o = ol.Oracle(pset, 42, n_override=n)
    assert np.array_equal(sk.lwe_key, o.lwe_key) and np.array_equal(sk.tlwe_key, o.tlwe_key)
    assert np.array_equal(sk.bk, o.bk) and np.array_equal(sk.ksk, o.ksk)
    bits = np.random.default_rng(1).integers(0, 2, 33)
    c = sk.encrypt_bits(bits, 5, 17)
    assert np.array_equal(c, o.encrypt_bits(bits, 5, 17))
    assert np.array_equal(sk.decrypt_bits(c), bits) and np.array_equal(o.decrypt_bits(c), bits)
    assert sk.phase(c[0]) == o.phases(c[:1])[0]
    L = eoc.lib()
    for mu, M in [(1, 8), (-1, 8), (42, 2**31 - 1), (-7, 12345)]:
        assert L.eoc_modswitch_to_torus32(mu, M) == ol.lib().orc_modswitch_to_torus32(mu, M)
    for ph in [0, 1, -1, 2**31 - 1, -2**31, 99999999]:
        assert L.eoc_modswitch_from_torus32(ph, 2048) == ol.lib().orc_modswitch_from_torus32(ph, 2048)


def test_key_image_sizes(eoc):
    import ctypes as C
    L = eoc.lib()
    a = eoc.default_params(0)
    assert L.eoc_bkfft_bytes(C.byref(a)) == 32_768_000            # SURVEY.md 8(d), Set A
    assert L.eoc_ksk_row_stride(C.byref(a)) == 512
    assert L.eoc_ksk_dev_bytes(C.byref(a)) == 1024 * 8 * 3 * 512 * 4
    b = eoc.default_params(1)
    assert L.eoc_bkfft_bytes(C.byref(b)) == 61_931_520            # Set B
    assert L.eoc_bk_len(C.byref(a)) == 500 * 4 * 2 * 1024 and L.eoc_ksk_len(C.byref(a)) == 1024 * 8 * 3 * 501


def test_hot_path_fails_loudly_without_gpu(eoc):
    """no CPU fallback: on a box without a GPU every gate entry point errors out"""
    import ctypes as C
    L = eoc.lib()
    if L.eoc_device_count() > 0:
        pytest.skip("GPU present")
    p = eoc.default_params(0)
    with pytest.raises(eoc.EocError, match="no usable HIP device"):
        eoc.Engine(p)
    with pytest.raises(eoc.EocError):
        eoc.gpu_init(p)
    x = np.zeros((1, 501), np.int32)
    with pytest.raises(eoc.EocError, match="no CPU fallback"):
        eoc.gate_batch(0, x, x)
    # string API: reference behaviour -- NULL (None) + message on stderr
    assert eoc.Tfhe.encryptBit(1) is None
    assert eoc.Tfhe.decryptBit("AAAA") == -1
    assert eoc.Tfhe.nand("AAAA", "AAAA") is None
    assert eoc.Tfhe.generateGateKey(80, 1) is None   # cannot bring the engine up
    # round-3 entry points: the asynchronous batch, the worker counter, the RCCL rehearsal
    with pytest.raises(eoc.EocError, match="no CPU fallback"):
        eoc.gate_batch_submit(0, x, x, out=np.zeros_like(x))
    with pytest.raises(eoc.EocError, match="unknown ticket"):
        eoc.gate_batch_wait(1)
    assert L.eoc_worker_wakeups(0) == 0 and L.eoc_gpu_engine_count() == 0
    assert L.eoc_rccl_selftest(0, 4096) != 0         # no device to run it on: an error code, not a crash
    with pytest.raises(eoc.EocError, match="no global engine"):
        eoc.Engine.borrow_global()


def test_circuit_bootstrap_count(eoc):
    G = eoc.Gate
    gates = [G(eoc.OPS["XOR"], 0, 1, -1, 2), G(eoc.OPS["MUX"], 0, 1, 2, 3), G(eoc.OPS["NOT"], 3, -1, -1, 4),
             G(eoc.OPS["AND"], 4, 2, -1, 5)]
    assert eoc.circuit_bootstraps(gates) == 1 + 2 + 0 + 1


def test_shard_partition():
    from eoc_tfhe_amd.distributed import shard
    for total in (0, 1, 7, 1024, 2**20, 1000003):
        for world in (1, 2, 3, 8):
            blocks = [shard(total, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == total
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in blocks]
            assert max(sizes) - min(sizes) <= 1


def test_chacha20_block_rfc8439_known_answer(built_lib):
    """the block function behind the secure sampler (PRNG v2) against RFC 8439 section 2.3.2"""
    key = np.arange(32, dtype=np.uint8)
    nonce = np.frombuffer(bytes.fromhex("000000090000004a00000000"), np.uint8).copy()
    out = np.zeros(64, np.uint8)
    built_lib.eoc_dbg_chacha20_block(key.ctypes.data, 1, nonce.ctypes.data, out.ctypes.data)
    assert bytes(out) == bytes.fromhex(
        "10f1e7e4d13b5915500fdd1fa32071c4c7d1f4c733c068030422aa9ac3d46c4e"
        "d2826446079faa0914c2d705d98b02a2b5129cd1de164eb9cbd083e8a2503c4e")


def test_secure_mode_keys_and_encryption(built_lib):
    """eoc_keygen_secure: keys from getrandom + ChaCha20 (two calls differ, a fixed master key reproduces), a valid
    TGSW/LWE structure (bits decrypt), export/import through the EOCSK2 blob, keyed encryption decrypts and never
    repeats a mask across indices"""
    import eoc_tfhe_amd as eoc
    import ctypes as C
    p = eoc.default_params(0)
    p.n = 48
    a, b = eoc.SecretKey(p, None, with_cloud_key=False), eoc.SecretKey(p, None, with_cloud_key=False)
    assert built_lib.eoc_sk_is_secure(a.h) == 1 and not np.array_equal(a.tlwe_key, b.tlwe_key)
    assert 300 < a.tlwe_key.sum() < 724                      # about half of 1024 key bits set
    m = bytes(range(7, 39))
    k1, k2 = eoc.SecretKey(p, None, master=m), eoc.SecretKey(p, None, master=m)
    assert np.array_equal(k1.bk, k2.bk) and np.array_equal(k1.ksk, k2.ksk)
    det = eoc.SecretKey(p, 5)                                 # the reproducible mode is a different generator
    assert built_lib.eoc_sk_is_secure(det.h) == 0 and not np.array_equal(det.tlwe_key, k1.tlwe_key)
    # keyed encryption
    bits = np.random.default_rng(1).integers(0, 2, 200).astype(np.uint8)
    enc_key = np.frombuffer(os.urandom(32), np.uint8).copy()
    cts = np.zeros((200, p.n + 1), np.int32)
    assert built_lib.eoc_encrypt_bits_keyed(k1.h, enc_key.ctypes.data, 0, bits.ctypes.data, 200, cts.ctypes.data) == 0
    assert np.array_equal(k1.decrypt_bits(cts), bits)
    assert len({row[:-1].tobytes() for row in cts}) == 200    # no mask is ever repeated
    # EOCSK2 round trip
    need = built_lib.eoc_secret_key_export(k1.h, None, 0)
    blob = np.zeros(need, np.uint8)
    built_lib.eoc_secret_key_export(k1.h, blob.ctypes.data, need)
    assert bytes(blob[:6]) == b"EOCSK2" and need == 8 + 36 + 32 + p.n + 1024
    h = C.c_void_p()
    assert built_lib.eoc_secret_key_import(blob.ctypes.data, need, 1, C.byref(h)) == 0
    n_bk = built_lib.eoc_bk_len(C.byref(p))
    got = np.ctypeslib.as_array(C.cast(built_lib.eoc_sk_bk(h), C.POINTER(C.c_int32)), (n_bk,))
    assert np.array_equal(got, k1.bk.ravel())
    built_lib.eoc_secret_key_free(h)
    blob[50] ^= 1                                             # a damaged master key no longer matches the key bits
    assert built_lib.eoc_secret_key_import(blob.ctypes.data, need, 0, C.byref(h)) != 0


def test_engine_refuses_gadget_shapes_beyond_the_fp64_contract(built_lib):
    """l * Bg > 8192: an external-product coefficient can exceed what binary64 holds exactly (l * Bg * 2^41); the engine
    refuses the shape at creation -- before it looks for a device, so the refusal is the same on a box without a GPU"""
    import eoc_tfhe_amd as eoc
    for l, bgbit in ((2, 16), (1, 14), (4, 12), (3, 12)):
        p = eoc.default_params(0)
        p.l, p.Bgbit = l, bgbit
        with pytest.raises(eoc.EocError, match="FP64 external product is not exact"):
            eoc.Engine(p)
    p = eoc.default_params(0)
    p.l, p.Bgbit = 2, 12                       # l * Bg = 8192: the last accepted shape (no device here: another error)
    try:
        eoc.Engine(p).close()
    except eoc.EocError as e:
        assert "no usable HIP device" in str(e)
