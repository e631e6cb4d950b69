"""ctypes binding of the CPU oracle (oracle/liboracle.so).  Test-side only.

The oracle is test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg load it (see oracle/tfhe_oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
N = 1024

OPS = dict(NAND=0, AND=1, OR=2, NOR=3, XOR=4, XNOR=5, ANDNY=6, ANDYN=7, ORNY=8, ORYN=9,
           MUX=10, NOT=11, COPY=12, CONST0=13, CONST1=14, MAJ=15, XOR3=16)


class OrcParams(C.Structure):
    _fields_ = [("n", C.c_int32), ("l", C.c_int32), ("Bgbit", C.c_int32), ("ks_t", C.c_int32),
                ("ks_basebit", C.c_int32), ("ks_stdev", C.c_double), ("bk_stdev", C.c_double)]


def build_oracle():
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("tfhe_oracle.c", "tfhe_oracle.h")]
    srcs.append(os.path.join(ROOT, "eoc_tfhe_amd", "csrc", "canon_twiddles.h"))   # the Makefile copies it into oracle/
    if (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return so


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build_oracle())
        i32p = np.ctypeslib.ndpointer(np.int32, flags="C")
        f64p = np.ctypeslib.ndpointer(np.float64, flags="C")
        PP = C.POINTER(OrcParams)
        L.orc_default_params.argtypes = [C.c_int, PP]
        L.orc_stream_key.restype = C.c_uint64
        L.orc_stream_key.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64]
        L.orc_rng_u64.restype = C.c_uint64
        L.orc_rng_u64.argtypes = [C.c_uint64, C.c_uint64]
        L.orc_gaussian32.restype = C.c_int32
        L.orc_gaussian32.argtypes = [C.c_uint64, C.c_uint64, C.c_int32, C.c_double]
        L.orc_modswitch_to_torus32.restype = C.c_int32
        L.orc_modswitch_to_torus32.argtypes = [C.c_int32, C.c_int32]
        L.orc_modswitch_from_torus32.restype = C.c_int32
        L.orc_modswitch_from_torus32.argtypes = [C.c_int32, C.c_int32]
        L.orc_keygen_secret.argtypes = [PP, C.c_uint64, i32p, i32p]
        L.orc_keygen_ksk.argtypes = [PP, C.c_uint64, i32p, i32p, i32p]
        L.orc_keygen_bk.argtypes = [PP, C.c_uint64, i32p, i32p, i32p]
        L.orc_bk_to_fft.argtypes = [PP, i32p, f64p]
        L.orc_lwe_encrypt.argtypes = [PP, i32p, C.c_uint64, C.c_uint64, C.c_int32, C.c_double, i32p]
        L.orc_lwe_phase.restype = C.c_int32
        L.orc_lwe_phase.argtypes = [PP, i32p, i32p]
        L.orc_encrypt_bit.argtypes = [PP, i32p, C.c_uint64, C.c_uint64, C.c_int, i32p]
        L.orc_decrypt_bit.argtypes = [PP, i32p, i32p]
        L.orc_fft_fwd.argtypes = [i32p, f64p]
        L.orc_fft_inv.argtypes = [f64p, i32p]
        L.orc_fft_inv_raw.argtypes = [f64p, f64p]
        L.orc_gate_linear.argtypes = [PP, C.c_int, i32p, i32p, i32p]
        L.orc_modswitch_sample.argtypes = [PP, i32p, i32p, i32p]
        L.orc_blind_rotate_step.argtypes = [PP, C.c_void_p, C.c_void_p, C.c_int, i32p, C.c_int]
        L.orc_blind_rotate_extract.argtypes = [PP, f64p, i32p, C.c_int32, i32p]
        L.orc_keyswitch.argtypes = [PP, i32p, i32p, i32p]
        L.orc_bootstrap.argtypes = [PP, f64p, i32p, i32p, C.c_int32, i32p]
        L.orc_gate.argtypes = [PP, f64p, i32p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, i32p]
        L.orc_gate_batch.argtypes = [PP, f64p, i32p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, i32p, C.c_size_t, C.c_int]
        L.orc_max_threads.restype = C.c_int
        L.orc_dbg_max_conv.restype = C.c_double
        L.orc_dbg_max_conv.argtypes = [C.c_int]
        _lib = L
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Oracle:
    """Keys + hot path of the oracle for one parameter set and key seed."""

    def __init__(self, pset=0, seed=1, n_override=None, with_bk=True):
        self.L = lib()
        self.p = OrcParams()
        assert self.L.orc_default_params(pset, C.byref(self.p)) == 0
        if n_override is not None:
            self.p.n = n_override
        self.seed = seed
        p = self.p
        self.n, self.l, self.kpl = p.n, p.l, 2 * p.l
        self.lwe_key = np.zeros(p.n, np.int32)
        self.tlwe_key = np.zeros(N, np.int32)
        self.L.orc_keygen_secret(C.byref(p), seed, self.lwe_key, self.tlwe_key)
        self.bk = self.bkfft = self.ksk = None
        if with_bk:
            self.gen_cloud()

    def gen_cloud(self):
        p = self.p
        base = 1 << p.ks_basebit
        self.ksk = np.zeros((N * p.ks_t * (base - 1), p.n + 1), np.int32)
        self.L.orc_keygen_ksk(C.byref(p), self.seed, self.lwe_key, self.tlwe_key, self.ksk)
        self.bk = np.zeros((p.n, self.kpl, 2, N), np.int32)
        self.L.orc_keygen_bk(C.byref(p), self.seed, self.lwe_key, self.tlwe_key, self.bk)
        self.bkfft = np.zeros((p.n, self.kpl, 2, N), np.float64)
        self.L.orc_bk_to_fft(C.byref(p), self.bk, self.bkfft)

    # -- samples ---------------------------------------------------------------------------
    def encrypt_bits(self, bits, enc_seed, first_idx=0):
        bits = np.asarray(bits).astype(np.int64).ravel()
        out = np.zeros((len(bits), self.n + 1), np.int32)
        for i, b in enumerate(bits):
            self.L.orc_encrypt_bit(C.byref(self.p), self.lwe_key, enc_seed, first_idx + i, int(b), out[i])
        return out

    def decrypt_bits(self, cts):
        cts = np.ascontiguousarray(cts, np.int32).reshape(-1, self.n + 1)
        return np.array([self.L.orc_decrypt_bit(C.byref(self.p), self.lwe_key, c) for c in cts], np.int64)

    def phases(self, cts):
        cts = np.ascontiguousarray(cts, np.int32).reshape(-1, self.n + 1)
        return np.array([self.L.orc_lwe_phase(C.byref(self.p), self.lwe_key, c) for c in cts], np.int64)

    # -- hot path --------------------------------------------------------------------------
    def gate_batch(self, op, in0, in1=None, in2=None, ops=None, nthreads=0):
        in0 = np.ascontiguousarray(in0, np.int32)
        in1 = None if in1 is None else np.ascontiguousarray(in1, np.int32)
        in2 = None if in2 is None else np.ascontiguousarray(in2, np.int32)
        ops = None if ops is None else np.ascontiguousarray(ops, np.uint8)
        out = np.zeros_like(in0)
        rc = self.L.orc_gate_batch(C.byref(self.p), self.bkfft, self.ksk, int(op), _ptr(ops), _ptr(in0),
                                   _ptr(in1), _ptr(in2), out, in0.shape[0], nthreads)
        assert rc == 0
        return out

    def blind_rotate_extract(self, t, mu=1 << 29):
        u = np.zeros(N + 1, np.int32)
        self.L.orc_blind_rotate_extract(C.byref(self.p), self.bkfft, np.ascontiguousarray(t, np.int32), mu, u)
        return u

    def keyswitch(self, u):
        out = np.zeros(self.n + 1, np.int32)
        self.L.orc_keyswitch(C.byref(self.p), self.ksk, np.ascontiguousarray(u, np.int32), out)
        return out

    def gate_linear(self, op, ca, cb):
        t = np.zeros(self.n + 1, np.int32)
        assert self.L.orc_gate_linear(C.byref(self.p), op, np.ascontiguousarray(ca, np.int32),
                                      np.ascontiguousarray(cb, np.int32), t) == 0
        return t


def fft_fwd(poly):
    spec = np.zeros(N, np.float64)
    lib().orc_fft_fwd(np.ascontiguousarray(poly, np.int32), spec)
    return spec


def fft_inv(spec):
    poly = np.zeros(N, np.int32)
    lib().orc_fft_inv(np.ascontiguousarray(spec, np.float64), poly)
    return poly


def fft_inv_raw(spec):
    """the inverse transform's values BEFORE Torus32(int64(.)): what the conversion contract is stated on"""
    vals = np.zeros(N, np.float64)
    lib().orc_fft_inv_raw(np.ascontiguousarray(spec, np.float64), vals)
    return vals
