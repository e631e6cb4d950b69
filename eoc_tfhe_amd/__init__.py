"""eoc_tfhe_amd -- Python host side of libeoc_tfhe_gpu.so (ctypes over the C ABI of
include/eoc_tfhe_gpu.h).

The product is the shared library; this module is the thin façade a Python host uses, in the way
ao-tfhe/tfhe.lua:1-53 wraps the Lua C module (`Tfhe.*` pass-throughs to `Tfhe.backend.*`).
Nothing here computes a gate on the CPU: every hot-path call goes to the HIP engine and raises
if the library or a GPU is missing.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EOC_TFHE_LIB") or os.path.join(_HERE, "libeoc_tfhe_gpu.so")
N = 1024

OPS = dict(NAND=0, AND=1, OR=2, NOR=3, XOR=4, XNOR=5, ANDNY=6, ANDYN=7, ORNY=8, ORYN=9,
           MUX=10, NOT=11, COPY=12, CONST0=13, CONST1=14,
           MAJ=15, XOR3=16)   # extension gates: majority / three-input parity, one bootstrap each (include/eoc_tfhe_gpu.h)


class EocError(RuntimeError):
    pass


class Params(C.Structure):
    """eoc_params (TFheGateBootstrappingParameterSet flattened)."""
    _fields_ = [("n", C.c_int32), ("l", C.c_int32), ("Bgbit", C.c_int32), ("ks_t", C.c_int32),
                ("ks_basebit", C.c_int32), ("ks_stdev", C.c_double), ("bk_stdev", C.c_double)]

    def copy(self):
        q = Params()
        C.memmove(C.byref(q), C.byref(self), C.sizeof(Params))
        return q


class Gate(C.Structure):
    """eoc_gate netlist entry."""
    _fields_ = [("op", C.c_int32), ("in0", C.c_int32), ("in1", C.c_int32), ("in2", C.c_int32),
                ("out", C.c_int32)]


_lib = None


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  PyTorch-ROCm bundles its own libamdhip64.so (SONAME
    libamdhip64.so.7); if our library pulled /opt/rocm's copy in first, a later `import torch`
    would load a second runtime and one of the two would see no device.  Pre-loading torch's copy
    (when torch is installed) makes both resolve to the same one, whatever the import order."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    """Load the C-ABI library (fails loudly when it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EocError(f"{LIB_PATH} is missing: run `python -m eoc_tfhe_amd.build` "
                       "(there is no Python/CPU fallback for the gate path)")
    _share_hip_runtime_with_torch()
    L = C.CDLL(LIB_PATH)
    PP = C.POINTER(Params)
    vp, sz, u64 = C.c_void_p, C.c_size_t, C.c_uint64
    sig = {
        "eoc_default_params": (C.c_int, [C.c_int, PP]),
        "eoc_params_for_lambda": (C.c_int, [C.c_int, PP]),
        "eoc_keygen": (C.c_int, [PP, u64, C.c_int, C.POINTER(vp)]),
        "eoc_keygen_secure": (C.c_int, [PP, C.c_int, C.POINTER(vp)]),
        "eoc_keygen_from_master": (C.c_int, [PP, vp, C.c_int, C.POINTER(vp)]),
        "eoc_sk_is_secure": (C.c_int, [vp]),
        "eoc_encrypt_bits_keyed": (C.c_int, [vp, vp, u64, vp, sz, vp]),
        "eoc_dbg_chacha20_block": (None, [vp, C.c_uint32, vp, vp]),
        "eoc_secret_key_free": (None, [vp]),
        "eoc_sk_params": (PP, [vp]),
        "eoc_sk_lwe_key": (vp, [vp]),
        "eoc_sk_tlwe_key": (vp, [vp]),
        "eoc_sk_bk": (vp, [vp]),
        "eoc_sk_ksk": (vp, [vp]),
        "eoc_bk_len": (sz, [PP]),
        "eoc_ksk_len": (sz, [PP]),
        "eoc_encrypt_bits": (C.c_int, [vp, u64, u64, vp, sz, vp]),
        "eoc_decrypt_bits": (C.c_int, [vp, vp, sz, vp]),
        "eoc_lwe_encrypt": (C.c_int, [vp, u64, u64, C.c_int32, C.c_double, vp]),
        "eoc_lwe_phase": (C.c_int32, [vp, vp]),
        "eoc_host_threads": (C.c_int, []),
        "eoc_modswitch_to_torus32": (C.c_int32, [C.c_int32, C.c_int32]),
        "eoc_modswitch_from_torus32": (C.c_int32, [C.c_int32, C.c_int32]),
        "eoc_device_count": (C.c_int, []),
        "eoc_engine_create": (C.c_int, [C.c_int, PP, C.POINTER(vp)]),
        "eoc_engine_destroy": (None, [vp]),
        "eoc_last_error": (C.c_char_p, []),
        "eoc_device_alloc": (C.c_int, [vp, sz, C.POINTER(vp)]),
        "eoc_device_free": (C.c_int, [vp, vp]),
        "eoc_host_to_device": (C.c_int, [vp, vp, vp, sz]),
        "eoc_device_to_host": (C.c_int, [vp, vp, vp, sz]),
        "eoc_engine_synchronize": (C.c_int, [vp]),
        "eoc_bkfft_bytes": (sz, [PP]),
        "eoc_ksk_dev_bytes": (sz, [PP]),
        "eoc_ksk_row_stride": (sz, [PP]),
        "eoc_engine_load_cloud_key": (C.c_int, [vp, vp, vp]),
        "eoc_engine_set_cloud_key_device": (C.c_int, [vp, vp, vp]),
        "eoc_engine_build_cloud_key_device": (C.c_int, [vp, vp, vp, vp, vp]),
        "eoc_engine_set_profiling": (C.c_int, [vp, C.c_int]),
        "eoc_engine_kernel_times": (C.c_int, [vp, C.POINTER(C.c_double * 3), C.POINTER(u64 * 3), C.c_int]),
        "eoc_engine_cloud_key_device": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp)]),
        "eoc_gate_batch_device": (C.c_int, [vp, C.c_int, vp, vp, vp, vp, vp, sz, vp]),
        "eoc_circuit_run_device": (C.c_int, [vp, vp, sz, vp, sz, sz, vp]),
        "eoc_circuit_bootstraps": (sz, [vp, sz]),
        "eoc_netlist_optimize": (C.c_int64, [vp, sz, vp, sz, vp]),
        "eoc_netlist_optimize_ex": (C.c_int64, [vp, sz, vp, sz, vp, C.c_uint]),
        "eoc_netlist_levels": (C.c_int64, [vp, sz, vp, vp]),
        "eoc_netlist_cost": (C.c_int64, [vp, sz, sz, sz]),
        "eoc_dbg_fft_fwd_device": (C.c_int, [vp, vp, vp, sz, vp]),
        "eoc_dbg_fft_inv_device": (C.c_int, [vp, vp, vp, sz, vp]),
        "eoc_blind_rotate_device": (C.c_int, [vp, vp, vp, sz, vp]),
        "eoc_keyswitch_device": (C.c_int, [vp, vp, vp, sz, vp]),
        "eoc_engine_stats": (C.c_int, [vp, C.POINTER(u64 * 3)]),
        "eoc_stats": (C.c_int, [C.POINTER(u64 * 3)]),
        "eoc_gpu_init": (C.c_int, [C.c_int, PP]),
        "eoc_gpu_init_multi": (C.c_int, [vp, C.c_int, PP]),
        "eoc_gpu_init_from_env": (C.c_int, [PP]),
        "eoc_gpu_set_devices": (C.c_int, [C.POINTER(C.c_int), C.c_int]),
        "eoc_gpu_engine_count": (C.c_int, []),
        "eoc_global_engine_at": (vp, [C.c_int]),
        "eoc_upload_cloud_key_arrays": (C.c_int, [vp, vp]),
        "eoc_stats_multi": (C.c_int, [vp, C.c_int, C.POINTER(C.c_double)]),
        "eoc_key_broadcast_method": (C.c_char_p, []),
        "eoc_host_path_buffer_grows": (u64, []),
        "eoc_rccl_origin": (C.c_char_p, []),
        "eoc_rccl_selftest": (C.c_int, [C.c_int, sz]),
        "eoc_worker_wakeups": (u64, [C.c_int]),
        "eoc_shard_range": (None, [sz, C.c_int, C.c_int, C.POINTER(sz), C.POINTER(sz)]),
        "eoc_gate_batch_submit": (C.c_int, [C.c_int, vp, vp, vp, vp, vp, sz, C.POINTER(u64)]),
        "eoc_gate_batch_wait": (C.c_int, [u64]),
        "eoc_global_gate_batch_submit": (C.c_int, [C.c_int, vp, vp, vp, vp, vp, sz, C.POINTER(u64)]),
        "eoc_host_alloc": (vp, [sz]),
        "eoc_host_free": (None, [vp]),
        "eoc_engine_reserve": (C.c_int, [vp, sz, sz, sz]),
        "eoc_engine_workspace_grows": (u64, [vp]),
        "eoc_engine_blind_rotate_launches": (u64, [vp]),
        "eoc_engine_blind_rotate_wide_launches": (u64, [vp]),
        "eoc_engine_resident_jobs": (C.c_size_t, [vp]),
        "eoc_engine_device": (C.c_int, [vp]),
        "eoc_engine_params": (PP, [vp]),
        "eoc_engine_adopt_cloud_key_device": (C.c_int, [vp, vp, vp]),
        "eoc_upload_cloud_key": (C.c_int, [vp]),
        "eoc_global_engine": (vp, []),
        "eoc_gpu_shutdown": (None, []),
        "eoc_gate_batch": (C.c_int, [C.c_int, vp, vp, vp, vp, vp, sz]),
        "eoc_circuit_run": (C.c_int, [vp, sz, vp, sz, sz]),
        "generateGateKey": (vp, [C.c_int, u64]),
        "resetGateKey": (None, []),
        "encryptBit": (vp, [C.c_int, C.c_char_p]),
        "decryptBit": (C.c_int, [C.c_char_p, C.c_char_p]),
        "constantBit": (vp, [C.c_int]),
        "gateNAND": (vp, [C.c_char_p, C.c_char_p, C.c_char_p]),
        "gateAND": (vp, [C.c_char_p, C.c_char_p, C.c_char_p]),
        "gateOR": (vp, [C.c_char_p, C.c_char_p, C.c_char_p]),
        "gateNOR": (vp, [C.c_char_p, C.c_char_p, C.c_char_p]),
        "gateXOR": (vp, [C.c_char_p, C.c_char_p, C.c_char_p]),
        "gateXNOR": (vp, [C.c_char_p, C.c_char_p, C.c_char_p]),
        "gateNOT": (vp, [C.c_char_p, C.c_char_p]),
        "gateMUX": (vp, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p]),
        "gateMAJ": (vp, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p]),
        "gateXOR3": (vp, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p]),
        # f1: the reference's 11 calls
        "generateSecretKey": (vp, [C.c_char_p, C.c_char_p]),
        "generatePublicKey": (vp, []),
        "encryptInteger": (vp, [C.c_int32, C.c_char_p]),
        "encryptInteger_dummy": (vp, [C.c_int32, C.c_char_p]),
        "decryptInteger": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p]),
        "addCiphertexts": (vp, [C.c_char_p, C.c_char_p, C.c_char_p]),
        "subtractCiphertexts": (vp, [C.c_char_p, C.c_char_p, C.c_char_p]),
        "encrypt8BitASCIIString": (vp, [C.c_char_p, C.c_int16, C.c_char_p]),
        "decrypt8BitASCIIString": (vp, [C.c_char_p, C.c_int16, C.c_char_p, C.c_char_p, C.c_char_p]),
        "info": (None, []),
        "testJWT": (None, []),
        # f2: key export / import
        "eoc_secret_key_export": (sz, [vp, vp, sz]),
        "eoc_secret_key_import": (C.c_int, [vp, sz, C.c_int, C.POINTER(vp)]),
        "eoc_cloud_key_blob_bytes": (sz, [PP]),
        "eoc_cloud_key_export": (C.c_int, [vp, vp, sz]),
        "eoc_cloud_key_blob_params": (C.c_int, [vp, sz, PP]),
        "eoc_engine_create_from_cloud_key_blob": (C.c_int, [C.c_int, vp, sz, C.POINTER(vp)]),
        "eoc_global_params": (C.c_int, [PP]),
        "eoc_global_encrypt_bits": (C.c_int, [vp, sz, vp]),
        "eoc_global_decrypt_bits": (C.c_int, [vp, sz, vp]),
        "eoc_global_gate_batch": (C.c_int, [C.c_int, vp, vp, vp, vp, vp, sz]),
        "eoc_global_circuit_run": (C.c_int, [vp, sz, vp, sz, sz]),
        "exportSecretKey": (vp, []),
        "importSecretKey": (C.c_int, [C.c_char_p]),
        # f2, server side: the cloud ("public") key on the global context
        "exportCloudKey": (vp, []),
        "importCloudKey": (C.c_int, [C.c_char_p]),
        "exportCloudKeyToFile": (C.c_int, [C.c_char_p]),
        "importCloudKeyFromFile": (C.c_int, [C.c_char_p]),
        "eoc_global_cloud_key_export": (sz, [vp, sz]),
        "eoc_global_import_cloud_key_blob": (C.c_int, [vp, sz]),
        "eoc_global_key_mode": (C.c_int, []),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    _lib = L
    return L


ABI_SYMBOLS = None  # filled lazily by abi_symbols()


def abi_symbols():
    """Every symbol include/eoc_tfhe_gpu.h declares (parsed from the header)."""
    import re
    hdr = os.path.join(os.path.dirname(_HERE), "include", "eoc_tfhe_gpu.h")
    text = open(hdr).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{}]*\)\s*;", text)
    return sorted(set(names))


def _check(rc, what):
    if rc != 0:
        msg = lib().eoc_last_error()
        raise EocError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


def default_params(pset=0):
    p = Params()
    _check(lib().eoc_default_params(pset, C.byref(p)), "eoc_default_params")
    return p


def _take_str(ptr):
    """Copy and free() a heap C string returned by the string API (NULL -> None, like Lua nil)."""
    if not ptr:
        return None
    s = C.string_at(ptr).decode()
    C.CDLL(None).free(C.c_void_p(ptr))
    return s


def _np_view(ptr, count, dtype):
    if not ptr:
        return None
    buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=count)


class SecretKey:
    """TFheGateBootstrappingSecretKeySet: LWE key, TLWE key and (optionally) the cloud key."""

    def __init__(self, params, seed, with_cloud_key=True, master=None):
        """seed: int -> reproducible test mode (PRNG v1, shared with the oracle; NOT secure);
        seed None -> secure mode (getrandom + ChaCha20), or from a 32-byte `master` key"""
        self.L = lib()
        self.h = C.c_void_p()
        self.params = params.copy()
        if master is not None:
            m = np.frombuffer(bytes(master), np.uint8)
            assert m.size == 32
            _check(self.L.eoc_keygen_from_master(C.byref(self.params), m.ctypes.data, int(with_cloud_key), C.byref(self.h)),
                   "eoc_keygen_from_master")
        elif seed is None:
            _check(self.L.eoc_keygen_secure(C.byref(self.params), int(with_cloud_key), C.byref(self.h)), "eoc_keygen_secure")
        else:
            _check(self.L.eoc_keygen(C.byref(self.params), seed, int(with_cloud_key), C.byref(self.h)), "eoc_keygen")
        self.n = params.n
        self.seed = seed

    def __del__(self):
        if getattr(self, "h", None):
            self.L.eoc_secret_key_free(self.h)
            self.h = None

    @property
    def lwe_key(self):
        return _np_view(self.L.eoc_sk_lwe_key(self.h), self.n, np.int32)

    @property
    def tlwe_key(self):
        return _np_view(self.L.eoc_sk_tlwe_key(self.h), N, np.int32)

    @property
    def bk(self):
        p = self.params
        v = _np_view(self.L.eoc_sk_bk(self.h), self.L.eoc_bk_len(C.byref(p)), np.int32)
        return None if v is None else v.reshape(p.n, 2 * p.l, 2, N)

    @property
    def ksk(self):
        p = self.params
        v = _np_view(self.L.eoc_sk_ksk(self.h), self.L.eoc_ksk_len(C.byref(p)), np.int32)
        return None if v is None else v.reshape(-1, p.n + 1)

    def export_bytes(self):
        """EOCSK1 blob (params | seed | key bits)."""
        need = self.L.eoc_secret_key_export(self.h, None, 0)
        buf = (C.c_ubyte * need)()
        self.L.eoc_secret_key_export(self.h, buf, need)
        return bytes(buf)

    @classmethod
    def from_bytes(cls, blob, with_cloud_key=True):
        self = cls.__new__(cls)
        self.L = lib()
        self.h = C.c_void_p()
        _check(self.L.eoc_secret_key_import(blob, len(blob), int(with_cloud_key), C.byref(self.h)),
               "eoc_secret_key_import")
        self.params = self.L.eoc_sk_params(self.h).contents.copy()
        self.n = self.params.n
        self.seed = None
        return self

    def export_cloud_key(self):
        """EOCCK1 blob (params | bk | ksk): everything a server needs, no secret material."""
        need = self.L.eoc_cloud_key_blob_bytes(C.byref(self.params))
        buf = np.empty(need, np.uint8)
        _check(self.L.eoc_cloud_key_export(self.h, buf.ctypes.data, need), "eoc_cloud_key_export")
        return buf

    def encrypt_bits(self, bits, enc_seed, first_idx=0):
        bits = np.ascontiguousarray(np.asarray(bits).ravel(), np.uint8)
        out = np.empty((bits.size, self.n + 1), np.int32)
        _check(self.L.eoc_encrypt_bits(self.h, enc_seed, first_idx, bits.ctypes.data, bits.size, out.ctypes.data),
               "eoc_encrypt_bits")
        return out

    def decrypt_bits(self, cts):
        cts = np.ascontiguousarray(cts, np.int32).reshape(-1, self.n + 1)
        out = np.empty(cts.shape[0], np.uint8)
        _check(self.L.eoc_decrypt_bits(self.h, cts.ctypes.data, cts.shape[0], out.ctypes.data), "eoc_decrypt_bits")
        return out

    def phase(self, ct):
        ct = np.ascontiguousarray(ct, np.int32)
        return self.L.eoc_lwe_phase(self.h, ct.ctypes.data)


class Engine:
    """One HIP engine (one GPU).  All array arguments are DEVICE pointers (ints) unless noted."""

    def __init__(self, params, device=0):
        self.L = lib()
        self.h = C.c_void_p()
        self.params = params.copy()
        _check(self.L.eoc_engine_create(device, C.byref(self.params), C.byref(self.h)), "eoc_engine_create")
        self.device = device
        self.n = params.n

    @classmethod
    def from_cloud_key_blob(cls, blob, device=0):
        """Server side: engine + key images from an EOCCK1 blob alone."""
        blob = np.ascontiguousarray(blob, np.uint8)
        self = cls.__new__(cls)
        self.L = lib()
        self.h = C.c_void_p()
        self.params = Params()
        _check(self.L.eoc_cloud_key_blob_params(blob.ctypes.data, blob.size, C.byref(self.params)),
               "eoc_cloud_key_blob_params")
        _check(self.L.eoc_engine_create_from_cloud_key_blob(device, blob.ctypes.data, blob.size, C.byref(self.h)),
               "eoc_engine_create_from_cloud_key_blob")
        self.device = device
        self.n = self.params.n
        return self

    @classmethod
    def borrow_global(cls, index=0):
        """Wrapper around engine `index` of the process-global context (eoc_gpu_init[_multi]): the same engine the
        host-buffer API drives, reachable through the device-pointer API too.  Not owned: close() leaves it alone."""
        self = cls.__new__(cls)
        self.L = lib()
        h = self.L.eoc_global_engine_at(index)
        if not h:
            raise EocError("borrow_global: no global engine (eoc_gpu_init not called)")
        self.h = C.c_void_p(h)
        self._borrowed = True
        self.params = self.L.eoc_engine_params(self.h).contents.copy()
        self.device = self.L.eoc_engine_device(self.h)
        self.n = self.params.n
        return self

    def close(self):
        if getattr(self, "h", None):
            if not getattr(self, "_borrowed", False):
                self.L.eoc_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    # -- cloud key ---------------------------------------------------------------------------
    def load_cloud_key(self, sk_or_bk, ksk=None):
        """Host torus-form BK/KSK -> device images (forward transform runs on the GPU)."""
        if isinstance(sk_or_bk, SecretKey):
            bk, ksk = sk_or_bk.bk, sk_or_bk.ksk
        else:
            bk = sk_or_bk
        bk = np.ascontiguousarray(bk, np.int32)
        ksk = np.ascontiguousarray(ksk, np.int32)
        _check(self.L.eoc_engine_load_cloud_key(self.h, bk.ctypes.data, ksk.ctypes.data), "eoc_engine_load_cloud_key")

    def build_cloud_key_device(self, sk, d_bkfft, d_ksk):
        """Like load_cloud_key but into caller-owned device buffers (e.g. torch tensors)."""
        bk = np.ascontiguousarray(sk.bk, np.int32)
        ksk = np.ascontiguousarray(sk.ksk, np.int32)
        _check(self.L.eoc_engine_build_cloud_key_device(self.h, bk.ctypes.data, ksk.ctypes.data, d_bkfft, d_ksk),
               "eoc_engine_build_cloud_key_device")

    def set_profiling(self, on=True):
        _check(self.L.eoc_engine_set_profiling(self.h, int(on)), "eoc_engine_set_profiling")

    def kernel_times(self, reset=True):
        ms, cnt = (C.c_double * 3)(), (C.c_uint64 * 3)()
        _check(self.L.eoc_engine_kernel_times(self.h, C.byref(ms), C.byref(cnt), int(reset)), "eoc_engine_kernel_times")
        names = ("prepare", "blind_rotate", "keyswitch")
        return {k: dict(ms=ms[i], launches=cnt[i]) for i, k in enumerate(names)}

    def set_cloud_key_device(self, d_bkfft, d_ksk):
        _check(self.L.eoc_engine_set_cloud_key_device(self.h, d_bkfft, d_ksk), "eoc_engine_set_cloud_key_device")

    def cloud_key_device(self):
        a, b = C.c_void_p(), C.c_void_p()
        _check(self.L.eoc_engine_cloud_key_device(self.h, C.byref(a), C.byref(b)), "eoc_engine_cloud_key_device")
        return a.value, b.value

    def download(self, d_ptr, nbytes, dtype=np.uint8):
        """device -> host numpy copy of a raw buffer"""
        out = np.empty(nbytes // np.dtype(dtype).itemsize, dtype)
        _check(self.L.eoc_device_to_host(self.h, out.ctypes.data, d_ptr, nbytes), "eoc_device_to_host")
        return out

    def synchronize(self):
        _check(self.L.eoc_engine_synchronize(self.h), "eoc_engine_synchronize")

    @property
    def bkfft_bytes(self):
        return self.L.eoc_bkfft_bytes(C.byref(self.params))

    @property
    def ksk_dev_bytes(self):
        return self.L.eoc_ksk_dev_bytes(C.byref(self.params))

    # -- hot path ----------------------------------------------------------------------------
    def gate_batch_device(self, op, d_in0, d_in1, d_in2, d_out, count, ops=None, stream=None):
        ops_p = None
        if ops is not None:
            ops = np.ascontiguousarray(ops, np.uint8)
            ops_p = ops.ctypes.data
        _check(self.L.eoc_gate_batch_device(self.h, int(op), ops_p, d_in0, d_in1, d_in2, d_out, count, stream),
               "eoc_gate_batch_device")

    def circuit_run_device(self, gates, d_wires, n_wires, instances, stream=None):
        arr = (Gate * len(gates))(*gates)
        _check(self.L.eoc_circuit_run_device(self.h, C.addressof(arr), len(gates), d_wires, n_wires, instances, stream),
               "eoc_circuit_run_device")

    def fft_fwd_device(self, d_polys, d_specs, count, stream=None):
        _check(self.L.eoc_dbg_fft_fwd_device(self.h, d_polys, d_specs, count, stream), "eoc_dbg_fft_fwd_device")

    def fft_inv_device(self, d_specs, d_polys, count, stream=None):
        _check(self.L.eoc_dbg_fft_inv_device(self.h, d_specs, d_polys, count, stream), "eoc_dbg_fft_inv_device")

    def blind_rotate_device(self, d_t, d_u, count, stream=None):
        _check(self.L.eoc_blind_rotate_device(self.h, d_t, d_u, count, stream), "eoc_blind_rotate_device")

    def keyswitch_device(self, d_u, d_out, count, stream=None):
        _check(self.L.eoc_keyswitch_device(self.h, d_u, d_out, count, stream), "eoc_keyswitch_device")

    def resident_jobs(self):
        """blind rotations that fill the device in one launch (8 x CUs where the one-wave-per-ciphertext kernel applies,
        4 x otherwise): cut long jobs at multiples of this"""
        return int(self.L.eoc_engine_resident_jobs(self.h))

    def stats(self):
        out = (C.c_uint64 * 3)()
        _check(self.L.eoc_engine_stats(self.h, C.byref(out)), "eoc_engine_stats")
        return dict(batches=out[0], bootstraps=out[1], keyswitches=out[2],
                    br_launches=int(self.L.eoc_engine_blind_rotate_launches(self.h)),
                    br_wide_launches=int(self.L.eoc_engine_blind_rotate_wide_launches(self.h)))


def circuit_bootstraps(gates):
    arr = (Gate * len(gates))(*gates)
    return lib().eoc_circuit_bootstraps(C.addressof(arr), len(gates))


def netlist_optimize(gates, outputs, extension_gates=True):
    """eoc_netlist_optimize(_ex): constant / NOT / COPY folding, MUX and carry fusion, dead-gate removal in the native library
    (the same rewriting as circuits.optimize).  extension_gates=False keeps the result inside libtfhe's boots* family (the
    carry as MUX instead of MAJ, no XOR3).  Raises EocError for netlists that are not single-assignment."""
    arr = (Gate * max(1, len(gates)))(*gates)
    out = (Gate * max(1, len(gates)))()
    outs = (C.c_int32 * max(1, len(outputs)))(*outputs)
    n = lib().eoc_netlist_optimize_ex(C.addressof(arr), len(gates), C.addressof(outs), len(outputs), C.addressof(out),
                                      0 if extension_gates else 1)
    if n < 0:
        raise EocError(f"eoc_netlist_optimize failed ({n}): not a single-assignment netlist?")
    return [Gate(out[k].op, out[k].in0, out[k].in1, out[k].in2, out[k].out) for k in range(n)]


def netlist_levels(gates):
    """eoc_netlist_levels: (level of every gate, number of levels, levels that hold a blind rotation)"""
    arr = (Gate * max(1, len(gates)))(*gates)
    lev = (C.c_int32 * max(1, len(gates)))()
    depth = C.c_int64(0)
    n = lib().eoc_netlist_levels(C.addressof(arr), len(gates), C.addressof(lev), C.addressof(depth))
    if n < 0:
        raise EocError(f"eoc_netlist_levels failed ({n})")
    return list(lev[:len(gates)]), int(n), int(depth.value)


def netlist_cost(gates, instances, resident_jobs=0):
    """eoc_netlist_cost: estimated run time over `instances` instances in units of 0.1 ms (the native twin of
    circuits.netlist_cost)"""
    arr = (Gate * max(1, len(gates)))(*gates)
    c = lib().eoc_netlist_cost(C.addressof(arr), len(gates), int(instances), int(resident_jobs))
    if c < 0:
        raise EocError(f"eoc_netlist_cost failed ({c})")
    return int(c)


# ---- host-buffer batch API (global context: one key, any number of GPUs), numpy in / numpy out ---------
def gpu_init(params, device=0, devices=None):
    """eoc_gpu_init / eoc_gpu_init_multi: `devices` = list of device ordinals (a device may repeat: several engines
    then share it, the one-GPU rehearsal of the N-GPU path)"""
    if devices is None:
        _check(lib().eoc_gpu_init(device, C.byref(params)), "eoc_gpu_init")
    else:
        arr = (C.c_int * len(devices))(*devices)
        _check(lib().eoc_gpu_init_multi(C.addressof(arr), len(devices), C.byref(params)), "eoc_gpu_init_multi")


def gpu_engine_count():
    return lib().eoc_gpu_engine_count()


def shard_range(total, rank, world):
    lo, hi = C.c_size_t(), C.c_size_t()
    lib().eoc_shard_range(total, rank, world, C.byref(lo), C.byref(hi))
    return lo.value, hi.value


def stats_multi():
    """per-engine counters, key replication seconds and method of the global context"""
    n = lib().eoc_gpu_engine_count()
    buf = (C.c_uint64 * (3 * max(1, n)))()
    secs = C.c_double()
    rc = lib().eoc_stats_multi(C.addressof(buf), n, C.byref(secs))
    if rc < 0:
        _check(rc, "eoc_stats_multi")
    per = [dict(batches=buf[3 * i], bootstraps=buf[3 * i + 1], keyswitches=buf[3 * i + 2]) for i in range(n)]
    return dict(engines=per, key_broadcast_s=secs.value, key_broadcast_method=lib().eoc_key_broadcast_method().decode(),
                host_buffer_grows=int(lib().eoc_host_path_buffer_grows()),
                rccl_origin=lib().eoc_rccl_origin().decode(),
                worker_wakeups=[int(lib().eoc_worker_wakeups(i)) for i in range(n)])


class PinnedArray:
    """numpy view of pinned host memory from eoc_host_alloc (true DMA source/target of the batch API)."""

    def __init__(self, shape, dtype=np.int32):
        self.nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self.ptr = lib().eoc_host_alloc(self.nbytes)
        if not self.ptr:
            raise EocError("eoc_host_alloc failed")
        buf = (C.c_char * self.nbytes).from_address(self.ptr)
        self.array = np.frombuffer(buf, dtype=dtype).reshape(shape)

    def free(self):
        if self.ptr:
            self.array = None
            lib().eoc_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def gate_batch_submit(op, in0, in1=None, in2=None, ops=None, out=None):
    """eoc_gate_batch_submit on PinnedArray-backed numpy views (C-contiguous int32 [count][n+1]); returns the ticket.
    The arrays must stay alive and untouched until gate_batch_wait(ticket)."""
    for a in (in0, in1, in2, out):
        if a is not None and (a.dtype != np.int32 or not a.flags.c_contiguous):
            raise EocError("gate_batch_submit: operands are C-contiguous int32 arrays")
    if out is None or (in0 is not None and out.shape != in0.shape):
        raise EocError("gate_batch_submit: `out` must be given and have the operands' shape")
    opsc = None if ops is None else np.ascontiguousarray(ops, np.uint8)
    t = C.c_uint64()
    _check(lib().eoc_gate_batch_submit(int(op), None if opsc is None else opsc.ctypes.data,
                                       None if in0 is None else in0.ctypes.data, None if in1 is None else in1.ctypes.data,
                                       None if in2 is None else in2.ctypes.data, out.ctypes.data, out.shape[0], C.byref(t)),
           "eoc_gate_batch_submit")
    return t.value


def gate_batch_wait(ticket):
    _check(lib().eoc_gate_batch_wait(ticket), "eoc_gate_batch_wait")


def gpu_shutdown():
    lib().eoc_gpu_shutdown()


def stats():
    """eoc_stats: counters of the global engine (batches, bootstraps, key switches)"""
    out = (C.c_uint64 * 3)()
    _check(lib().eoc_stats(C.byref(out)), "eoc_stats")
    return dict(batches=out[0], bootstraps=out[1], keyswitches=out[2])


def upload_cloud_key(sk):
    _check(lib().eoc_upload_cloud_key(sk.h), "eoc_upload_cloud_key")


def gate_batch(op, in0, in1=None, in2=None, ops=None, out=None, count=None, rowlen=None):
    """eoc_gate_batch on host arrays [count][n+1] int32.  Operand shapes are checked here (the C ABI takes bare
    pointers): every supplied operand must have exactly in0's shape, `ops` one opcode per row.  CONST0/CONST1 take
    no input: pass in0=None with `count` and `rowlen`."""
    if in0 is None:
        if ops is not None or int(op) not in (OPS["CONST0"], OPS["CONST1"]) or count is None or rowlen is None:
            raise EocError("gate_batch: in0 may be omitted only for CONST0/CONST1 with count and rowlen given")
        shape = (int(count), int(rowlen))
    else:
        in0 = np.ascontiguousarray(in0, np.int32)
        if in0.ndim != 2:
            raise EocError("gate_batch: operands are 2-d arrays [count][n+1]")
        shape = in0.shape
    in1c = None if in1 is None else np.ascontiguousarray(in1, np.int32)
    in2c = None if in2 is None else np.ascontiguousarray(in2, np.int32)
    opsc = None if ops is None else np.ascontiguousarray(ops, np.uint8)
    for name, a in (("in1", in1c), ("in2", in2c)):
        if a is not None and a.shape != shape:
            raise EocError(f"gate_batch: {name} has shape {a.shape}, expected {shape}")
    if opsc is not None and opsc.shape != (shape[0],):
        raise EocError(f"gate_batch: ops has shape {opsc.shape}, expected ({shape[0]},)")
    e0 = lib().eoc_global_engine()
    if e0 and shape[1] != lib().eoc_engine_params(e0).contents.n + 1:
        raise EocError(f"gate_batch: rows have {shape[1]} words, the engine's samples have n + 1 = "
                       f"{lib().eoc_engine_params(e0).contents.n + 1}")
    if out is None:
        out = np.empty(shape, np.int32)
    elif out.shape != shape or out.dtype != np.int32 or not out.flags.c_contiguous:
        raise EocError("gate_batch: out must be a C-contiguous int32 array of the operands' shape")
    _check(lib().eoc_gate_batch(int(op), None if opsc is None else opsc.ctypes.data,
                                None if in0 is None else in0.ctypes.data,
                                None if in1c is None else in1c.ctypes.data,
                                None if in2c is None else in2c.ctypes.data, out.ctypes.data, shape[0]),
           "eoc_gate_batch")
    return out


def circuit_run(gates, wires, instances):
    wires = np.ascontiguousarray(wires, np.int32)
    arr = (Gate * len(gates))(*gates)
    _check(lib().eoc_circuit_run(C.addressof(arr), len(gates), wires.ctypes.data, wires.shape[0], instances),
           "eoc_circuit_run")
    return wires


# ---- raw-buffer calls on the string API's global key (what the Node / Lua batch wrappers use) ------------------
def global_key_mode():
    return lib().eoc_global_key_mode()


def global_params():
    p = Params()
    _check(lib().eoc_global_params(C.byref(p)), "eoc_global_params")
    return p


def global_import_cloud_key_blob(blob):
    """Server side: install an EOCCK1 blob (numpy uint8 / bytes) as the cloud-key-only global context."""
    blob = np.ascontiguousarray(np.frombuffer(blob, np.uint8) if isinstance(blob, (bytes, bytearray)) else blob, np.uint8)
    _check(lib().eoc_global_import_cloud_key_blob(blob.ctypes.data, blob.size), "eoc_global_import_cloud_key_blob")


def global_cloud_key_export():
    need = lib().eoc_global_cloud_key_export(None, 0)
    if not need:
        raise EocError("eoc_global_cloud_key_export: no key on the global context")
    buf = np.empty(need, np.uint8)
    if lib().eoc_global_cloud_key_export(buf.ctypes.data, need) != need:
        raise EocError("eoc_global_cloud_key_export failed")
    return buf


def global_encrypt_bits(bits):
    bits = np.ascontiguousarray(np.asarray(bits).ravel(), np.uint8)
    out = np.empty((bits.size, global_params().n + 1), np.int32)
    _check(lib().eoc_global_encrypt_bits(bits.ctypes.data, bits.size, out.ctypes.data), "eoc_global_encrypt_bits")
    return out


def global_decrypt_bits(cts):
    cts = np.ascontiguousarray(cts, np.int32).reshape(-1, global_params().n + 1)
    out = np.empty(cts.shape[0], np.uint8)
    _check(lib().eoc_global_decrypt_bits(cts.ctypes.data, cts.shape[0], out.ctypes.data), "eoc_global_decrypt_bits")
    return out


def global_gate_batch(op, in0, in1=None, in2=None, ops=None):
    """eoc_global_gate_batch: brings the GPU engine(s) up behind the global key on first use."""
    in0 = np.ascontiguousarray(in0, np.int32)
    in1c = None if in1 is None else np.ascontiguousarray(in1, np.int32)
    in2c = None if in2 is None else np.ascontiguousarray(in2, np.int32)
    opsc = None if ops is None else np.ascontiguousarray(ops, np.uint8)
    for a in (in1c, in2c):
        if a is not None and a.shape != in0.shape:
            raise EocError("global_gate_batch: operand shapes differ")
    out = np.empty_like(in0)
    _check(lib().eoc_global_gate_batch(int(op), None if opsc is None else opsc.ctypes.data, in0.ctypes.data,
                                       None if in1c is None else in1c.ctypes.data,
                                       None if in2c is None else in2c.ctypes.data, out.ctypes.data, in0.shape[0]),
           "eoc_global_gate_batch")
    return out


def global_circuit_run(gates, wires, instances):
    wires = np.ascontiguousarray(wires, np.int32)
    arr = (Gate * len(gates))(*gates)
    _check(lib().eoc_global_circuit_run(C.addressof(arr), len(gates), wires.ctypes.data, wires.shape[0], instances),
           "eoc_global_circuit_run")
    return wires


class Tfhe:
    """String façade in the shape of ao-tfhe/tfhe.lua (Tfhe.* -> backend.*), for the Boolean path."""

    # ---- the reference's 11 functions (ao-tfhe/tfhe.lua:4-53), same names and argument order ----
    @staticmethod
    def info():
        return lib().info()

    @staticmethod
    def testJWT():
        return lib().testJWT()

    @staticmethod
    def generateSecretKey(jwtToken, jwksBase64):
        return _take_str(lib().generateSecretKey(jwtToken.encode(), jwksBase64.encode()))

    @staticmethod
    def generatePublicKey():
        return _take_str(lib().generatePublicKey())

    @staticmethod
    def encryptInteger(value, key=""):
        return _take_str(lib().encryptInteger(int(value), key.encode()))

    @staticmethod
    def encryptInteger_dummy(value, key=""):
        return _take_str(lib().encryptInteger_dummy(int(value), key.encode()))

    @staticmethod
    def decryptInteger(value, key, jwtToken, jwksBase64):
        return lib().decryptInteger(value.encode(), key.encode(), jwtToken.encode(), jwksBase64.encode())

    @staticmethod
    def addCiphertexts(c1, c2, public_key=""):
        return _take_str(lib().addCiphertexts(c1.encode(), c2.encode(), public_key.encode()))

    @staticmethod
    def subtractCiphertexts(c1, c2, public_key=""):
        # ao-tfhe/tfhe.lua:41-43 forwards subtract to backend.addCiphertexts; tests/tfhe.test.js:185
        # pins the result (50 "-" 8 = 58).  Kept for drop-in behaviour; the real one is below.
        return _take_str(lib().addCiphertexts(c1.encode(), c2.encode(), public_key.encode()))

    @staticmethod
    def subtractCiphertexts_backend(c1, c2, public_key=""):
        return _take_str(lib().subtractCiphertexts(c1.encode(), c2.encode(), public_key.encode()))

    @staticmethod
    def encryptASCIIString(value, length, key=""):
        return _take_str(lib().encrypt8BitASCIIString(value.encode(), int(length), key.encode()))

    @staticmethod
    def decryptASCIIString(value, length, key, jwtToken, jwksBase64):
        return _take_str(lib().decrypt8BitASCIIString(value.encode(), int(length), key.encode(),
                                                      jwtToken.encode(), jwksBase64.encode()))

    @staticmethod
    def exportSecretKey():
        return _take_str(lib().exportSecretKey())

    @staticmethod
    def importSecretKey(b64):
        return lib().importSecretKey(b64.encode())

    # ---- the cloud ("public") key: exported by the client, imported by a server that never sees the secret key ----
    @staticmethod
    def exportCloudKey():
        return _take_str(lib().exportCloudKey())

    @staticmethod
    def importCloudKey(b64):
        return lib().importCloudKey(b64.encode())

    @staticmethod
    def exportCloudKeyToFile(path):
        return lib().exportCloudKeyToFile(os.fsencode(path))

    @staticmethod
    def importCloudKeyFromFile(path):
        return lib().importCloudKeyFromFile(os.fsencode(path))

    @staticmethod
    def keyMode():
        """0 no key, 1 secret + cloud key (client / single host), 2 cloud key only (server)"""
        return lib().eoc_global_key_mode()

    # ---- Boolean path -----------------------------------------------------------------------------
    @staticmethod
    def generateGateKey(minimum_lambda=80, seed=1):
        return _take_str(lib().generateGateKey(minimum_lambda, seed))

    @staticmethod
    def resetGateKey():
        lib().resetGateKey()

    @staticmethod
    def encryptBit(bit, key=""):
        return _take_str(lib().encryptBit(int(bit), key.encode()))

    @staticmethod
    def constantBit(bit):
        """bootsCONSTANT: the noiseless trivial sample of `bit` as a ciphertext string"""
        return _take_str(lib().constantBit(int(bit)))

    @staticmethod
    def decryptBit(ct, key=""):
        return lib().decryptBit(ct.encode(), key.encode())

    @staticmethod
    def _g2(name, a, b, pk=""):
        return _take_str(getattr(lib(), name)(a.encode(), b.encode(), pk.encode()))

    nand = staticmethod(lambda a, b, pk="": Tfhe._g2("gateNAND", a, b, pk))
    and_ = staticmethod(lambda a, b, pk="": Tfhe._g2("gateAND", a, b, pk))
    or_ = staticmethod(lambda a, b, pk="": Tfhe._g2("gateOR", a, b, pk))
    nor = staticmethod(lambda a, b, pk="": Tfhe._g2("gateNOR", a, b, pk))
    xor = staticmethod(lambda a, b, pk="": Tfhe._g2("gateXOR", a, b, pk))
    xnor = staticmethod(lambda a, b, pk="": Tfhe._g2("gateXNOR", a, b, pk))

    @staticmethod
    def not_(a, pk=""):
        return _take_str(lib().gateNOT(a.encode(), pk.encode()))

    @staticmethod
    def mux(a, b, c, pk=""):
        return _take_str(lib().gateMUX(a.encode(), b.encode(), c.encode(), pk.encode()))

    @staticmethod
    def maj(a, b, c, pk=""):
        """extension gate: the majority of three bits in ONE bootstrap (a full adder's carry)"""
        return _take_str(lib().gateMAJ(a.encode(), b.encode(), c.encode(), pk.encode()))

    @staticmethod
    def xor3(a, b, c, pk=""):
        """extension gate: the parity of three bits in ONE bootstrap (a full adder's sum)"""
        return _take_str(lib().gateXOR3(a.encode(), b.encode(), c.encode(), pk.encode()))

    # ---- word-level circuits, one backend call each; the FORM is picked by the instance count (circuits.pick_form: ----
    # ---- fewest levels for a handful of instances, fewest bootstraps for thousands), like Tfhe.addBits in tfhe.js / .lua ----
    @staticmethod
    def _samples(cts):
        import base64
        n1 = global_params().n + 1
        return np.stack([np.frombuffer(base64.b64decode(c), "<i4", count=n1) for c in cts])[:, None, :]

    @staticmethod
    def _strings(planes):
        import base64
        return [base64.b64encode(np.ascontiguousarray(p[0], "<i4").tobytes() + bytes(8)).decode() for p in planes]

    @staticmethod
    def _run(built, a_planes, b_planes, instances, outputs):
        gates, n_wires, a, b = built[0], built[1], built[2], built[3]
        wires = np.zeros((n_wires, instances, a_planes.shape[-1]), np.int32)
        wires[a[0]: a[0] + len(a)] = a_planes
        wires[b[0]: b[0] + len(b)] = b_planes
        return global_circuit_run(gates, wires, instances)[list(outputs)]

    @staticmethod
    def addBitsBatch(A, B):
        """A, B: samples [nbits][instances][n+1] (LSB first) -> [nbits + 1][instances][n+1]"""
        from . import circuits
        A, B = np.ascontiguousarray(A, np.int32), np.ascontiguousarray(B, np.int32)
        built = circuits.adder(A.shape[0], A.shape[1])
        return Tfhe._run(built, A, B, A.shape[1], built[4])

    @staticmethod
    def lessThanBitsBatch(A, B):
        """A, B: samples [nbits][instances][n+1] -> [instances][n+1]: 1 iff A < B (unsigned)"""
        from . import circuits
        A, B = np.ascontiguousarray(A, np.int32), np.ascontiguousarray(B, np.int32)
        built = circuits.less_than_for(A.shape[0], A.shape[1])
        return Tfhe._run(built, A, B, A.shape[1], [built[4]])[0]

    @staticmethod
    def subtractBitsBatch(A, B):
        """A, B: samples [nbits][instances][n+1] -> [nbits + 1][instances][n+1]: difference bits, then the borrow (= A < B)"""
        from . import circuits
        A, B = np.ascontiguousarray(A, np.int32), np.ascontiguousarray(B, np.int32)
        built = circuits.subtractor_for(A.shape[0], A.shape[1])
        return Tfhe._run(built, A, B, A.shape[1], list(built[4]) + [built[5]])

    @staticmethod
    def multiplyBitsBatch(A, B):
        """A, B: samples [nbits][instances][n+1] -> [2 nbits][instances][n+1]: column compression + one prefix addition for
        small batches, the row-by-row form for wide ones (both through the netlist optimizer)"""
        from . import circuits
        A, B = np.ascontiguousarray(A, np.int32), np.ascontiguousarray(B, np.int32)
        built = circuits.multiplier_for(A.shape[0], A.shape[1])
        return Tfhe._run(built, A, B, A.shape[1], built[4])

    @staticmethod
    def minMaxBitsBatch(A, B):
        """A, B: samples [nbits][instances][n+1] -> (min, max), each [nbits][instances][n+1]"""
        from . import circuits
        A, B = np.ascontiguousarray(A, np.int32), np.ascontiguousarray(B, np.int32)
        built = circuits.min_max_for(A.shape[0], A.shape[1])
        out = Tfhe._run(built, A, B, A.shape[1], list(built[4]) + list(built[5]))
        return out[: A.shape[0]], out[A.shape[0]:]

    @staticmethod
    def addBits(A, B):
        """arrays of base64 bit ciphertexts (LSB first) -> len(A) + 1 ciphertext strings (the log-depth adder)"""
        return Tfhe._strings(Tfhe.addBitsBatch(Tfhe._samples(A), Tfhe._samples(B)))

    @staticmethod
    def lessThanBits(A, B):
        return Tfhe._strings(Tfhe.lessThanBitsBatch(Tfhe._samples(A), Tfhe._samples(B))[None])[0]


class Circuit:
    """Deferred gates: the reference's call style (one ciphertext operation per call, ao-tfhe/tfhe.lua:4-53) evaluated by ONE
    backend call.  A gate call on the string API costs a whole blind rotation's n sequential steps (1.8 ms) however little
    it computes; a Circuit records the same calls on wire handles and run() sends the recorded netlist through
    eoc_netlist_optimize and one eoc_global_circuit_run, where every LEVEL costs those 1.8 ms.  The twin of Tfhe.newCircuit
    in tfhe_gates.lua / tfhe.js.

        c = Circuit(); x, y = c.input(ct_x), c.input(ct_y)
        s, k = c.run([c.xor(x, y), c.and_(x, y)])          # base64 ciphertext strings, one backend call
    """

    def __init__(self):
        self.gates, self.n_wires, self.inputs, self.instances = [], 0, {}, None

    def _wire(self):
        self.n_wires += 1
        return self.n_wires - 1

    def _gate(self, name, a=-1, b=-1, c=-1):
        out = self._wire()
        self.gates.append(Gate(OPS[name], a, b, c, out))
        return out

    def _add_input(self, planes):
        if self.instances not in (None, planes.shape[0]):
            raise EocError("every input of a Circuit has the same instance count")
        self.instances = planes.shape[0]
        w = self._wire()
        self.inputs[w] = planes
        return w

    def input(self, ct):
        """one base64 ciphertext string"""
        return self._add_input(Tfhe._samples([ct])[0])

    def input_samples(self, samples):
        """raw samples [instances][n+1]"""
        return self._add_input(np.ascontiguousarray(samples, np.int32))

    def constant(self, bit):
        return self._gate("CONST1" if bit else "CONST0")

    def nand(self, a, b): return self._gate("NAND", a, b)
    def and_(self, a, b): return self._gate("AND", a, b)
    def or_(self, a, b): return self._gate("OR", a, b)
    def nor(self, a, b): return self._gate("NOR", a, b)
    def xor(self, a, b): return self._gate("XOR", a, b)
    def xnor(self, a, b): return self._gate("XNOR", a, b)
    def not_(self, a): return self._gate("NOT", a)
    def mux(self, a, b, c): return self._gate("MUX", a, b, c)
    def maj(self, a, b, c): return self._gate("MAJ", a, b, c)
    def xor3(self, a, b, c): return self._gate("XOR3", a, b, c)

    def run(self, outs):
        """outs: handles -> list of base64 strings (one instance) or of sample arrays [instances][n+1]"""
        instances = self.instances or 1
        rowlen = next(iter(self.inputs.values())).shape[-1] if self.inputs else global_params().n + 1
        wires = np.zeros((self.n_wires, instances, rowlen), np.int32)
        for w, planes in self.inputs.items():
            wires[w] = planes
        gates = netlist_optimize(self.gates, list(outs)) if self.gates else []
        if gates:
            global_circuit_run(gates, wires, instances)
        return Tfhe._strings([wires[w] for w in outs]) if instances == 1 else [wires[w] for w in outs]

