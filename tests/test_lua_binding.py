"""The Lua binding (integration/lua/eoc-tfhe-gate-bindings.c; SURVEY.md 8 f4, the reference's only host:
/root/reference/ao-tfhe/eoc-tfhe-bindings.c:128-148) COMPILED AND EXECUTED -- against a test double of the Lua 5.3 C API
(tests/lua_double/: a value stack with the dozen calls the binding uses), because the build image has no Lua SDK.
This tests THIS REPOSITORY'S C (argument marshalling, length arithmetic, NULL -> nil, heap-result ownership) on top of the
real library; it does not test Lua, and the Lua facade text (integration/lua/tfhe_gates.lua) still has no interpreter
(tests/test_binding_surfaces.py keeps it in step with the executed JS twin).

  * CPU: tests/c/lua_binding_driver.c drives all 34 entries (client-side calls, cloud-key export / import through strings
    and files, netlist helpers, every refusal and every luaL_check* error) -- plain build and ASan/UBSan build;
  * GPU: the same double driven from Python: l_generateGateKey, l_encryptBits, l_gateBatch (NAND, MUX, mixed opcodes) and
    l_circuitRun against the CPU oracle bit for bit, wrong-length and nil operands refused with nil.
"""
import base64
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DBL = os.path.join(ROOT, "tests", "lua_double")
LIBDIR = os.path.join(ROOT, "eoc_tfhe_amd")
SRCS = [os.path.join(ROOT, "integration", "lua", "eoc-tfhe-gate-bindings.c"), os.path.join(DBL, "lua_double.c")]
INC = ["-I" + DBL, "-I" + os.path.join(ROOT, "include")]
LINK = ["-L" + LIBDIR, "-leoc_tfhe_gpu", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"]


@pytest.fixture(scope="module")
def key_file(built_lib, tmp_path_factory):
    """base64(EOCSK1) of a seeded Set-A-shaped key with n = 16 (keygen in milliseconds)"""
    import eoc_tfhe_amd as eoc
    p = eoc.default_params(0)
    p.n = 16
    d = tmp_path_factory.mktemp("luakey")
    path = d / "secret.b64"
    path.write_bytes(base64.b64encode(eoc.SecretKey(p, 77).export_bytes()))
    return str(path), str(d)


def _driver(tmp_path, extra, cc="gcc"):
    exe = str(tmp_path / "lua_binding_driver")
    subprocess.check_call([cc, "-std=c11", "-Wall", "-Werror", *extra, *INC, os.path.join(ROOT, "tests", "c", "lua_binding_driver.c"),
                           *SRCS, "-o", exe, *LINK])
    return exe


def test_lua_binding_compiles_and_runs_cpu_legs(tmp_path, key_file):
    exe = _driver(tmp_path, ["-O1"])
    r = subprocess.run([exe, *key_file], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "lua_binding_driver OK: 34 entries" in r.stdout, (r.stdout + r.stderr)[-3000:]
    # the reference's stderr conventions come through the binding (eoc-tfhe-run.cpp:277-278, :465-468)
    assert "Secret key not initialized" in r.stderr


def test_lua_binding_cpu_legs_under_asan_ubsan(tmp_path, key_file):
    """binding + double + driver instrumented (gcc -fsanitize=address,undefined); leak check on: every heap result the
    library hands out must be freed by the binding exactly once, as ao-tfhe/eoc-tfhe-bindings.c:21 does"""
    exe = _driver(tmp_path, ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all"])
    supp = tmp_path / "lsan.supp"
    supp.write_text("leak:libgomp\nleak:libomp\nleak:__kmp\nleak:libamdhip64\nleak:libhsa-runtime64\n")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:protect_shadow_gap=0", LSAN_OPTIONS=f"suppressions={supp}",
               OMP_NUM_THREADS="2")
    r = subprocess.run([exe, *key_file], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "lua_binding_driver OK" in r.stdout, (r.stdout + r.stderr)[-4000:]


# ---- the same double, driven from Python (GPU legs) -------------------------------------------------------------------
class Lua:
    """ctypes view of the test double + the compiled binding (one shared object)"""
    TNIL, TNUM, TSTR, TTAB = 0, 3, 4, 5

    def __init__(self, tmp_path):
        so = str(tmp_path / "libluabinding_double.so")
        subprocess.check_call(["gcc", "-std=c11", "-O1", "-Wall", "-Werror", "-shared", "-fPIC", *INC, *SRCS, "-o", so, *LINK])
        L = self.lib = C.CDLL(so)
        L.ld_new.restype = C.c_void_p
        for name, args, res in (("ld_close", [C.c_void_p], None), ("ld_settop0", [C.c_void_p], None),
                                ("ld_push_nil", [C.c_void_p], None), ("ld_push_int", [C.c_void_p, C.c_longlong], None),
                                ("ld_push_lstr", [C.c_void_p, C.c_char_p, C.c_size_t], None),
                                ("ld_call", [C.c_void_p, C.c_void_p, C.c_int], C.c_int),
                                ("ld_error", [C.c_void_p], C.c_char_p), ("ld_type", [C.c_void_p, C.c_int], C.c_int),
                                ("ld_to_int", [C.c_void_p, C.c_int], C.c_longlong),
                                ("ld_to_lstr", [C.c_void_p, C.c_int, C.POINTER(C.c_size_t)], C.c_void_p),
                                ("ld_table_get", [C.c_void_p, C.c_int, C.c_char_p], C.c_void_p),
                                ("luaopen_tfhe_gates", [C.c_void_p], C.c_int)):
            getattr(L, name).argtypes = args
            getattr(L, name).restype = res
        self.S = L.ld_new()
        assert L.luaopen_tfhe_gates(self.S) == 1
        self.fn = {}
        for n in ("generateGateKey", "resetGateKey", "encryptBits", "decryptBits", "gateBatch", "circuitRun", "sampleInts",
                  "keyMode", "engineCount", "gateNAND", "encryptBit", "decryptBit", "gateMUX", "gateNOT", "netlistOptimize",
                  "circuitBootstraps", "netlistCost", "netlistDepth"):
            self.fn[n] = L.ld_table_get(self.S, 1, n.encode())
            assert self.fn[n]
        L.ld_settop0(self.S)

    def call(self, name, *args):
        """args: int, bytes or None (nil) -> int, bytes, None; raises on a luaL_check* error"""
        L, S = self.lib, self.S
        for a in args:
            if a is None:
                L.ld_push_nil(S)
            elif isinstance(a, (bytes, bytearray)):
                L.ld_push_lstr(S, bytes(a), len(a))
            else:
                L.ld_push_int(S, int(a))
        n = L.ld_call(S, self.fn[name], len(args))
        if n < 0:
            raise RuntimeError(L.ld_error(S).decode())
        if n == 0:
            return None
        t = L.ld_type(S, 1)
        if t == self.TNIL:
            return None
        if t == self.TNUM:
            return int(L.ld_to_int(S, 1))
        ln = C.c_size_t()
        p = L.ld_to_lstr(S, 1, C.byref(ln))
        return C.string_at(p, ln.value)


@pytest.mark.gpu
def test_lua_binding_gate_batch_and_circuit_run_against_the_oracle(tmp_path, built_lib):
    from gpu_util import torch_cuda
    torch_cuda()
    import eoc_tfhe_amd as eoc
    import oracle_lib as ol
    eoc.gpu_shutdown()
    eoc.Tfhe.resetGateKey()
    lua = Lua(tmp_path)
    try:
        tok = lua.call("generateGateKey", 80, 5)                 # lambda 80 -> Set A, seeded (reproducible) key
        assert tok and base64.b64decode(tok).startswith(b"EOCGATEKEY n=500 l=2")
        assert lua.call("generateGateKey", 80, 5) is None        # one key per process (eoc-tfhe-run.cpp:245-249)
        assert lua.call("keyMode") == 1 and lua.call("engineCount") == 1 and lua.call("sampleInts") == 501
        orc = ol.Oracle(0, 5)
        rng = np.random.default_rng(12)
        cnt = 24
        bits = [rng.integers(0, 2, cnt).astype(np.uint8) for _ in range(3)]
        cts = [lua.call("encryptBits", b.tobytes()) for b in bits]
        assert all(len(c) == cnt * 501 * 4 for c in cts)
        assert lua.call("decryptBits", cts[0]) == bits[0].tobytes()
        c = [np.frombuffer(x, np.int32).reshape(cnt, 501) for x in cts]
        # NAND, MUX, a mixed batch in arbitrary opcode order: the binding's bytes == the oracle's
        got = lua.call("gateBatch", ol.OPS["NAND"], cts[0], cts[1])
        assert got == orc.gate_batch(ol.OPS["NAND"], c[0], c[1]).tobytes()
        assert lua.call("decryptBits", got) == (1 - (bits[0] & bits[1])).astype(np.uint8).tobytes()
        got = lua.call("gateBatch", ol.OPS["MUX"], cts[0], cts[1], cts[2])
        assert got == orc.gate_batch(ol.OPS["MUX"], c[0], c[1], c[2]).tobytes()
        ops = rng.choice([0, 4, 10, 11, 2, 13], cnt).astype(np.uint8)
        got = lua.call("gateBatch", 0, cts[0], cts[1], cts[2], ops.tobytes())
        assert got == orc.gate_batch(0, c[0], c[1], c[2], ops=ops).tobytes()
        # refusals: nil / short / ragged operands, wrong opcode count, unknown opcode -> nil, never a crash
        assert lua.call("gateBatch", 0, cts[0], None) is None                       # NAND without its second operand
        assert lua.call("gateBatch", 0, None, cts[1]) is None
        assert lua.call("gateBatch", 0, cts[0], cts[1][:-2004]) is None
        assert lua.call("gateBatch", 0, cts[0][:-1], cts[1][:-1]) is None
        assert lua.call("gateBatch", 10, cts[0], cts[1], None) is None              # MUX without its third
        assert lua.call("gateBatch", 0, cts[0], cts[1], cts[2], ops.tobytes()[:-1]) is None
        assert lua.call("gateBatch", 99, cts[0], cts[1]) is None
        with pytest.raises(RuntimeError, match="bad argument #1"):
            lua.call("gateBatch", b"NAND", cts[0], cts[1])
        # circuitRun: a full adder over 5 instances, every written wire against the oracle
        inst = 5
        g = np.array([[4, 0, 1, -1, 3], [4, 3, 2, -1, 4], [1, 0, 1, -1, 5], [1, 3, 2, -1, 6], [2, 5, 6, -1, 7]], np.int32)
        wires = np.zeros((8, inst, 501), np.int32)
        for w in range(3):
            wires[w] = c[w][:inst]
        out = lua.call("circuitRun", g.tobytes(), wires.tobytes(), 8, inst)
        res = np.frombuffer(out, np.int32).reshape(8, inst, 501)
        want = wires.copy()
        for op, a, b, _, o in g:
            want[o] = orc.gate_batch(int(op), want[a], want[b])
        assert np.array_equal(res, want)
        s = bits[0][:inst] ^ bits[1][:inst] ^ bits[2][:inst]
        assert lua.call("decryptBits", res[4].tobytes()) == s.astype(np.uint8).tobytes()
        assert lua.call("circuitRun", g.tobytes(), wires.tobytes(), 8, inst + 1) is None    # wires do not match the shape
        assert lua.call("circuitRun", g.tobytes()[:-4], wires.tobytes(), 8, inst) is None
        bad = g.copy()
        bad[0, 4] = 8                                                                       # writes a wire that is not there
        assert lua.call("circuitRun", bad.tobytes(), wires.tobytes(), 8, inst) is None
        # string API through the binding: one gate per call, like l_addCiphertexts (:12-24)
        a, b = lua.call("encryptBit", 1), lua.call("encryptBit", 0)
        assert lua.call("decryptBit", lua.call("gateNAND", a, b)) == 1
        assert lua.call("decryptBit", lua.call("gateMUX", a, b, a)) == 0
        assert lua.call("decryptBit", lua.call("gateNOT", b)) == 1
        assert lua.call("gateNAND", a, b"AAAA") is None
    finally:
        lua.call("resetGateKey")
        lua.lib.ld_close(lua.S)
        eoc.gpu_shutdown()
