"""The Lua binding is compiled and executed against a test double of the Lua C API (tests/test_lua_binding.py), never
against liblua (no Lua SDK in the image); the Node addon is built against the real N-API and GPU-tested.  This test keeps
the two surfaces -- and the Lua facade text (executed by tests/test_lua_facade.py through the minilua test double, never
by a Lua VM) -- mechanically in step:

  * the luaL_Reg entries integration/lua/eoc-tfhe-gate-bindings.c appends to luaopen_tfhe's table
    (/root/reference/ao-tfhe/eoc-tfhe-bindings.c:128-148 holds the reference's eleven) == the Node addon's exports,
    up to a LISTED set of Node-only entries;
  * the `Tfhe.*` functions integration/lua/tfhe_gates.lua appends to ao-tfhe/tfhe.lua (:4-53) == tfhe.js's methods,
    up to the keyword renames and a LISTED pair of Lua-only base64 helpers;
  * every backend function either facade calls is registered by its binding; every registered l_* has a definition;
  * every C-ABI symbol the Lua binding calls is declared in include/eoc_tfhe_gpu.h.
Pure text processing: no GPU, no Lua, no Node.
"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LUA_C = os.path.join(ROOT, "integration", "lua", "eoc-tfhe-gate-bindings.c")
LUA_F = os.path.join(ROOT, "integration", "lua", "tfhe_gates.lua")
NODE_C = os.path.join(ROOT, "integration", "node", "eoc_tfhe_node.c")
NODE_F = os.path.join(ROOT, "integration", "node", "tfhe.js")

# the table of luaopen_tfhe as the reference has it (ao-tfhe/eoc-tfhe-bindings.c:130-144) and the facade functions of
# ao-tfhe/tfhe.lua:4-53 -- the text under integration/lua is APPENDED to those two files, so it does not repeat them
REF_REG = {"addCiphertexts", "subtractCiphertexts", "generateSecretKey", "generatePublicKey", "encryptInteger",
           "encryptInteger_dummy", "encryptASCIIString", "decryptInteger", "decryptASCIIString", "info", "testJWT"}
REF_FACADE = {"info", "testJWT", "generateSecretKey", "generatePublicKey", "encryptInteger", "encryptInteger_dummy",
              "decryptInteger", "addCiphertexts", "subtractCiphertexts", "encryptASCIIString", "decryptASCIIString"}
# asynchronous batches need pinned, mutable, caller-kept host buffers: a Node Buffer can be one, a Lua string cannot
NODE_ONLY_EXPORTS = {"hostAlloc", "gateBatchSubmit", "gateBatchWait"}
# string-level circuits decode base64 ciphertext strings into samples: the Lua facade carries its own base64 (no such thing
# in the Lua 5.3 standard library) and exposes it; Node has Buffer
JS_ONLY_FACADE = set()
LUA_ONLY_FACADE = {"base64Decode", "base64Encode"}
JS_STRUCTURAL = {"backend", "Netlist"}             # Tfhe.backend exists in ao-tfhe/tfhe.lua:2; Netlist is a JS class
KEYWORD_RENAMES = {"and": "band", "or": "bor", "not": "bnot"}


def read(path):
    return open(path).read()


def strip_c_comments(text):
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def lua_registered():
    return set(re.findall(r'\{"([A-Za-z_0-9]+)",\s*l_[A-Za-z_0-9]+\}', strip_c_comments(read(LUA_C))))


def node_exported():
    return set(re.findall(r'\{"([A-Za-z_0-9]+)",\s*n_[A-Za-z_0-9]+\}', strip_c_comments(read(NODE_C))))


def test_lua_registry_equals_node_exports():
    lua, node = lua_registered(), node_exported()
    assert not (lua & REF_REG), "the appended entries must not repeat the reference's own eleven"
    assert REF_REG <= node, REF_REG - node
    assert NODE_ONLY_EXPORTS <= node and not (NODE_ONLY_EXPORTS & lua)
    assert lua | REF_REG | NODE_ONLY_EXPORTS == node, (sorted(node - lua - REF_REG - NODE_ONLY_EXPORTS),
                                                       sorted(lua - node))


def test_every_registered_lua_function_is_defined_once():
    text = strip_c_comments(read(LUA_C))
    defined = re.findall(r"static int (l_[A-Za-z_0-9]+)\(lua_State \*L\)", text)
    defined += ["l_" + m for m in re.findall(r"EOC_GATE2\((gate[A-Z]+)\)", text)]
    registered = re.findall(r'\{"[A-Za-z_0-9]+",\s*(l_[A-Za-z_0-9]+)\}', text)
    assert sorted(defined) == sorted(set(defined)), "duplicate definition"
    assert set(registered) == set(defined), (set(registered) ^ set(defined))
    for name, fn in re.findall(r'\{"([A-Za-z_0-9]+)",\s*l_([A-Za-z_0-9]+)\}', text):
        assert name == fn, (name, fn)                              # {"x", l_x}: the reference's naming (:130-144)
    assert text.count("{") == text.count("}") and text.count("(") == text.count(")")


def test_lua_binding_calls_only_declared_abi_symbols():
    import eoc_tfhe_amd
    declared = set(eoc_tfhe_amd.abi_symbols())
    text = strip_c_comments(read(LUA_C))
    text = re.sub(r'"(?:\\.|[^"\\])*"', '""', text)                # string literals out of the way
    called = set(re.findall(r"\b([A-Za-z_][A-Za-z_0-9]*)\s*\(", text))
    host_api = {c for c in called if c.startswith(("lua_", "luaL_", "l_", "luaopen_"))}
    libc_and_syntax = {"malloc", "free", "memcpy", "sizeof", "if", "for", "return", "EOC_GATE2", "NAME", "defined"}
    ours = called - host_api - libc_and_syntax
    assert ours, "no C-ABI call found: the parser is broken"
    assert ours <= declared, sorted(ours - declared)
    # and it uses a real share of the surface (a renamed symbol would drop out of `ours` silently otherwise)
    for must in ("generateGateKey", "gateMUX", "importCloudKey", "exportCloudKey", "eoc_global_gate_batch",
                 "eoc_global_circuit_run", "eoc_netlist_optimize", "eoc_gpu_set_devices", "eoc_global_key_mode"):
        assert must in ours, must


def lua_facade():
    text = re.sub(r"--.*", "", read(LUA_F))
    return set(re.findall(r"^function Tfhe\.([A-Za-z_0-9]+)\s*\(", text, flags=re.M)) | \
        set(re.findall(r"^Tfhe\.([A-Za-z_0-9]+)\s*=", text, flags=re.M))


def js_facade():
    text = re.sub(r"//.*", "", read(NODE_F))
    return set(re.findall(r"^Tfhe\.([A-Za-z_0-9]+)\s*=", text, flags=re.M))


def test_lua_facade_equals_js_facade():
    lua, js = lua_facade(), js_facade()
    assert REF_FACADE <= js and not (REF_FACADE & lua)             # the reference's own functions live in tfhe.lua itself
    js_as_lua = {KEYWORD_RENAMES.get(n, n) for n in js - REF_FACADE - JS_STRUCTURAL - JS_ONLY_FACADE}
    assert JS_ONLY_FACADE <= js and not (JS_ONLY_FACADE & lua)
    assert LUA_ONLY_FACADE <= lua and not (LUA_ONLY_FACADE & js)
    assert js_as_lua == lua - LUA_ONLY_FACADE, (sorted(js_as_lua - lua), sorted(lua - LUA_ONLY_FACADE - js_as_lua))


def test_facades_call_only_registered_backend_functions():
    lua_text = re.sub(r"--.*", "", read(LUA_F))
    lua_calls = set(re.findall(r"Tfhe\.backend\.([A-Za-z_0-9]+)", lua_text))
    assert lua_calls <= lua_registered() | REF_REG, sorted(lua_calls - lua_registered() - REF_REG)
    js_text = re.sub(r"//.*", "", read(NODE_F))
    js_calls = set(re.findall(r"\bB\.([A-Za-z_0-9]+)\s*\(", js_text))
    assert js_calls <= node_exported(), sorted(js_calls - node_exported())
    # the same backend entry behind the same facade name on both sides (pass-throughs only)
    lua_map = dict(re.findall(r"^function Tfhe\.([A-Za-z_0-9]+)\([^)]*\)\s+return Tfhe\.backend\.([A-Za-z_0-9]+)\(", lua_text,
                              flags=re.M))
    js_map = dict(re.findall(r"^Tfhe\.([A-Za-z_0-9]+) = (?:\([^)]*\)|[A-Za-z_]+) => B\.([A-Za-z_0-9]+)\(", js_text, flags=re.M))
    common = {KEYWORD_RENAMES.get(k, k): v for k, v in js_map.items() if KEYWORD_RENAMES.get(k, k) in lua_map}
    assert len(common) >= 20
    for name, backend in common.items():
        assert lua_map[name] == backend, (name, lua_map[name], backend)


def test_lua_facade_text_is_balanced():
    """the cheapest syntax check available without an interpreter: block openers and `end`s pair up"""
    text = re.sub(r"--.*", "", read(LUA_F))
    text = re.sub(r'"(?:\\.|[^"\\])*"', '""', text)
    words = re.findall(r"\b(function|if|for|while|do|end|repeat|until)\b", text)
    depth, openers = 0, 0
    pending_do = 0                                                  # `for ... do` / `while ... do` open ONE block
    for w in words:
        if w in ("for", "while"):
            pending_do += 1
        elif w == "do":
            if pending_do:
                pending_do -= 1
                depth += 1
            else:
                depth += 1
            openers += 1
        elif w in ("function", "if"):
            depth += 1
            openers += 1
        elif w == "end":
            depth -= 1
            assert depth >= 0
    assert depth == 0 and pending_do == 0 and openers > 40
    assert text.count("(") == text.count(")") and text.count("{") == text.count("}")
