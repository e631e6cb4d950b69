"""The Node.js host binding (integration/node): N-API addon over the C ABI + tfhe.js, the JS twin of
ao-tfhe/tfhe.lua.  CPU leg = the reference's six tests (tests/tfhe.test.js) in the reference's host language."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NODE_DIR = os.path.join(ROOT, "integration", "node")

needs_node = pytest.mark.skipif(shutil.which("node") is None or not os.path.exists("/usr/include/node/node_api.h"),
                                reason="node or N-API headers missing")


def _build():
    subprocess.check_call(["bash", os.path.join(NODE_DIR, "build.sh")], stdout=subprocess.DEVNULL)


@needs_node
def test_reference_tests_in_node_cpu(built_lib):
    _build()
    r = subprocess.run(["node", os.path.join(NODE_DIR, "test_cpu.js")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "node cpu tests OK" in r.stdout
    assert "TFHE Library: Enabling fully homomorphic encryption" in r.stdout


@needs_node
def test_netlist_forms_in_node_cpu(built_lib):
    """round 6: tfhe.js's circuit layer on plaintext -- every adder / comparator form against integer arithmetic, the
    bootstrap counts and depths eoc_tfhe_amd/circuits.py states, the optimizer's carry rewrite through the addon, and the
    form picked by instance count (netlistCost / netlistDepth)"""
    _build()
    r = subprocess.run(["node", os.path.join(NODE_DIR, "test_netlists_cpu.js")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "node netlist cpu tests OK" in r.stdout


@needs_node
def test_cloud_key_client_server_in_node_cpu(built_lib, tmp_path):
    """f2 through the N-API addon: a client process exports the cloud key, a second process installs ONLY that key
    (keyMode 2: no encrypt / decrypt / secret export), adds two ciphertexts, and the client decrypts the sum"""
    _build()
    for mode in ("client", "server", "verify"):
        r = subprocess.run(["node", os.path.join(NODE_DIR, "test_cpu_cloud.js"), mode, str(tmp_path)],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
        assert "node cloud %s OK" % mode in r.stdout


@needs_node
@pytest.mark.gpu
def test_cloud_key_server_in_node_gpu(built_lib, tmp_path):
    """the same three Node processes on the GPU box: the cloud-key-only server evaluates a string-API gate and a raw
    XOR batch on the GPU, the client decrypts them"""
    _build()
    for mode in ("client", "server", "verify"):
        r = subprocess.run(["node", os.path.join(NODE_DIR, "test_cpu_cloud.js"), mode, str(tmp_path)],
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "node cloud gpu leg verified" in r.stdout


@needs_node
@pytest.mark.gpu
def test_gates_through_node_gpu(built_lib):
    _build()
    r = subprocess.run(["node", os.path.join(NODE_DIR, "test_gpu.js")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "node gpu tests OK" in r.stdout


@needs_node
@pytest.mark.gpu
def test_gate_batch_through_node_equals_golden_bytes(built_lib, tmp_path):
    """the binding's gateBatch compared with oracle BITS, not by decryption -- the committed golden inputs
    (Set A, key seed 1) through the N-API addon must come back as the committed golden outputs"""
    import numpy as np
    _build()
    z = np.load(os.path.join(ROOT, "tests", "golden", "golden_arrays.npz"))
    for name in ("A_c0", "A_c1", "A_c2", "A_NAND_out", "A_MUX_out"):
        np.ascontiguousarray(z[name], "<i4").tofile(str(tmp_path / (name + ".bin")))
    env = dict(os.environ, EOC_GOLDEN_DIR=str(tmp_path))
    r = subprocess.run(["node", os.path.join(NODE_DIR, "test_gpu_golden.js")], capture_output=True, text=True,
                       timeout=900, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "node gpu golden OK" in r.stdout
