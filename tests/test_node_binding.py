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
@pytest.mark.gpu
def test_gates_through_node_gpu(built_lib):
    _build()
    r = subprocess.run(["node", os.path.join(NODE_DIR, "test_gpu.js")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "node gpu tests OK" in r.stdout
