"""N > 1 path on the HIP engine: two ranks (gloo control plane) sharing GPU 0 run the key replication and their
shard on their own engine; the gathered result equals the oracle bit for bit."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_two_ranks_on_the_hip_engine(tmp_path, built_lib, oracle_mod):
    out = tmp_path / "res.json"
    env = dict(os.environ, OMP_NUM_THREADS="4", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(HERE, "dist_gpu_worker.py"), str(out), "11"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.load(open(out))
    assert res["ok"] and res["world"] == 2
    assert res["blocks"] == [[0, 6], [6, 11]] and res["bootstraps_rank0"] == 6
