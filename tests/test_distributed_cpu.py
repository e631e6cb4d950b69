"""N > 1 path on CPU: two gloo ranks, key broadcast + contiguous sharding + gather."""
import json
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("total", [11])
def test_two_rank_gloo_sharded_gates(tmp_path, built_lib, oracle_mod, total):
    out = tmp_path / "res.json"
    env = dict(os.environ, OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(HERE, "dist_worker.py"), str(out), str(total)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.load(open(out))
    assert res["ok"] and res["world"] == 2
    assert res["blocks"] == [[0, 6], [6, 11]]
