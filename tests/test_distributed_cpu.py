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


def test_config3_blocks_partition_the_op_stream():
    """BASELINE configs[3] at N > 1 (bench.py's `config3_mixed_1M` leg, tests/test_gpu_baseline_configs.py): for every
    world size the ranks' blocks are contiguous, cover the seed-4 op stream exactly once and agree on the total number
    of bootstraps (MUX = 2)."""
    import numpy as np
    from eoc_tfhe_amd.distributed import CONFIG3_OPCODES, config3_block, config3_ops
    total = 1 << 20
    ops = config3_ops(total)
    assert set(np.unique(ops)) == set(CONFIG3_OPCODES)
    counts = [(ops == o).sum() / total for o in CONFIG3_OPCODES]
    assert all(abs(c - 1 / 3) < 0.005 for c in counts)
    for world in (1, 2, 3, 4, 8):
        blocks = [config3_block(total, r, world) for r in range(world)]
        assert blocks[0][0] == 0 and blocks[-1][1] == total
        assert all(blocks[r][1] == blocks[r + 1][0] for r in range(world - 1))
        assert max(b[1] - b[0] for b in blocks) - min(b[1] - b[0] for b in blocks) <= 1
        assert np.array_equal(np.concatenate([b[2] for b in blocks]), ops)
        boots = {b[3] for b in blocks}
        assert boots == {int(total + (ops == 10).sum())}
        assert sum(len(b[2]) + int((b[2] == 10).sum()) for b in blocks) == boots.pop()
