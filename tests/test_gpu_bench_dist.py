"""bench.py's N > 1 code under test on ONE GPU.

The first time the driver gets an 8-GPU node, `python -m torch.distributed.run ... bench.py --gpus N` is what it runs;
until this round nothing in the suite executed that code (make_config3_block / make_config4_block, reduce_max /
reduce_all_ok on tensors, the process-group bring-up, the multi-leg loop).  Two rehearsals that a one-GPU box CAN run:
  * two ranks, gloo control plane, both on GPU 0: the whole N = 2 flow incl. key replication and both BASELINE legs;
  * one rank, backend nccl (= RCCL), EOC_BENCH_FORCE_DIST=1: init_process_group("nccl", device_id=...), the broadcast
    of the key images and the device-tensor all-reduces execute through RCCL itself.
Neither is a scaling measurement -- none can be made without the node (README: "never executed on hardware").
"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run_bench(nproc, extra, env_extra=None):
    env = dict(os.environ, OMP_NUM_THREADS="4", HSA_ENABLE_IPC_MODE_LEGACY="0", EOC_BENCH_NO_INLIB="1")
    env.update(env_extra or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--gates", "128", "--steps", "2", "--warmup", "1",
           "--config3-total", "8192"] + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:] + "\n---\n" + r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                      # stdout carries exactly ONE JSON line (rank 0)
    return json.loads(lines[0])


def _check_line(res, world):
    assert res["n_gpus"] == world and res["steps"] == 2 and res["warmup"] == 1
    assert res["metric"].startswith("gate bootstraps/sec") and res["unit"] == "gate bootstraps/s"
    assert res["scaling"] == "weak" and res["higher_is_better"] is True and res["vs_baseline"] is None
    assert res["decrypt_ok"] is True and res["paths_bit_identical"] is True
    assert res["value"] > 0 and abs(res["value"] - 128 * world * 2 / (res["ms_per_step"] * 2e-3)) < 0.01 * res["value"]
    assert "key_broadcast_s" in res["config"] and res["config"]["key_broadcast_s"] >= 0
    assert res["roofline"]["kernel"] == "k_blind_rotate" and res["roofline"]["avg_launch_ms"] > 0
    sec = res["secondary"]
    for leg, boots in (("config3_mixed_1M", None), ("config4_streq_1024x32", 511 * 1024)):
        assert sec[leg]["decrypt_ok"] is True, leg
        assert sec[leg]["n_gpus"] == world and sec[leg]["scaling"] == "strong" and sec[leg]["bootstraps_per_s"] > 0
        if boots:
            assert sec[leg]["bootstraps"] == boots
    assert 8192 < sec["config3_mixed_1M"]["bootstraps"] < 2 * 8192  # MUX counts twice


def test_bench_two_ranks_gloo_share_one_gpu(built_lib):
    res = _run_bench(2, ["--dist-backend", "gloo"])
    _check_line(res, 2)


def test_bench_one_rank_through_rccl(built_lib):
    res = _run_bench(1, ["--dist-backend", "nccl"], {"EOC_BENCH_FORCE_DIST": "1"})
    _check_line(res, 1)
    assert "cpu_baseline" in res                                    # rank 0 at N = 1 still times the oracle beside it
