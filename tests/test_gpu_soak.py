"""The randomised parity soak inside the driver-run suite.

Round 3's only real bug -- a NULL-stream hipMemset racing launches on hipStreamNonBlocking streams when a fresh global
context grew its workspace (7 wrong host-path results in 20 124 cases, profiles/r03_soak_parity.txt) -- was found by
tools/soak_parity.py, which nothing ran automatically.  Here: a fixed-seed slice of that soak over all six kinds (round 6:
netlists through eoc_netlist_optimize -- rewritten netlist vs oracle bit for bit, outputs vs the original's plaintext), and
a targeted regression that hammers exactly the racing shape (fresh context, first call = host-buffer batch that grows
the workspace).  Everything is compared bit for bit with the CPU oracle.
"""
import os
import sys

import numpy as np
import pytest

import oracle_lib as ol
from gpu_util import torch_cuda

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def eoc(built_lib):
    torch_cuda()
    import eoc_tfhe_amd
    return eoc_tfhe_amd


def test_soak_slice_all_six_kinds(eoc):
    import soak_parity
    lines = []
    cases, bad, per_kind = soak_parity.soak(budget_s=30.0, seed=4, round_robin=True, log=lines.append)
    assert bad == 0, [ln for ln in lines if "MISMATCH" in ln]
    assert set(per_kind) == set(soak_parity.KINDS) and min(per_kind.values()) >= 5, per_kind
    assert cases >= 40


def test_fresh_contexts_whose_first_call_grows_the_workspace(eoc):
    """>= 50 FRESH global contexts; the first call of each is a host-buffer batch, i.e. the call that allocates and
    zero-fills the key-switch operand buffer (engine.hip ensure_ws) right before launching on non-blocking streams"""
    p = eoc.default_params(0)
    p.n = 24
    sk = eoc.SecretKey(p, 31)
    orc = ol.Oracle(0, 31, n_override=24)
    rng = np.random.default_rng(8)
    widths = [1, 63, 64, 65, 1023, 1024, 1025, 2049]
    bad = []
    eoc.gpu_shutdown()
    for it in range(56):
        count = widths[it % len(widths)]
        op = [eoc.OPS["NAND"], eoc.OPS["XOR"], eoc.OPS["MUX"], eoc.OPS["ORYN"]][it % 4]
        c = [sk.encrypt_bits(rng.integers(0, 2, count).astype(np.uint8), 1000 + 3 * it + k, 0) for k in range(3)]
        want = orc.gate_batch(op, c[0], c[1], c[2])
        eoc.gpu_init(p, devices=[0] * (1 + it % 3))
        try:
            eoc.upload_cloud_key(sk)
            if it % 2:                         # pinned operands: read in place by k_prepare (no H2D copy to hide the race)
                pins = [eoc.PinnedArray(c[0].shape) for _ in range(4)]
                for k in range(3):
                    pins[k].array[:] = c[k]
                got = eoc.gate_batch(op, pins[0].array, pins[1].array, pins[2].array, out=pins[3].array).copy()
                for a in pins:
                    a.free()
            else:
                got = eoc.gate_batch(op, c[0], c[1], c[2])
        finally:
            eoc.gpu_shutdown()
        if not np.array_equal(got, want):
            bad.append((it, count, op, int((got != want).any(axis=1).sum())))
    assert not bad, bad
