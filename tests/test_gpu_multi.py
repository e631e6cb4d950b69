"""Multi-GPU inside the C ABI: ONE process, ONE global key, several engines.  A one-GPU box
rehearses the N-GPU path by listing device 0 several times (eoc_gpu_init_multi): same sharding, same replication code
(device-to-device copies instead of the RCCL broadcast), results compared with the oracle bit for bit."""
import os

import numpy as np
import pytest

import oracle_lib as ol
from gpu_util import torch_cuda

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eoc(built_lib):
    torch_cuda()
    import eoc_tfhe_amd
    return eoc_tfhe_amd


@pytest.fixture()
def ctx3(eoc):
    """global context with three engines on device 0, Set A with a short LWE dimension (fast oracle)"""
    p = eoc.default_params(0)
    p.n = 40
    sk = eoc.SecretKey(p, 9)
    eoc.gpu_shutdown()
    eoc.gpu_init(p, devices=[0, 0, 0])
    eoc.upload_cloud_key(sk)
    orc = ol.Oracle(0, 9, n_override=40)
    yield p, sk, orc
    eoc.gpu_shutdown()


def test_blocks_are_distributed_shard(eoc, ctx3):
    from eoc_tfhe_amd.distributed import shard
    for total in (0, 1, 2, 3, 11, 1024, 1 << 20):
        for world in (1, 2, 3, 8):
            assert [eoc.shard_range(total, r, world) for r in range(world)] == [shard(total, r, world) for r in range(world)]


def test_gate_batch_three_engines_ragged(eoc, ctx3):
    p, sk, orc = ctx3
    assert eoc.gpu_engine_count() == 3
    st = eoc.stats_multi()
    assert st["key_broadcast_method"] == "peer-copy" and st["key_broadcast_s"] > 0
    total = 11                                   # blocks 4 + 4 + 3
    rng = np.random.default_rng(1)
    b0, b1, b2 = (rng.integers(0, 2, total).astype(np.uint8) for _ in range(3))
    c0, c1, c2 = sk.encrypt_bits(b0, 2, 0), sk.encrypt_bits(b1, 3, 0), sk.encrypt_bits(b2, 4, 0)
    before = [e["bootstraps"] for e in eoc.stats_multi()["engines"]]
    out = eoc.gate_batch(eoc.OPS["NAND"], c0, c1)
    after = [e["bootstraps"] for e in eoc.stats_multi()["engines"]]
    assert [a - b for a, b in zip(after, before)] == [4, 4, 3]
    assert np.array_equal(sk.decrypt_bits(out), 1 - (b0 & b1))
    assert np.array_equal(out, orc.gate_batch(ol.OPS["NAND"], c0, c1))
    # MUX (two blind rotations per gate) and a mixed batch in arbitrary opcode order
    out = eoc.gate_batch(eoc.OPS["MUX"], c0, c1, c2)
    assert np.array_equal(out, orc.gate_batch(ol.OPS["MUX"], c0, c1, c2))
    ops = np.array([0, 10, 4, 4, 11, 0, 10, 2, 13, 4, 14], np.uint8)
    out = eoc.gate_batch(0, c0, c1, c2, ops=ops)
    assert np.array_equal(out, orc.gate_batch(0, c0, c1, c2, ops=ops))
    # fewer gates than engines: empty blocks are skipped
    out = eoc.gate_batch(eoc.OPS["XOR"], c0[:2], c1[:2])
    assert np.array_equal(out, orc.gate_batch(ol.OPS["XOR"], c0[:2], c1[:2]))
    # bootsCONSTANT needs no operand at all
    out = eoc.gate_batch(eoc.OPS["CONST1"], None, count=5, rowlen=p.n + 1)
    assert np.array_equal(sk.decrypt_bits(out), np.ones(5, np.uint8)) and not out[:, :-1].any()


def test_circuit_run_instances_stay_on_one_engine(eoc, ctx3):
    from eoc_tfhe_amd import circuits
    p, sk, orc = ctx3
    gates, n_wires, aw, bw, sw = circuits.ripple_carry_adder(4)
    S = 7                                        # blocks 3 + 2 + 2
    rng = np.random.default_rng(3)
    A, B = rng.integers(0, 16, S), rng.integers(0, 16, S)
    wires = np.zeros((n_wires, S, p.n + 1), np.int32)
    for i in range(4):
        wires[aw[0] + i] = sk.encrypt_bits(((A >> i) & 1).astype(np.uint8), 100 + i, 0)
        wires[bw[0] + i] = sk.encrypt_bits(((B >> i) & 1).astype(np.uint8), 200 + i, 0)
    # a wire no gate touches keeps the caller's bytes, and so do the inputs (only live-in wires travel to the device,
    # only written wires travel back)
    wires = np.concatenate([wires, np.full((1, S, p.n + 1), 0x5A5A5A5A, np.int32)])
    ref = wires.copy()
    eoc.circuit_run(gates, wires, S)
    assert np.array_equal(wires[-1], ref[-1]) and np.array_equal(wires[aw[0]: aw[0] + 4], ref[aw[0]: aw[0] + 4])
    tot = sum(sk.decrypt_bits(wires[sw[0] + i]).astype(np.int64) << i for i in range(5))
    assert np.array_equal(tot, A + B)
    # the oracle evaluates the same netlist gate by gate on the whole instance range
    for g in gates:
        i0 = ref[g.in0] if g.in0 >= 0 else None
        i1 = ref[g.in1] if g.in1 >= 0 else None
        i2 = ref[g.in2] if g.in2 >= 0 else None
        ref[g.out] = orc.gate_batch(g.op, i0, i1, i2)
    assert np.array_equal(wires[sw[0]: sw[0] + 5], ref[sw[0]: sw[0] + 5])


def test_pinned_io_and_steady_state_buffers(eoc, ctx3):
    """eoc_host_alloc buffers are read in place by the linear stage (512 rows per engine: one un-chunked launch each --
    the chunked pipeline is test_chunked_host_pipeline's); a second call of the same size neither grows the persistent
    I/O buffers nor the engines' workspaces"""
    p, sk, orc = ctx3
    total = 1536                                 # 512 per engine
    rng = np.random.default_rng(5)
    b0, b1 = rng.integers(0, 2, total).astype(np.uint8), rng.integers(0, 2, total).astype(np.uint8)
    pin = [eoc.PinnedArray((total, p.n + 1)) for _ in range(3)]
    pin[0].array[:] = sk.encrypt_bits(b0, 7, 0)
    pin[1].array[:] = sk.encrypt_bits(b1, 8, 0)
    eoc.gate_batch(eoc.OPS["AND"], pin[0].array, pin[1].array, out=pin[2].array)
    g0 = eoc.stats_multi()["host_buffer_grows"]
    w0 = [eoc.lib().eoc_engine_workspace_grows(eoc.lib().eoc_global_engine_at(i)) for i in range(3)]
    pin[2].array[:] = 0
    eoc.gate_batch(eoc.OPS["AND"], pin[0].array, pin[1].array, out=pin[2].array)
    assert eoc.stats_multi()["host_buffer_grows"] == g0
    assert [eoc.lib().eoc_engine_workspace_grows(eoc.lib().eoc_global_engine_at(i)) for i in range(3)] == w0
    assert np.array_equal(sk.decrypt_bits(pin[2].array), b0 & b1)
    sl = slice(500, 530)                         # straddles the first block boundary (512)
    assert np.array_equal(pin[2].array[sl], orc.gate_batch(ol.OPS["AND"], pin[0].array[sl].copy(), pin[1].array[sl].copy()))
    # pageable operands give the same bits
    out = eoc.gate_batch(eoc.OPS["AND"], pin[0].array.copy(), pin[1].array.copy())
    assert np.array_equal(out, pin[2].array)
    for a in pin:
        a.free()


def test_operand_shape_checks(eoc, ctx3):
    p, sk, orc = ctx3
    c = sk.encrypt_bits(np.zeros(4, np.uint8), 1, 0)
    with pytest.raises(eoc.EocError):
        eoc.gate_batch(eoc.OPS["NAND"], c, c[:3])            # short second operand
    with pytest.raises(eoc.EocError):
        eoc.gate_batch(eoc.OPS["NAND"], c[:, :-1], c[:, :-1])  # wrong row length
    with pytest.raises(eoc.EocError):
        eoc.gate_batch(0, c, c, ops=np.zeros(3, np.uint8))    # one opcode per row


def test_concurrent_host_threads_are_serialised(eoc, ctx3):
    """SURVEY 8b state/threading: one global context, calls serialised by a mutex -- four host threads issue batch and
    circuit calls at once (ctypes releases the GIL during the call) and every result equals the oracle's"""
    import threading
    from eoc_tfhe_amd import circuits
    p, sk, orc = ctx3
    rng = np.random.default_rng(5)
    gates, n_wires, aw, bw, sw = circuits.ripple_carry_adder(2)
    jobs, results, errors = [], {}, []
    for t in range(4):
        total = 6 + 3 * t
        b0, b1 = rng.integers(0, 2, total).astype(np.uint8), rng.integers(0, 2, total).astype(np.uint8)
        jobs.append((t, sk.encrypt_bits(b0, 20 + t, 0), sk.encrypt_bits(b1, 30 + t, 0)))

    def work(t, c0, c1):
        try:
            outs = []
            for op in ("NAND", "XOR", "OR"):
                outs.append(eoc.gate_batch(eoc.OPS[op], c0, c1))
            S = 3
            wires = np.zeros((n_wires, S, p.n + 1), np.int32)
            wires[aw[0]: aw[0] + 2] = c0[: 2 * S].reshape(2, S, -1)
            wires[bw[0]: bw[0] + 2] = c1[: 2 * S].reshape(2, S, -1)
            eoc.circuit_run(gates, wires, S)
            results[t] = (outs, wires[sw[0]: sw[0] + 3].copy())
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work, args=j) for j in jobs]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for t, c0, c1 in jobs:
        outs, sums = results[t]
        for got, op in zip(outs, ("NAND", "XOR", "OR")):
            assert np.array_equal(got, orc.gate_batch(ol.OPS[op], c0, c1))
        a = sk.decrypt_bits(c0[:6].reshape(2, 3, -1)[0]) + 2 * sk.decrypt_bits(c0[:6].reshape(2, 3, -1)[1])
        b = sk.decrypt_bits(c1[:6].reshape(2, 3, -1)[0]) + 2 * sk.decrypt_bits(c1[:6].reshape(2, 3, -1)[1])
        tot = sum(sk.decrypt_bits(sums[i]).astype(np.int64) << i for i in range(3))
        assert np.array_equal(tot, a.astype(np.int64) + b)


def test_chunked_host_pipeline(eoc, monkeypatch):
    """the chunked host pipeline (nchunks > 1: operand DMA on the H2D stream, all kernels on one stream,
    results leaving on the D2H stream, 2 x nchunks events) compared with the oracle row for row:
    5000 pinned rows on ONE engine (3 chunks by size: a chunk is the engine's resident set, 2048 rows where the
    one-wave-per-ciphertext kernel applies), then EOC_TFHE_HOST_CHUNKS=3 forced on pageable operands and on
    a mixed batch in arbitrary opcode order (MUX included, so the third operand travels too)."""
    p = eoc.default_params(0)
    p.n = 40
    sk = eoc.SecretKey(p, 9)
    orc = ol.Oracle(0, 9, n_override=40)
    eoc.gpu_shutdown()
    eoc.gpu_init(p, devices=[0])
    eoc.upload_cloud_key(sk)
    try:
        total = 5000
        assert eoc.lib().eoc_engine_resident_jobs(eoc.lib().eoc_global_engine_at(0)) in (1024, 2048)
        rng = np.random.default_rng(15)
        bits = [rng.integers(0, 2, total).astype(np.uint8) for _ in range(3)]
        cts = [sk.encrypt_bits(bits[k], 40 + k, 0) for k in range(3)]
        pin = [eoc.PinnedArray((total, p.n + 1)) for _ in range(4)]
        for k in range(3):
            pin[k].array[:] = cts[k]
        eoc.gate_batch(eoc.OPS["NAND"], pin[0].array, pin[1].array, out=pin[3].array)
        want = orc.gate_batch(ol.OPS["NAND"], cts[0], cts[1])
        assert np.array_equal(pin[3].array, want)
        assert np.array_equal(sk.decrypt_bits(pin[3].array), 1 - (bits[0] & bits[1]))
        # forced chunking: pageable operands (staged copies) and a mixed batch, ragged chunk sizes (700 = 234 + 233 + 233)
        monkeypatch.setenv("EOC_TFHE_HOST_CHUNKS", "3")
        m = 700
        out = eoc.gate_batch(eoc.OPS["XOR"], cts[0][:m].copy(), cts[1][:m].copy())
        assert np.array_equal(out, orc.gate_batch(ol.OPS["XOR"], cts[0][:m], cts[1][:m]))
        ops = rng.choice(np.array([0, 4, 10, 11, 2, 13, 12], np.uint8), m)
        out = eoc.gate_batch(0, cts[0][:m].copy(), cts[1][:m].copy(), cts[2][:m].copy(), ops=ops)
        assert np.array_equal(out, orc.gate_batch(0, cts[0][:m], cts[1][:m], cts[2][:m], ops=ops))
        pin[3].array[:] = 0
        eoc.gate_batch(0, pin[0].array[:m], pin[1].array[:m], pin[2].array[:m], ops=ops, out=pin[3].array[:m])
        assert np.array_equal(pin[3].array[:m], out)
        # an error inside the chunk loop (bad opcode in the LAST chunk) returns with nothing in flight:
        # the next call on the same buffers is correct
        bad = ops.copy()
        bad[-1] = 99
        with pytest.raises(eoc.EocError):
            eoc.gate_batch(0, pin[0].array[:m], pin[1].array[:m], pin[2].array[:m], ops=bad, out=pin[3].array[:m])
        eoc.gate_batch(0, pin[0].array[:m], pin[1].array[:m], pin[2].array[:m], ops=ops, out=pin[3].array[:m])
        assert np.array_equal(pin[3].array[:m], out)
        for a in pin:
            a.free()
    finally:
        eoc.gpu_shutdown()


def test_one_gate_calls_wake_no_worker(eoc):
    """engines 1..n-1 have persistent host threads; a call whose other blocks are empty (every
    one-gate call of the string API) wakes nobody and runs on the calling thread -- it costs what it costs on a
    single-engine context."""
    import time
    p = eoc.default_params(0)
    p.n = 40
    sk = eoc.SecretKey(p, 9)
    c0 = sk.encrypt_bits(np.array([1], np.uint8), 2, 0)
    c1 = sk.encrypt_bits(np.array([0], np.uint8), 3, 0)

    def median_call_us(devices):
        eoc.gpu_shutdown()
        eoc.gpu_init(p, devices=devices)
        eoc.upload_cloud_key(sk)
        for _ in range(20):
            eoc.gate_batch(eoc.OPS["NAND"], c0, c1)
        ts = []
        for _ in range(300):
            t0 = time.perf_counter()
            eoc.gate_batch(eoc.OPS["NAND"], c0, c1)
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts)) * 1e6

    try:
        t1 = median_call_us([0])
        t3 = median_call_us([0, 0, 0])
        st = eoc.stats_multi()
        assert st["worker_wakeups"] == [0, 0, 0], st
        # the deterministic statement is the wake-up counter above; the latency check is deliberately loose (medians of
        # 300 calls of ~1.9 ms; measured 1.00x) so that a noisy box cannot fail the suite
        assert t3 <= 1.25 * t1 + 20.0, (t1, t3)
        # a call with three non-empty blocks wakes engines 1 and 2 exactly once each
        c = sk.encrypt_bits(np.ones(11, np.uint8), 4, 0)
        eoc.gate_batch(eoc.OPS["AND"], c, c)
        assert eoc.stats_multi()["worker_wakeups"] == [0, 1, 1]
        # two gates: blocks 1 + 1 + 0 -- engine 2 stays asleep
        eoc.gate_batch(eoc.OPS["AND"], c[:2], c[:2])
        assert eoc.stats_multi()["worker_wakeups"] == [0, 2, 1]
    finally:
        eoc.gpu_shutdown()


def test_rccl_key_broadcast_two_devices(eoc):
    """the in-library RCCL broadcast (ncclCommInitAll + grouped ncclBroadcast through the dlopen'ed
    table) on DISTINCT devices.  Needs two visible GPUs -- skipped on the one-GPU boxes every builder round has had, so it
    has NEVER EXECUTED (README "never executed on hardware"); wherever two GPUs are visible it runs and FAILS LOUDLY if
    the broadcast, the method, or a single ciphertext is wrong (round 3 masked it with a non-strict xfail: removed).
    A process that already maps an RCCL (this harness: torch's) must re-use that copy."""
    if eoc.lib().eoc_device_count() < 2:
        pytest.skip("needs >= 2 visible GPUs (the peer-copy branch is what a one-GPU box can rehearse)")
    from eoc_tfhe_amd import circuits
    p = eoc.default_params(0)
    p.n = 40
    sk = eoc.SecretKey(p, 9)
    orc = ol.Oracle(0, 9, n_override=40)
    eoc.gpu_shutdown()
    eoc.gpu_init(p, devices=[0, 1])
    try:
        eoc.upload_cloud_key(sk)
        st = eoc.stats_multi()
        assert st["key_broadcast_method"] == "rccl", st
        assert st["rccl_origin"] in ("already mapped", "process symbols"), st   # torch is imported: no second RCCL
        total = 37
        rng = np.random.default_rng(2)
        b = [rng.integers(0, 2, total).astype(np.uint8) for _ in range(3)]
        c = [sk.encrypt_bits(b[k], 60 + k, 0) for k in range(3)]
        ops = rng.choice(np.array([0, 4, 10], np.uint8), total)
        out = eoc.gate_batch(0, c[0], c[1], c[2], ops=ops)
        assert np.array_equal(out, orc.gate_batch(0, c[0], c[1], c[2], ops=ops))
        per = [e["bootstraps"] for e in eoc.stats_multi()["engines"]]
        assert all(x > 0 for x in per), per                      # both devices evaluated their block with THEIR replica
        gates, n_wires, aw, bw, sw = circuits.ripple_carry_adder(4)
        S = 9
        A, B = rng.integers(0, 16, S), rng.integers(0, 16, S)
        wires = np.zeros((n_wires, S, p.n + 1), np.int32)
        for i in range(4):
            wires[aw[0] + i] = sk.encrypt_bits(((A >> i) & 1).astype(np.uint8), 100 + i, 0)
            wires[bw[0] + i] = sk.encrypt_bits(((B >> i) & 1).astype(np.uint8), 200 + i, 0)
        ref = wires.copy()
        eoc.circuit_run(gates, wires, S)
        for g in gates:
            ref[g.out] = orc.gate_batch(g.op, ref[g.in0], None if g.in1 < 0 else ref[g.in1], None if g.in2 < 0 else ref[g.in2])
        assert np.array_equal(wires[sw[0]: sw[0] + 5], ref[sw[0]: sw[0] + 5])
    finally:
        eoc.gpu_shutdown()


def test_process_exit_without_shutdown_is_clean(eoc):
    """the persistent worker threads are owned by the library's static context; a process that exits
    without eoc_gpu_shutdown / resetGateKey used to reach std::terminate (joinable std::thread destroyed at exit, SIGABRT,
    rc 134).  A child brings up two engines on device 0, runs one batch that wakes the worker, and simply returns."""
    import subprocess
    import sys
    import textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import sys
        sys.path.insert(0, %r)
        import numpy as np
        import eoc_tfhe_amd as eoc
        p = eoc.default_params(0); p.n = 12
        sk = eoc.SecretKey(p, 3)
        eoc.gpu_init(p, devices=[0, 0])
        eoc.upload_cloud_key(sk)
        c = sk.encrypt_bits(np.array([0, 1, 1, 0, 1], np.uint8), 7, 0)
        out = eoc.gate_batch(eoc.OPS["NAND"], c, c)
        assert sk.decrypt_bits(out).tolist() == [1, 0, 0, 1, 0]
        assert eoc.stats_multi()["worker_wakeups"] == [0, 1]
        print("child done, exiting WITHOUT eoc_gpu_shutdown", flush=True)
        %s
    """)
    for how in ("", "sys.exit(0)", "import os; os._exit(0)"):
        r = subprocess.run([sys.executable, "-c", code % (root, how)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (how, r.returncode, (r.stdout + r.stderr)[-2000:])
        assert "child done" in r.stdout


def test_rccl_call_path_on_one_gpu(eoc):
    """what a one-GPU box CAN execute of the RCCL branch: the library is found (the copy torch already mapped, not a
    second one), the dlopen'ed table works, a one-rank communicator broadcasts 8 MiB out of place and the bytes arrive"""
    L = eoc.lib()
    rc = L.eoc_rccl_selftest(0, 8 << 20)
    assert rc == 0, eoc.lib().eoc_last_error()
    assert L.eoc_rccl_origin().decode() in ("already mapped", "process symbols")


@pytest.mark.parametrize("devices", [[0], [0, 0, 0]])
def test_async_submit_wait_pipeline(eoc, devices):
    """eoc_gate_batch_submit / _wait: seven batches of different widths and opcodes kept two deep in flight (operands by
    DMA while the previous batch computes, results leaving under the next batch's kernels) equal the oracle row for
    row; a synchronous call in between drains the pipeline; pageable buffers and unknown tickets are refused."""
    p = eoc.default_params(0)
    p.n = 40
    sk = eoc.SecretKey(p, 9)
    orc = ol.Oracle(0, 9, n_override=40)
    eoc.gpu_shutdown()
    eoc.gpu_init(p, devices=devices)
    eoc.upload_cloud_key(sk)
    try:
        rng = np.random.default_rng(33)
        widths = [700, 64, 1030, 5, 333, 1, 512]
        jobs = []
        for k, w in enumerate(widths):
            pin = [eoc.PinnedArray((w, p.n + 1)) for _ in range(4)]
            for j in range(3):
                pin[j].array[:] = sk.encrypt_bits(rng.integers(0, 2, w).astype(np.uint8), 100 + 10 * k + j, 0)
            ops = rng.choice(np.array([0, 4, 10, 11, 2, 13], np.uint8), w) if k % 2 else None
            op = int(rng.choice([0, 1, 4, 10])) if ops is None else 0
            jobs.append((pin, op, ops))
        tickets = []
        for k, (pin, op, ops) in enumerate(jobs):
            if k >= 2:
                eoc.gate_batch_wait(tickets[k - 2])
                pk, opk, opsk = jobs[k - 2]
                want = orc.gate_batch(opk, pk[0].array, pk[1].array, pk[2].array, ops=opsk)
                assert np.array_equal(pk[3].array, want), f"batch {k - 2}"
            tickets.append(eoc.gate_batch_submit(op, pin[0].array, pin[1].array, pin[2].array, ops=ops, out=pin[3].array))
        assert tickets == sorted(tickets) and len(set(tickets)) == len(tickets)
        # a synchronous call drains what is still in flight (the last two submissions)
        c = sk.encrypt_bits(np.ones(9, np.uint8), 5, 0)
        assert np.array_equal(eoc.gate_batch(eoc.OPS["AND"], c, c), orc.gate_batch(ol.OPS["AND"], c, c))
        for k in (len(jobs) - 2, len(jobs) - 1):
            pk, opk, opsk = jobs[k]
            assert np.array_equal(pk[3].array, orc.gate_batch(opk, pk[0].array, pk[1].array, pk[2].array, ops=opsk)), f"batch {k}"
        for t in tickets:                      # waiting again (or late) is harmless
            eoc.gate_batch_wait(t)
        with pytest.raises(eoc.EocError):
            eoc.gate_batch_wait(tickets[-1] + 1)
        with pytest.raises(eoc.EocError, match="eoc_host_alloc"):
            eoc.gate_batch_submit(0, c, c, out=np.empty_like(c))
        for pin, _, _ in jobs:
            for a in pin:
                a.free()
    finally:
        eoc.gpu_shutdown()


def test_eight_engines_on_one_device(eoc):
    """VERDICT r5 task 3b: the in-library path at its REAL world size -- eight engines, seven persistent workers, peer-copy
    key replication to seven replicas, 8-way gather_blocks -- rehearsed on one GPU at small n: ragged blocks (8-way split of
    61 rows: 8 8 8 8 8 7 7 7), fewer gates than engines (5 rows: three engines idle and NOT woken), a mixed batch (every
    engine pools its block's groups), a circuit whose instances stay on their engine, and the asynchronous two-deep
    pipeline across all eight -- every result against the oracle, bit for bit"""
    from eoc_tfhe_amd import circuits
    p = eoc.default_params(0)
    p.n = 40
    sk = eoc.SecretKey(p, 9)
    orc = ol.Oracle(0, 9, n_override=40)
    eoc.gpu_shutdown()
    eoc.gpu_init(p, devices=[0] * 8)
    eoc.upload_cloud_key(sk)
    try:
        assert eoc.gpu_engine_count() == 8
        st = eoc.stats_multi()
        assert st["key_broadcast_method"] == "peer-copy" and len(st["engines"]) == 8
        rng = np.random.default_rng(88)
        total = 61
        b = [rng.integers(0, 2, total).astype(np.uint8) for _ in range(3)]
        c = [sk.encrypt_bits(b[k], 2 + k, 0) for k in range(3)]
        before = [e["bootstraps"] for e in eoc.stats_multi()["engines"]]
        out = eoc.gate_batch(eoc.OPS["NAND"], c[0], c[1])
        after = [e["bootstraps"] for e in eoc.stats_multi()["engines"]]
        assert [x - y for x, y in zip(after, before)] == [8, 8, 8, 8, 8, 7, 7, 7]
        assert np.array_equal(out, orc.gate_batch(ol.OPS["NAND"], c[0], c[1]))
        ops = rng.choice(np.array([0, 4, 10, 11, 2, 13, 10], np.uint8), total)
        assert np.array_equal(eoc.gate_batch(0, c[0], c[1], c[2], ops=ops), orc.gate_batch(0, c[0], c[1], c[2], ops=ops))
        assert np.array_equal(eoc.gate_batch(eoc.OPS["MUX"], c[0], c[1], c[2]), orc.gate_batch(ol.OPS["MUX"], c[0], c[1], c[2]))
        # fewer rows than engines: blocks 1 1 1 1 1 0 0 0 -- the idle engines' workers stay asleep
        wake0 = eoc.stats_multi()["worker_wakeups"]
        out = eoc.gate_batch(eoc.OPS["XOR"], c[0][:5], c[1][:5])
        wake1 = eoc.stats_multi()["worker_wakeups"]
        assert np.array_equal(out, orc.gate_batch(ol.OPS["XOR"], c[0][:5], c[1][:5]))
        assert [x - y for x, y in zip(wake1, wake0)] == [0, 1, 1, 1, 1, 0, 0, 0], (wake0, wake1)
        # a circuit over 13 instances (blocks 2 2 2 2 2 1 1 1): the picked adder form, every sum decrypts, bytes vs oracle
        gates, n_wires, aw, bw, sw = circuits.adder(4, 13)
        S = 13
        A, B = rng.integers(0, 16, S), rng.integers(0, 16, S)
        wires = np.zeros((n_wires, S, p.n + 1), np.int32)
        for i in range(4):
            wires[aw[0] + i] = sk.encrypt_bits(((A >> i) & 1).astype(np.uint8), 100 + i, 0)
            wires[bw[0] + i] = sk.encrypt_bits(((B >> i) & 1).astype(np.uint8), 200 + i, 0)
        ref = wires.copy()
        eoc.circuit_run(gates, wires, S)
        assert np.array_equal(sum(sk.decrypt_bits(wires[w]).astype(np.int64) << i for i, w in enumerate(sw)), A + B)
        for g in gates:
            ref[g.out] = orc.gate_batch(g.op, ref[g.in0] if g.in0 >= 0 else None, ref[g.in1] if g.in1 >= 0 else None,
                                        ref[g.in2] if g.in2 >= 0 else None)
        assert np.array_equal(wires[sw], ref[sw])
        # asynchronous, two deep, five batches of ragged widths over all eight engines
        widths = [100, 3, 257, 64, 9]
        jobs, tickets = [], []
        for k, w in enumerate(widths):
            pin = [eoc.PinnedArray((w, p.n + 1)) for _ in range(4)]
            for j in range(3):
                pin[j].array[:] = sk.encrypt_bits(rng.integers(0, 2, w).astype(np.uint8), 300 + 10 * k + j, 0)
            opsk = rng.choice(np.array([0, 4, 10, 12], np.uint8), w) if k % 2 else None
            jobs.append((pin, 10 if opsk is None else 0, opsk))
        for k, (pin, op, opsk) in enumerate(jobs):
            if k >= 2:
                eoc.gate_batch_wait(tickets[k - 2])
            tickets.append(eoc.gate_batch_submit(op, pin[0].array, pin[1].array, pin[2].array, ops=opsk, out=pin[3].array))
        for t in tickets:
            eoc.gate_batch_wait(t)
        for k, (pin, op, opsk) in enumerate(jobs):
            assert np.array_equal(pin[3].array, orc.gate_batch(op, pin[0].array, pin[1].array, pin[2].array, ops=opsk)), k
            for a in pin:
                a.free()
    finally:
        eoc.gpu_shutdown()


def test_freeing_a_pinned_buffer_drains_pending_submissions(eoc):
    """eoc_host_free (a Node Buffer finalizer, a PinnedArray going out of scope) must not release memory under
    a DMA in flight -- it first completes what is pending; a key re-upload drains too.  Two submissions in flight, the
    operand buffer of the second is freed at once; both results must still be the oracle's, and the engine stays usable"""
    p = eoc.default_params(0)
    p.n = 40
    sk = eoc.SecretKey(p, 9)
    orc = ol.Oracle(0, 9, n_override=40)
    eoc.gpu_shutdown()
    eoc.gpu_init(p, devices=[0, 0])
    eoc.upload_cloud_key(sk)
    try:
        rng = np.random.default_rng(6)
        w = 1500
        pins = [[eoc.PinnedArray((w, p.n + 1)) for _ in range(3)] for _ in range(2)]
        wants = []
        for b in range(2):
            for j in range(2):
                pins[b][j].array[:] = sk.encrypt_bits(rng.integers(0, 2, w).astype(np.uint8), 300 + 10 * b + j, 0)
            wants.append(orc.gate_batch(eoc.OPS["XOR"], pins[b][0].array, pins[b][1].array))
        t0 = eoc.gate_batch_submit(eoc.OPS["XOR"], pins[0][0].array, pins[0][1].array, out=pins[0][2].array)
        t1 = eoc.gate_batch_submit(eoc.OPS["XOR"], pins[1][0].array, pins[1][1].array, out=pins[1][2].array)
        pins[1][0].free()                                  # operand of the batch that was queued a moment ago
        assert np.array_equal(pins[0][2].array, wants[0]) and np.array_equal(pins[1][2].array, wants[1])
        eoc.gate_batch_wait(t0)
        eoc.gate_batch_wait(t1)
        t2 = eoc.gate_batch_submit(eoc.OPS["XOR"], pins[0][0].array, pins[0][1].array, out=pins[0][2].array)
        eoc.upload_cloud_key(sk)                           # re-upload under a pending submission: drains first
        assert np.array_equal(pins[0][2].array, wants[0])
        eoc.gate_batch_wait(t2)
        c = sk.encrypt_bits(np.ones(5, np.uint8), 5, 0)
        assert np.array_equal(eoc.gate_batch(eoc.OPS["NAND"], c, c), orc.gate_batch(ol.OPS["NAND"], c, c))
        for pb in pins:
            for a in pb:
                a.free()
    finally:
        eoc.gpu_shutdown()


def test_error_from_a_worker_block_reaches_the_caller(eoc, ctx3):
    """an error raised while engine 2's persistent thread evaluates its block comes back with ITS message
    (eoc_last_error is per thread: the worker hands the text over), and the context stays usable"""
    p, sk, orc = ctx3
    c = sk.encrypt_bits(np.ones(12, np.uint8), 1, 0)
    ops = np.zeros(12, np.uint8)
    ops[11] = 99                                   # blocks 4 + 4 + 4: the bad opcode is in engine 2's block
    with pytest.raises(eoc.EocError, match="bad opcode 99"):
        eoc.gate_batch(0, c, c, ops=ops)
    assert np.array_equal(eoc.gate_batch(eoc.OPS["OR"], c, c), orc.gate_batch(ol.OPS["OR"], c, c))


def test_key_of_another_shape_is_refused_by_the_engines(eoc):
    """the global engines size every copy and stride from THEIR parameters.  A key set or an imported cloud-key
    blob of another shape (Set A behind Set B engines and the reverse) must be refused by name -- before this check the
    first order read host memory out of bounds in hipMemcpy2D and the second evaluated garbage."""
    pa, pb = eoc.default_params(0), eoc.default_params(1)
    pa.n, pb.n = 24, 20
    ska, skb = eoc.SecretKey(pa, 3), eoc.SecretKey(pb, 3)
    for eng_p, other in ((pb, ska), (pa, skb)):
        eoc.gpu_shutdown()
        eoc.Tfhe.resetGateKey()
        eoc.gpu_init(eng_p, devices=[0])
        with pytest.raises(eoc.EocError) as ei:
            eoc.upload_cloud_key(other)
        assert "differs from the engines'" in str(ei.value)
        # the string / global surface: a cloud-key-only context of the other shape, engines already up
        eoc.global_import_cloud_key_blob(other.export_cloud_key())
        assert eoc.global_key_mode() == 2
        z = np.zeros((2, other.params.n + 1), np.int32)
        with pytest.raises(eoc.EocError) as ei:
            eoc.global_gate_batch(eoc.OPS["NAND"], z, z)
        assert "global engine was initialised for" in str(ei.value)
        eoc.Tfhe.resetGateKey()
    eoc.gpu_shutdown()
