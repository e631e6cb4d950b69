"""bench.py's live PMC measurement (live_traffic), without a GPU: a stand-in `rocprofv3` on PATH writes the counter CSV a
real pass writes (rows per dispatch and counter, other kernels in between), fails or hangs on request -- the parser takes
the pair kernel's rows only, applies the microarch guide's correction ((2 x FETCH_SIZE + WRITE_SIZE) x 1024), derives the
co-bound figures, and turns every failure into an `error` / `pass_errors` entry instead of an exception."""
import importlib.util
import os
import stat
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FAKE = r'''#!/usr/bin/env python3
import os, sys, time
a = sys.argv[1:]
counters = a[a.index("--pmc") + 1: a.index("--output-format")]
out = a[a.index("-d") + 1]
mode = os.environ.get("FAKE_ROCPROF_MODE", "ok")
if mode == "fail" or (mode == "fail_sq" and "SQ_WAVE_CYCLES" in counters):
    sys.stderr.write("no such counter\n"); sys.exit(3)
if mode == "hang":
    time.sleep(60)
assert a[a.index("--") + 1].endswith("python3") or "python" in a[a.index("--") + 1]     # the program itself behind `--`
os.makedirs(os.path.join(out, "box", "1234"), exist_ok=True)
vals = {"FETCH_SIZE": 1000.0, "WRITE_SIZE": 50.0, "TCC_HIT_sum": 990.0, "TCC_MISS_sum": 10.0, "SQ_WAVE_CYCLES": 4.0e6,
        "SQ_WAIT_INST_LDS": 1.0e6, "SQ_INSTS_LDS": 2048 * 500 * 145.0, "SQ_INSTS_VALU": 2048 * 500 * 990.0,
        "SQ_LDS_BANK_CONFLICT": 0.0, "SQ_INSTS_VALU_FMA_F64": 2048 * 500 * 700.0, "SQ_INSTS_VALU_ADD_F64": 2048 * 500 * 88.0,
        "SQ_INSTS_VALU_MUL_F64": 2048 * 500 * 32.0}
with open(os.path.join(out, "box", "1234", "1234_counter_collection.csv"), "w") as f:
    f.write('"Correlation_Id","Dispatch_Id","Agent_Id","Kernel_Name","Counter_Name","Counter_Value"\n')
    for d in range(4):
        for c in counters:
            f.write(f'{d},{d},0,"void eoc::k_blind_rotate<2, 10, false>(eoc::BrArgs)","{c}",{vals[c] + (d - 1.5) * 2}\n')
            f.write(f'{d},{d},0,"void eoc::k_keyswitch_waves<8, 8, 32>(int)","{c}",7777777\n')
            f.write(f'{d},{d},0,"void eoc::k_blind_rotate_wide<10, false>(eoc::BrArgs)","{c}",5555555\n')
'''


@pytest.fixture
def bench(tmp_path, monkeypatch):
    exe = tmp_path / "rocprofv3"
    exe.write_text(FAKE)
    exe.chmod(exe.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv("PATH", str(tmp_path) + os.pathsep + os.environ["PATH"])
    for k in list(os.environ):
        if k.startswith(("ROCPROF", "ROCP_TOOL")):
            monkeypatch.delenv(k)
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_passes_are_parsed_and_corrected(bench, monkeypatch):
    monkeypatch.setenv("FAKE_ROCPROF_MODE", "ok")
    r = bench.live_traffic(1024, 500)
    assert "error" not in r and "pass_errors" not in r, r
    assert r["bytes_per_launch"] == (2 * 1000 + 50) * 1024 and r["dispatches_averaged"] == 4      # the pair kernel's rows only
    cb = r["cobounds"]
    assert cb["tcc_hit_rate"] == 0.99 and cb["tcc_hit_bytes_per_launch"] == 990 * 128 and cb["lds_wait_frac"] == 0.25
    assert cb["lds_insts_per_wave_step"] == 145.0 and cb["valu_insts_per_wave_step"] == 990.0
    assert cb["fp64_insts_per_wave_step"] == 700 + 88 + 32 + 16 and cb["lds_bank_conflict_cycles"] == 0.0
    # ... and the roofline block reports it as measured, with the stored figure beside it
    import eoc_tfhe_amd as eoc
    blk = bench.roofline_block(eoc.default_params(0), "A", 1024, 1024, 2.9, True, live=r)
    assert blk["traffic"] == r["bytes_per_launch"] and blk["traffic_source"].startswith("measured in this run")
    assert blk["traffic_stored"] is not None and blk["lds"]["source"].startswith("measured in this run")
    assert blk["hbm_measured"]["source"] == "this run's PMC passes"


def test_failures_become_entries_not_exceptions(bench, monkeypatch):
    monkeypatch.setenv("FAKE_ROCPROF_MODE", "fail")
    r = bench.live_traffic(1024, 500)
    assert "error" in r and "bytes_per_launch" not in r
    monkeypatch.setenv("FAKE_ROCPROF_MODE", "fail_sq")                                             # the byte counts survive
    r = bench.live_traffic(1024, 500)
    assert r["bytes_per_launch"] == 2050 * 1024 and "lds_wait_frac" not in r.get("cobounds", {}) and r["pass_errors"]
    monkeypatch.setenv("FAKE_ROCPROF_MODE", "hang")
    r = bench.live_traffic(1024, 500, timeout_s=1)
    assert "timed out" in r["error"]
    import eoc_tfhe_amd as eoc
    blk = bench.roofline_block(eoc.default_params(0), "A", 1024, 1024, 2.9, True, live=r)              # the labelled fallback
    assert "not measured in this run" in blk["traffic_source"] and blk["traffic"] == blk["traffic_stored"]
    monkeypatch.setenv("ROCPROFILER_FAKE", "1")                                                    # under a profiler: no nesting
    assert "being profiled" in bench.live_traffic(1024, 500)["error"]
