"""The C ABI from plain C: tests/c/abi_smoke.c is compiled with gcc against include/eoc_tfhe_gpu.h and run."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = str(tmp_path / "abi_smoke")
    libdir = os.path.join(ROOT, "eoc_tfhe_amd")
    cmd = ["gcc", "-std=c11", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c", "abi_smoke.c"), "-o", exe, "-L" + libdir, "-leoc_tfhe_gpu",
           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return exe


def test_header_is_c_and_client_side_runs(tmp_path, built_lib):
    exe = _build(tmp_path)
    r = subprocess.run([exe, "cpu"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "abi_smoke cpu OK" in r.stdout


@pytest.mark.gpu
def test_plain_c_host_runs_gates(tmp_path, built_lib):
    exe = _build(tmp_path)
    r = subprocess.run([exe, "gpu"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "abi_smoke gpu OK" in r.stdout
