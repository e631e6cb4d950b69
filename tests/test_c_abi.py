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


def test_host_code_under_asan_ubsan(tmp_path):
    """host.cpp + legacy.cpp + the engine's host side under ASan/UBSan (clang), driven by the plain-C host;
    leak check on, the OpenMP runtime's own start-up allocations suppressed"""
    import shutil
    if not os.path.exists("/opt/rocm/lib/llvm/bin/clang"):
        pytest.skip("no ROCm clang")
    out = subprocess.run(["bash", os.path.join(ROOT, "tools", "sanitize_host.sh"), str(tmp_path)],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    exe = out.stdout.strip().splitlines()[-1]
    supp = tmp_path / "lsan.supp"
    supp.write_text("leak:libomp.so\nleak:__kmp\n")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:protect_shadow_gap=0", LSAN_OPTIONS=f"suppressions={supp}",
               OMP_NUM_THREADS="2")
    r = subprocess.run([exe, "cpu"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "abi_smoke cpu OK" in r.stdout, (r.stdout + r.stderr)[-3000:]
    shutil.rmtree(tmp_path, ignore_errors=True)
