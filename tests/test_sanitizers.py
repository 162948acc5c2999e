"""Host-side native code under sanitizers: the routing core's concurrency self-test is built with
AddressSanitizer+UBSan and with ThreadSanitizer and must finish clean.  (GPU code cannot run under
a sanitizer on the test machines; its checks are the numerics tests in test_kernels_gpu.py.)"""

import os
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
SRC = [str(ROOT / "csrc/routercore/selftest.cpp"), str(ROOT / "csrc/routercore/routercore.cpp")]


@pytest.mark.parametrize("name,flags", [
    ("asan-ubsan", ["-fsanitize=address,undefined", "-fno-omit-frame-pointer"]),
    ("tsan", ["-fsanitize=thread"]),
])
def test_routercore_selftest_under_sanitizer(tmp_path, name, flags):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    exe = tmp_path / f"selftest-{name}"
    build = subprocess.run([gxx, "-std=c++17", "-O1", "-g", *flags, "-I", str(ROOT / "csrc"), *SRC,
                            "-o", str(exe), "-lpthread"], capture_output=True, text=True)
    if build.returncode != 0 and "cannot find" in build.stderr:
        pytest.skip(f"sanitizer runtime not installed: {build.stderr.splitlines()[-1]}")
    assert build.returncode == 0, build.stderr
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", TSAN_OPTIONS="halt_on_error=1")
    env.pop("LD_PRELOAD", None)  # sanitizer runtimes insist on being the first loaded library
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300, env=env)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "selftest ok" in run.stdout
    assert "WARNING: ThreadSanitizer" not in run.stderr and "ERROR: AddressSanitizer" not in run.stderr
