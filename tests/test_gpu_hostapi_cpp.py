"""GPU: the C++ host wrapper (include/aether_hip.hpp) replays the reference's own unit
tests and doctests through the C ABI (tests/cpp/test_hostapi.cpp)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "test_hostapi")


@pytest.mark.gpu
def test_cpp_host_api(ctx):
    assert os.path.exists(BIN), "tests/cpp/test_hostapi not built (run __graft_entry__.build())"
    p = subprocess.run([BIN], capture_output=True, text=True, timeout=120)
    print(p.stdout)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "PASSED" in p.stdout and "FAIL " not in p.stdout


def test_cpp_host_api_builds():
    """CPU: the wrapper compiles against the header and links the library."""
    assert os.path.exists(BIN), "tests/cpp/test_hostapi not built (run __graft_entry__.build())"
    out = subprocess.check_output(["ldd", BIN], text=True)
    assert "libaether_hip.so" in out and "not found" not in out.split("libaether_hip.so")[1].split("\n")[0]
