"""CPU: sanitizer runs of the host-side code ("host ASan/UBSan build of the CPU restatement + shim
tests"; GPU sanitizers are not available on the pool).

* csrc/aeth_hostcore.h -- the pinned-range registry, the object pool of src/pool.rs:43-221 and the copy threads of the
  stream pipeline, exactly the code libaether_hip.so runs, with malloc standing in for hipHostMalloc through the pool's
  allocator hook -- driven from 8 threads by tests/cpp/hostcore_sanitize.cpp under -fsanitize=thread and
  -fsanitize=address,undefined;
* the oracle's C restatement rebuilt with -fsanitize=address,undefined (`make -C oracle asan`) and run through the
  reference's known-answer tests (tests/test_oracle_golden.py) in a child interpreter."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "hostcore_sanitize.cpp")
INC = os.path.join(ROOT, "aether_primitives_amd", "csrc")
OUT = os.path.join(ROOT, "tests", "cpp", "build")


@pytest.mark.parametrize("san", ["thread", "address,undefined"])
def test_hostcore_under_sanitizer(san):
    os.makedirs(OUT, exist_ok=True)
    exe = os.path.join(OUT, "hostcore_" + san.split(",")[0])
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", f"-fsanitize={san}", "-fno-sanitize-recover=all", "-I", INC,
                           SRC, "-o", exe, "-lpthread"])
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 second_deadlock_stack=1", ASAN_OPTIONS="detect_leaks=1",
               UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0 and "hostcore: ok" in p.stdout, p.stdout + p.stderr
    assert "WARNING: ThreadSanitizer" not in p.stderr and "ERROR: AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr


def test_oracle_known_answers_under_asan_ubsan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    so = os.path.join(ROOT, "oracle", "libaeth_oracle_asan.so")
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    env = dict(os.environ, AETH_ORACLE_SO=so, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle_golden.py"), "-x", "-q",
                        "-p", "no:cacheprovider"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "ERROR: AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr, p.stderr[-3000:]
