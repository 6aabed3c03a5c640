"""CPU: the AWGN generator's definition without a GPU.  csrc/aeth_rng.h compiled for the host (plain C, the same
-ffp-contract=off as the device build) must give the oracle's independently restated samples bit for bit, and its two
floating-point stages must hold the accuracy the header states.  (On the device the square root comes from v_rsq_f32
plus one correcting step instead of sqrtf; that the two agree for every reachable input is tools/rng_lab.hip's
exhaustive check and what test_awgn_*_bit_exact re-check on every sample they draw.)"""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_product_header_on_the_host_equals_the_oracle_restatement():
    out = os.path.join(ROOT, "tests", "cpp", "build")
    os.makedirs(out, exist_ok=True)
    exe = os.path.join(out, "rng_header_vs_oracle")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-std=gnu11",
                           "-I", os.path.join(ROOT, "aether_primitives_amd", "csrc"), "-I", os.path.join(ROOT, "oracle"),
                           os.path.join(ROOT, "tests", "cpp", "rng_header_vs_oracle.c"), os.path.join(ROOT, "oracle", "aeth_oracle.c"),
                           "-o", exe, "-lm", "-lpthread"])
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    m = re.match(r"differ (\d+) radius_rel_err (\S+) cossin_abs_err (\S+)", p.stdout)
    assert m, p.stdout
    assert int(m.group(1)) == 0, "product header and oracle restatement disagree on the generator's definition"
    assert float(m.group(2)) < 2.5e-7 and float(m.group(3)) < 6e-7, p.stdout        # header: 1.1e-7 / 2-3.5e-7
