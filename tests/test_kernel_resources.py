"""Occupancy guard (CPU, no GPU needed): reads the gfx950 code objects out of the built library and checks the
register budget of the kernels whose speed depends on it.  A template flag that pushes the fused FIR kernel past 256
VGPRs silently halves its occupancy (one wave per SIMD): the decimating-store build did exactly that once (62 us per
launch instead of 50)."""
import os
import re
import shutil
import subprocess

import pytest

LLVM = "/opt/rocm/lib/llvm/bin"
LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "aether_primitives_amd", "lib", "libaether_hip.so")


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    objdump, readelf = os.path.join(LLVM, "llvm-objdump"), os.path.join(LLVM, "llvm-readelf")
    if not (os.path.exists(objdump) and os.path.exists(readelf)):
        pytest.skip("llvm-objdump / llvm-readelf of the ROCm toolchain not present")
    if not os.path.exists(LIB):
        pytest.skip("library not built")
    d = tmp_path_factory.mktemp("co")
    so = shutil.copy(LIB, d / "lib.so")
    subprocess.run([objdump, "--offloading", so], check=True, capture_output=True, cwd=d)
    out = {}
    for f in sorted(os.listdir(d)):
        if "gfx950" not in f:
            continue
        notes = subprocess.run([readelf, "--notes", str(d / f)], check=True, capture_output=True, text=True).stdout
        cur = {}
        for line in notes.splitlines():
            m = re.match(r"\s+\.(name|vgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size):\s+(\S+)", line)
            if not m:
                continue
            if m.group(1) == "name":
                cur = out.setdefault(m.group(2), {})
            else:
                cur[m.group(1)] = int(m.group(2))
    assert len(out) > 500, f"only {len(out)} kernels found in the library's code objects"
    return out


# kernels that were measured WITH their spills and still won their slot: the one-workgroup transforms of 7290 points and
# of 16000 ... 20480 points (the tuner timed every candidate as built), the 8192-point power-of-two kernel held to two
# waves per SIMD, and step A of the four-step for columns of 1024+ points (1024-lane workgroups, 128 VGPRs; the
# 512-lane build without spills was measured and is no faster: AETH_4S_MAXL)
SPILLS_MEASURED = (r"fft_ragged_kernel.*RCfgILi(7290|1[6-9]\d{3}|20\d{3})E", r"fft_pow2_stream_kernel.*CfgILi8192E",
                   r"fourstep_cols.*CfgILi(1024|2048|4096)E")


def test_no_other_kernel_spills_vector_registers_or_uses_scratch(kernels):
    bad = {k: v for k, v in kernels.items()
           if (v.get("vgpr_spill_count", 0) or v.get("private_segment_fixed_size", 0))
           and not any(re.search(p, k) for p in SPILLS_MEASURED)}
    assert not bad, f"{len(bad)} kernels spill or use scratch, e.g. {list(bad.items())[:3]}"


def test_fused_fir_kernels_keep_two_waves_per_simd(kernels):
    fmi = {k: v for k, v in kernels.items() if "fmi_kernel" in k}
    assert len(fmi) >= 40
    over = {k: v["vgpr_count"] for k, v in fmi.items() if v["vgpr_count"] > 256}
    assert not over, f"fused FIR builds above 256 VGPRs (one wave per SIMD): {over}"


def test_lean_mixed_radix_build_stays_lean(kernels):
    """plans without a factor 11..23 use the build without the big register butterflies (five waves per SIMD)"""
    lean = {k: v["vgpr_count"] for k, v in kernels.items() if "fft_mixed_kernel" in k and "Li0ELb0E" in k}
    assert lean and max(lean.values()) <= 100, lean
