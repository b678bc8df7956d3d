"""A hazard net that does not depend on the container's compiler.  libxfeat_hip.so is built in the development
container (ROCm 7.2 hipcc) and carries a hand-counted MFMA -> VALU-read pad (common.h: XFH_MFMA_SETTLE).  This test compiles
tests/cpp/hazard_probe.hip with the hipcc found ON THE GPU BOX, using the library's own macro, and checks MFMA -> XFH_MFMA_SETTLE -> taken
branch -> VALU read against a host fma chain for both MFMA forms the library issues; it also records the versions on both sides
(xfh_version(): the clang / HIP the library was built with and the runtime it met; the probe: the box's compiler)."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_library_reports_build_and_runtime_versions(gpu_lib):
    v = gpu_lib.xfh_version().decode()
    print("\n  " + v)
    assert "gfx950" in v and "built with clang" in v and "runtime HIP" in v
    rt = int(v.split("runtime HIP ")[1].split(",")[0])
    assert rt > 0                                   # the runtime answered


def test_mfma_settle_macro_with_the_boxs_own_compiler(gpu_lib, tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box: the padding cannot be re-checked against a local compiler (the oracle parity tests remain the net)")
    exe = str(tmp_path / "hazard_probe")
    ver = subprocess.run([hipcc, "--version"], capture_output=True, text=True).stdout.strip().splitlines()
    print("\n  box compiler: " + " | ".join(l.strip() for l in ver[:2]))
    # the library's own flags (csrc/Makefile): -O3 -ffp-contract=off
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-unused-value",
                           os.path.join(ROOT, "tests", "cpp", "hazard_probe.hip"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    print("  " + r.stdout.strip().replace("\n", "\n  "))
    assert r.returncode == 0 and "hazard probe ok" in r.stdout, (r.returncode, r.stdout, r.stderr)
