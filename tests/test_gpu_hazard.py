"""The MFMA -> read-of-the-result dependency, measured on the GPU box with that box's own compiler.
libxfeat_hip.so is built in the development container (ROCm 7.2 hipcc).  Rounds 2-5 carried a hand-counted pad between every K loop and its epilogue
(common.h: XFH_MFMA_SETTLE); round 6's sweep (tests/cpp/hazard_probe.hip, part 2: the last MFMA, N = 0..20 wait states, an optional taken branch and the first
VALU / global_store / ds_write read of the accumulators in one inline-asm statement, where no compiler pads anything) shows gfx950 interlocks the dependency: every
configuration is exact from N = 0 on, so the pad was removed.  The test prints the sweep, keeps the log honest about what it found (`red configurations`), fails
if the library-shaped kernels return a wrong value with the macro as shipped, and records the versions on both sides (xfh_version(): the clang / HIP the library
was built with and the runtime it met; the probe: the box's compiler)."""
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
    red = int(r.stdout.split("sweep red configurations:")[1].split()[0])
    if red:
        # the hardware did NOT interlock somewhere: the shipped macro has no wait states any more, so the library-shaped kernels above are the judge -- they
        # passed (returncode 0) -- but say it loudly: rebuild with -DXFH_SETTLE_NOPS before trusting this box
        print(f"  WARNING: {red} sweep configurations returned stale accumulators on this GPU: the interlock round 6 measured is not there; rebuild with -DXFH_SETTLE_NOPS")
