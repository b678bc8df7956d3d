"""xfh_bench_calib under a profiler: each mode moves exactly NBYTES per launch (1 GiB: far beyond the 256 MB Infinity Cache).
Run by tools/pmc_calib.sh under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE; tools/summarize_profiles.py derives the factors."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xfeatslam_amd import capi  # noqa: E402
from xfeatslam_amd.extractor import Context  # noqa: E402

NBYTES = 1 << 30
ctx = Context(nfeatures=64, max_height=32, max_width=32)
for mode in range(6):
    capi.check(capi.lib().xfh_bench_calib(ctx.h, mode, NBYTES, 3), ctx.h)
ctx.close()
print("calib done", NBYTES)
