"""Register / LDS / scratch usage of the compiled kernels (hipcc -Rpass-analysis=kernel-resource-usage), every translation unit of the library.
usage: python tools/kernel_resources.py [regex] [extra hipcc flags ...]      (runs here, no GPU needed)"""
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oareactdiff_amd.build import SOURCES  # noqa: E402

CSRC = os.path.join(ROOT, "oareactdiff_amd", "csrc")
pat = re.compile(sys.argv[1] if len(sys.argv) > 1 else r"k_(gcl_edge|equi_edge|wgrad)")


def unit(src):
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "--cuda-device-only", "-Wno-unused-result",
                        "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:] + [src, "-o", f"/tmp/oard_res_{os.path.splitext(src)[0]}.o"],
                       cwd=CSRC, capture_output=True, text=True)
    return r.stderr


with ThreadPoolExecutor(max_workers=len(SOURCES)) as pool:
    errs = list(pool.map(unit, SOURCES))
for err in errs:
    for b in err.split("Function Name: ")[1:]:
        name = b.split("\n")[0].split()[0].strip()
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        if not pat.search(dem):
            continue
        g = lambda k: re.search(re.escape(k) + r": (\d+)", b).group(1)          # noqa: E731
        print("%-120s VGPR %3s AGPR %3s scratch %4s waves/SIMD %s LDS %s" % (
            dem[:120], g("VGPRs"), g("AGPRs"), g("ScratchSize [bytes/lane]"), g("Occupancy [waves/SIMD]"),
            g("LDS Size [bytes/block]")))
