"""Register / LDS / scratch usage of the compiled kernels (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kernel_resources.py [regex]      (runs here, no GPU needed)"""
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oareactdiff_amd", "csrc")
pat = re.compile(sys.argv[1] if len(sys.argv) > 1 else r"k_(gcl_edge|equi_edge|wgrad)")
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "-Wno-unused-result",
                    "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:] + ["oard_hip.hip", "-o", "/tmp/oard_res.o"], cwd=CSRC,
                   capture_output=True, text=True)
for b in r.stderr.split("Function Name: ")[1:]:
    name = b.split("\n")[0].split()[0].strip()
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if not pat.search(dem):
        continue
    g = lambda k: re.search(re.escape(k) + r": (\d+)", b).group(1)
    print("%-120s VGPR %3s AGPR %3s scratch %4s waves/SIMD %s LDS %s" % (
        dem[:120], g("VGPRs"), g("AGPRs"), g("ScratchSize [bytes/lane]"), g("Occupancy [waves/SIMD]"),
        g("LDS Size [bytes/block]")))
