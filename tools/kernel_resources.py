#!/usr/bin/env python3
"""Register / LDS / occupancy table of the kernels in one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
usage: tools/kernel_resources.py spike-petsc_amd/csrc/spike_kernels.hip [name-filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-c", src, "-o", "/dev/null",
                    "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
blocks = re.split(r"remark: Function Name: ", r.stderr)[1:]
dem = lambda n: subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
for b in blocks:
    name = b.split()[0]
    if flt and flt not in name:
        continue
    g = lambda k: re.search(re.escape(k) + r": (\d+)", b).group(1)
    print("%-70s VGPR %3s AGPR %3s SGPR %3s scratch %4s occ %s LDS %6s" % (dem(name)[:70], g("VGPRs"), g("AGPRs"), g("TotalSGPRs"),
          g("ScratchSize [bytes/lane]"), g("Occupancy [waves/SIMD]"), g("LDS Size [bytes/block]")))
