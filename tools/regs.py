#!/usr/bin/env python3
"""usage: tools/regs.py mesh-vae_amd/csrc/foo.hip [filter-regex]
Prints kernel | VGPRs | spills | SGPRs | occupancy from hipcc's kernel-resource-usage remarks."""
import re
import subprocess
import sys

src = sys.argv[1]
filt = re.compile(sys.argv[2] if len(sys.argv) > 2 else ".")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-slp-vectorize",
       "-I/root/repo/include", "-c", src, "-o", "/tmp/regs_tmp.o", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
name, row = None, {}
keys = (("VGPRs Spill:", "sp"), ("    VGPRs:", "v"), ("    SGPRs:", "s"), ("Occupancy [waves/SIMD]:", "occ"),
        ("LDS Size [bytes/block]:", "lds"))
for line in out.splitlines():
    if " error" in line:
        print(line)
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name, row = m.group(1), {}
    for key, tag in keys:
        if key in line and name:
            row[tag] = line.split(key)[1].split()[0]
            if tag == "lds":
                short = re.sub(r"^_ZN3mvh\d+", "", name)
                short = re.sub(r"EEv.*$", "", short)
                if filt.search(short):
                    print("%-46s vgpr=%s spill=%s sgpr=%s occ=%s" % (short, row.get("v"), row.get("sp"), row.get("s"), row.get("occ")))
            break
