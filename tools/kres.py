#!/usr/bin/env python3
"""Register / scratch / occupancy of the kernels of one csrc file whose mangled name contains a pattern:
   python tools/kres.py walkq.hip itemgen [-DFOO ...]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, pat, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "--cuda-device-only",
                      "-Rpass-analysis=kernel-resource-usage", *extra, "-c", src, "-o", os.devnull],
                     cwd=os.path.join(ROOT, "dynesty_amd", "csrc"), capture_output=True, text=True)
if out.returncode:
    sys.exit(out.stderr[-4000:])
name, rows = None, {}
for line in out.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = m.group(1); rows[name] = {}; continue
    m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", line)
    if m and name:
        rows[name][m.group(1).strip()] = int(m.group(2))
for k, r in rows.items():
    if pat in k:
        print(k)
        print("   ", {a: r.get(a) for a in ("VGPRs", "AGPRs", "SGPRs", "VGPRs Spill", "SGPRs Spill", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]")})
