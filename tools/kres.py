#!/usr/bin/env python3
"""Print per-kernel register/LDS/scratch usage of a .hip file (hipcc -Rpass-analysis)."""
import re, subprocess, sys
src = sys.argv[1]; extra = sys.argv[2:]
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", src, "-o", "/dev/null",
                      "-Rpass-analysis=kernel-resource-usage"] + extra, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
    if not m: continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip(); rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1); rows[cur][k.strip()] = v.strip()
for k, r in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(anonymous namespace\)::", "", name)[:70]
    print(f"{name:70s} vgpr={r.get('VGPRs')} agpr={r.get('AGPRs')} sgpr={r.get('SGPRs')} scratch={r.get('ScratchSize [bytes/lane]')} occ={r.get('Occupancy [waves/SIMD]')} lds={r.get('LDS Size [bytes/block]')}")
