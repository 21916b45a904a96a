#!/usr/bin/env python3
"""Per-kernel register / spill / scratch report of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage), run in the
build container: tools/kernel_resources.py gnode_pers64.hip [extra flags].  Kernels that use scratch are marked: a kernel
with scratch cannot be captured into a HIP graph on first use (the runtime grows the scratch pool synchronously)."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gn-ode-sir_amd", "csrc")
src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
       "-I" + CSRC, *sys.argv[2:], "-c", os.path.join(CSRC, src), "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}
rows = []
for line in out.splitlines():
    m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|SGPRs Spill|VGPRs Spill|ScratchSize \[bytes/lane\]|TotalSGPRs|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        if "error" in line: print(line)
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": v}; rows.append(cur)
    else:
        cur[k] = v
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void ", "")
    flag = "  <-- SCRATCH" if r.get("ScratchSize [bytes/lane]", "0") != "0" else ""
    print(f"{name:48s} vgpr {r.get('VGPRs'):>4s} sgpr {r.get('TotalSGPRs'):>4s} vspill {r.get('VGPRs Spill'):>3s} sspill {r.get('SGPRs Spill'):>3s} scratch {r.get('ScratchSize [bytes/lane]'):>4s}{flag}")
