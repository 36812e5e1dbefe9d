#!/usr/bin/env python3
"""Per-kernel register / spill / LDS summary of rt_kernels.hip for gfx950 (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel.
usage: python tools/isa_stats.py [substring filter]      (also leaves the listing in /tmp/rt_kernels.s)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run(["make", "-C", os.path.join(ROOT, "raytracer-in-cpp_amd", "csrc"), "isa"], capture_output=True, text=True)
rows, cur = [], None
for ln in (out.stdout + out.stderr).splitlines():
    m = re.search(r"remark:\s+(.*?):\s+(.*?) \[-Rpass", ln)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
flt = sys.argv[1] if len(sys.argv) > 1 else ""
def short(n):
    d = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    d = d.replace("void rtamd::", "").replace("rtamd::", "")
    return d.split("(")[0]
print(f"{'kernel':48s} {'VGPR':>5s} {'vspill':>6s} {'sspill':>6s} {'scratch':>7s} {'occ':>3s} {'LDS':>6s}")
for r in rows:
    n = short(r["name"])
    if flt and flt not in n:
        continue
    print(f"{n:48s} {r.get('VGPRs','?'):>5s} {r.get('VGPRs Spill','?'):>6s} {r.get('SGPRs Spill','?'):>6s} {r.get('ScratchSize [bytes/lane]','?'):>7s} "
          f"{r.get('Occupancy [waves/SIMD]','?'):>3s} {r.get('LDS Size [bytes/block]','?'):>6s}")
