"""Per-kernel register / scratch use of one csrc translation unit (hipcc -Rpass-analysis=kernel-resource-usage), compactly.
usage: python tools/kernel_usage.py kernels_reproj.hip [name-filter]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
extra = ["-fno-slp-vectorize"] if src.startswith("kernels_") else []
extra += [a for a in sys.argv[3:] if a.startswith("-D")]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-mllvm", "-pragma-unroll-threshold=1000000", "-mllvm",
       "-unroll-threshold=1000000", *extra, "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
if src.endswith(".cpp"):
    cmd[1:1] = ["-x", "hip"]
out = subprocess.run(cmd, cwd=os.path.join(ROOT, "calibration_amd", "csrc"), capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark: .*?(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|VGPRs Spill|SGPRs Spill|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        continue
    if m.group(1) == "Function Name":
        cur = {"name": m.group(2)}
        rows.append(cur)
    elif cur is not None:
        cur[m.group(1).split(" ")[0] + ("Spill" if "Spill" in m.group(1) else "")] = m.group(2)
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n).replace("void cba::", "").replace("cba::", "")
    if flt in n:
        print(f"{n:90s} vgpr {r.get('VGPRs', '?'):>4s} agpr {r.get('AGPRs', '?'):>3s} scratch {r.get('ScratchSize', '?'):>5s} spill {r.get('VGPRsSpill', '?'):>4s} occ {r.get('Occupancy', '?')}")
