"""Register / scratch / LDS table of every kernel of one source file, from the compiler's own remarks (no GPU needed).
   python scripts/kernel_resources.py conv3x3_tile_bf3.hip [extra hipcc flags...]"""
import os
import re
import subprocess
import sys

src = sys.argv[1]
csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "footprints_amd", "csrc")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-fno-vectorize", "-I../../include", "-I.",
       "-Wno-unused-function", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + sys.argv[2:]
if src == "data_path.hip":
    cmd.insert(-4, "-ffp-contract=off")
out = subprocess.run(cmd, cwd=csrc, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark: [^:]+:\d+:\d+:\s+(.*?)\s*\[-Rpass-analysis", line) or re.search(r"remark:\s+(.*?)\s*\[-Rpass-analysis", line)
    if not m:
        m = re.search(r":\d+:\d+:\s+(.*?)\s*\[-Rpass-analysis", line)
    if not m:
        continue
    t = m.group(1)
    if t.startswith("Function Name:") or t.startswith("Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
dem = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
print("%-110s %5s %5s %7s %5s %4s %6s" % ("kernel", "VGPR", "AGPR", "scratch", "SGPR", "occ", "LDS"))
for r, d in zip(rows, dem):
    d = re.sub(r"\(anonymous namespace\)::", "", d)
    d = re.sub(r"\(.*\)$", "", d)
    print("%-110s %5s %5s %7s %5s %4s %6s" % (d[:110], r.get("VGPRs", "?"), r.get("AGPRs", "?"), r.get("ScratchSize [bytes/lane]", "?"),
                                            r.get("SGPRs", "?"), r.get("Occupancy [waves/SIMD]", "?"), r.get("LDS Size [bytes/block]", "?")))
