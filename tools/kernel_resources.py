#!/usr/bin/env python3
"""Register / scratch census of the gfx950 kernels: compiles csrc/gc_kernels.hip with
-Rpass-analysis=kernel-resource-usage and prints one line per kernel (VGPRs, AGPRs, spilled VGPRs, scratch bytes,
waves per SIMD).  usage: tools/kernel_resources.py [-DGC_TU_A16] [regex]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
flags = [a for a in sys.argv[1:] if a.startswith("-")]
pats = [a for a in sys.argv[1:] if not a.startswith("-")]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function", *flags,
       "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(ROOT, "gencast-flax-nnx_amd", "csrc", "gc_kernels.hip"),
       "-o", "/tmp/kernel_resources_probe.o"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
keys = (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("spill", r"VGPRs? Spill: (\d+)"),
        ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"))
for ln in out.splitlines():
  m = re.search(r"Function Name: (\S+)", ln)
  if m:
    cur = {"name": m.group(1)}
    rows.append(cur)
    continue
  for k, p in keys:
    m = re.search(p, ln)
    if m and cur is not None:
      cur[k] = int(m.group(1))
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows),
                       capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
  n = re.sub(r"^void ", "", n)
  n = re.sub(r"\(.*$", "", n)
  if pats and not any(re.search(p, n) for p in pats):
    continue
  print("%-96s v%4d a%4d spill%5d scratch%6d occ%2d" % (n[:96], r.get("vgpr", -1), r.get("agpr", -1), r.get("spill", -1),
                                                         r.get("scratch", -1), r.get("occ", -1)))
