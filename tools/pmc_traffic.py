"""Turns rocprofv3 --pmc passes into per-kernel and per-class counter summaries.

usage: pmc_traffic.py <out_prefix> <counter_collection.csv> [<counter_collection.csv> ...]

Each CSV is one pass (`rocprofv3 --kernel-trace --pmc <COUNTERS> --output-format csv -- python3
tests/gpu_one_sample.py`).  Writes <out_prefix>_pmc_per_kernel.json (every counter averaged per
launch, per kernel symbol) and, when FETCH_SIZE and WRITE_SIZE are both present, traffic.json
next to it: HBM-side bytes per launch per kernel class = (2*FETCH_SIZE + WRITE_SIZE) * 1024
(gfx950 tallies a 128-byte read request as 64 bytes: MI355X_MICROARCH.md, HBM / rocprofv3 section).
"""
import collections
import csv
import json
import os
import re
import sys

CLASS_OF_GEMM = {5: "gc_gemm_qkv", 8: "gc_gemm_out", 9: "gc_gemm_ffw1", 10: "gc_gemm_ffw2", 11: "gc_gemm_node"}


def kernel_class(name):
  m = re.match(r"gc_gemm(\w*)_kernel<([^>]*)>", name)
  if m:
    parts = [p.strip() for p in m.group(2).split(",")]
    # template lists: gc_gemm_kernel<WM, WN, MT, NT, EPI, CLS, ..>, gc_gemm_dma_kernel<WM, WN, MT, NT, NS, EPI, CLS>,
    # gc_gemm_ws_kernel<MT, EPI, CLS, ..>, gc_gemm_rowop_kernel<NT, AMODE, CLS>
    idx = {"": 5, "_dma": 6, "_ws": 2, "_rowop": 2, "_lt": 1, "_lt2": 1, "_lt3": 1}.get(m.group(1))   # gc_gemm_lt*_kernel<EPI, CLS, ..>
    if idx is not None and idx < len(parts) and parts[idx].isdigit():
      return CLASS_OF_GEMM.get(int(parts[idx]))
    return None
  for prefix, cls in (("gc_ffw_fused", "gc_gemm_ffw1"), ("gc_attention", "gc_attention"), ("gc_attn_combine", "gc_attn_combine"),
                      ("gc_rowop", "gc_rowop"), ("gc_mlp", "gc_mlp"), ("gc_segsum", "gc_segsum"),
                      ("gc_cond", "gc_cond")):
    if name.startswith(prefix):
      return cls
  return "gc_pack"


def main():
  # optional leading "--mode NAME" (e.g. fp16_features): per-kernel JSON gets the suffix _NAME and the class table is
  # stored under traffic.json["NAME"] beside the default (float32-feature) table
  mode = None
  if sys.argv[1] == "--mode":
    mode = sys.argv[2]
    del sys.argv[1:3]
  prefix, paths = sys.argv[1], sys.argv[2:]
  per = collections.defaultdict(lambda: collections.defaultdict(list))
  for path in paths:
    for r in csv.DictReader(open(path)):
      name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void gc::", "").replace("gc::", "").replace("void gc_a16::", "").replace("gc_a16::", "").replace("void gc_lt::", "").replace("gc_lt::", "")
      if name.startswith("_ZN2gc") or name.startswith("_ZN6gc_a16"):   # rocprofv3 leaves some symbols mangled (e.g. _Float16 arguments)
        name = name.replace("_ZN6gc_a16", "_ZN2gc", 1)
        m = re.search(r"(gc_\w+?_kernel)(IL[\w]*?E)?E", name)
        name = m.group(1) + (("<" + m.group(2) + ">") if m and m.group(2) else "") if m else name
      if not name.startswith("gc_"):
        continue
      per[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
  out = {}
  for name, ctrs in sorted(per.items()):
    out[name] = {"launches": max(len(v) for v in ctrs.values())}
    for c, v in sorted(ctrs.items()):
      out[name][c + "_per_launch"] = sum(v) / len(v)
    if "FETCH_SIZE" in ctrs and "WRITE_SIZE" in ctrs:
      out[name]["hbm_mb_corrected"] = (2 * out[name]["FETCH_SIZE_per_launch"] +
                                       out[name]["WRITE_SIZE_per_launch"]) * 1024 / 1e6
    if "SQ_VALU_MFMA_BUSY_CYCLES" in ctrs and "GRBM_GUI_ACTIVE" in ctrs:
      # MFMA_BUSY sums over the chip's 1024 SIMDs; GUI_ACTIVE sums over the 8 XCDs
      out[name]["mfma_busy_frac"] = (out[name]["SQ_VALU_MFMA_BUSY_CYCLES_per_launch"] / 1024) / \
                                    (out[name]["GRBM_GUI_ACTIVE_per_launch"] / 8)
  json.dump(out, open(prefix + ("_" + mode if mode else "") + "_pmc_per_kernel.json", "w"), indent=1)
  classes = collections.defaultdict(lambda: [0.0, 0])
  for name, d in out.items():
    if "hbm_mb_corrected" in d:
      c = classes[kernel_class(name)]
      c[0] += d["hbm_mb_corrected"] * 1e6 * d["launches"]
      c[1] += d["launches"]
  if classes:
    t = {k: int(v[0] / v[1]) for k, v in classes.items()}
    t["_note"] = ("bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 from separate rocprofv3 --pmc passes (" +
                  os.path.basename(prefix) + "_pmc_per_kernel.json); FETCH_SIZE doubled per MI355X_MICROARCH.md "
                  "(gfx950 reports half of wide coalesced reads); counted at the L2's memory side, "
                  "Infinity-Cache hits included")
    tpath = os.path.join(os.path.dirname(prefix) or ".", "traffic.json")
    if mode:
      full = json.load(open(tpath)) if os.path.exists(tpath) else {}
      full[mode] = t
      t = full
    json.dump(t, open(tpath, "w"), indent=1)
  for name, d in out.items():
    print(name, {k: round(v, 3) for k, v in d.items()})


if __name__ == "__main__":
  main()
