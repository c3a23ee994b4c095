"""Summarises a rocprofv3 kernel-trace CSV by (kernel, grid size): calls, avg us, total ms."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
  name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void gc::", "").replace("gc::", "")
  key = (name, r.get("Grid_Size", r.get("Grid_Size_X", "?")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?")))
  agg[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in agg.values())
print(f"{'kernel':46s} {'grid':>9s} {'wg':>5s} {'calls':>6s} {'avg us':>9s} {'total ms':>9s} {'%':>6s}")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
  print(f"{k[0][:46]:46s} {k[1]:>9s} {k[2]:>5s} {len(v):6d} {sum(v) / len(v) / 1e3:9.2f} {sum(v) / 1e6:9.3f} {100 * sum(v) / tot:6.2f}")
