"""Summarises a rocprofv3 kernel-trace CSV by (kernel, grid size): calls, avg us, total ms."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
  name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void gc::", "").replace("gc::", "").replace("void gc_a16::", "a16 ").replace("gc_a16::", "a16 ")
  _a16 = name.startswith("_ZN6gc_a16")
  name = name.replace("_ZN6gc_a16", "_ZN2gc", 1)
  _m = re.search(r"(gc_\w+?_kernel)(IL\w*?E)?E", name) if name.startswith("_ZN2gc") else None
  name = (("a16 " if _a16 else "") + _m.group(1) + (("<" + _m.group(2) + ">") if _m.group(2) else "")) if _m else name
  key = (name, r.get("Grid_Size", r.get("Grid_Size_X", "?")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?")))
  agg[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in agg.values())
print(f"{'kernel':46s} {'grid':>9s} {'wg':>5s} {'calls':>6s} {'avg us':>9s} {'total ms':>9s} {'%':>6s}")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
  print(f"{k[0][:46]:46s} {k[1]:>9s} {k[2]:>5s} {len(v):6d} {sum(v) / len(v) / 1e3:9.2f} {sum(v) / 1e6:9.3f} {100 * sum(v) / tot:6.2f}")

# inter-kernel gaps on the device timeline (launch boundaries): consecutive kernels by start time
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
gaps = [ev[i][0] - ev[i - 1][1] for i in range(1, len(ev))]
short = [g for g in gaps if 0 <= g < 20000]          # boundaries inside a sample (longer ones are host phases)
over = sum(1 for g in gaps if g < 0)
print(f"\nlaunches {len(ev)}  kernel time {tot / 1e6:.3f} ms  inter-kernel gaps < 20 us: {len(short)} "
      f"sum {sum(short) / 1e6:.3f} ms avg {sum(short) / max(len(short), 1) / 1e3:.2f} us  overlapping {over}")
