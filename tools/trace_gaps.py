"""Device-timeline gaps by kernel pair from a rocprofv3 kernel-trace CSV: for consecutive kernels (by start time) the time
between the end of one and the start of the next, averaged per (previous kernel -> next kernel) pair, plus the timeline of
one denoiser call.  usage: trace_gaps.py <kernel_trace.csv>"""
import collections
import csv
import re
import sys

def short(n):
  m = re.search(r"(gc_\w+?_kernel)", n)
  return m.group(1)[3:-7] if m else re.sub(r"\(.*", "", n)[:24]

rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(open(sys.argv[1]))))
pair = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
  g = b[0] - a[1]
  if g < 50000:
    pair[(a[2], b[2])].append(g)
tot = sum(sum(v) for v in pair.values())
print(f"{'previous -> next':44s} {'count':>7s} {'avg gap us':>10s} {'sum ms':>8s}")
for k, v in sorted(pair.items(), key=lambda kv: -sum(kv[1]))[:24]:
  print(f"{(k[0] + ' -> ' + k[1])[:44]:44s} {len(v):7d} {sum(v) / len(v) / 1e3:10.2f} {sum(v) / 1e6:8.3f}")
span = rows[-1][1] - rows[0][0]
busy = sum(e - s for s, e, _ in rows)
print(f"\nlaunches {len(rows)}  span {span / 1e6:.3f} ms  kernel time {busy / 1e6:.3f} ms  gaps (< 50 us) {tot / 1e6:.3f} ms")
