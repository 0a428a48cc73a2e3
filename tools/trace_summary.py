"""Per-launch view of a rocprofv3 kernel trace (csv): durations and the gaps between consecutive kernels of ONE call of
the band SVD, sampled along the panel sequence.  usage: python tools/trace_summary.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict


def short(name):
  name = name.replace("tnh::svdb::", "").replace("void ", "")
  return name.split("(")[0][:28]


def main():
  rows = list(csv.DictReader(open(sys.argv[1])))
  rows.sort(key=lambda r: int(r["Start_Timestamp"]))
  # the last complete call: from the last gram_kernel<false> with the largest grid to the end
  names = [short(r["Kernel_Name"]) for r in rows]
  starts = [i for i, n in enumerate(names) if n.startswith("band_kernel")]
  # one call = ... stage 1 ... band_kernel ... ; take the segment that ends at the last band_kernel
  firsts = [i for i, n in enumerate(names) if n.startswith("gram_kernel<false>")]
  # split calls: a new call starts when a gram_kernel<false> follows something that is not part of stage 1
  calls, cur = [], None
  for i, n in enumerate(names):
    if n.startswith("gram_kernel<false>") and (i == 0 or not any(names[i - 1].startswith(x) for x in
                                                                 ("update_kernel", "rowupdate", "factor", "gram", "wpass", "wreduce", "ypass", "yreduce"))):
      cur = [i, i]
      calls.append(cur)
    if cur is not None:
      cur[1] = i
  if not calls:
    print("no call found")
    return
  a = calls[-1][0]
  b = len(rows) - 1
  seg = rows[a:b + 1]
  t0 = int(seg[0]["Start_Timestamp"])
  t1 = int(seg[-1]["End_Timestamp"])
  busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
  print(f"last call: {len(seg)} launches, span {(t1 - t0) / 1e6:.2f} ms, kernel time {busy / 1e6:.2f} ms, "
        f"idle {(t1 - t0 - busy) / 1e6:.2f} ms")
  gaps = defaultdict(list)
  for x, y in zip(seg[:-1], seg[1:]):
    gaps[short(y["Kernel_Name"])].append(int(y["Start_Timestamp"]) - int(x["End_Timestamp"]))
  print("mean gap in front of each kernel (us):")
  for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {k:30s} n={len(v):5d} mean={sum(v) / len(v) / 1e3:7.2f} total={sum(v) / 1e6:7.2f} ms")
  # per-panel samples: walk stage 1 by gram_kernel<false> occurrences
  idx = [i for i, r in enumerate(seg) if short(r["Kernel_Name"]).startswith("factor_kernel<false>")]
  for p in (0, 1, 16, 64, 128, 192, 240):
    if p >= len(idx):
      continue
    lo = idx[p]
    hi = idx[p + 1] if p + 1 < len(idx) else lo + 12
    parts = []
    for r in seg[lo:hi]:
      parts.append(f"{short(r['Kernel_Name'])[:14]}={(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:.1f}")
    span = (int(seg[hi - 1]["End_Timestamp"]) - int(seg[lo]["Start_Timestamp"])) / 1e3
    print(f"panel {p:3d}: span {span:7.1f} us | " + " ".join(parts))


if __name__ == "__main__":
  main()
