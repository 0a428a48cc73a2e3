"""Time-ordered list of the launches of a rocprofv3 kernel trace (csv) that last at least MIN_US, newest last.
usage: python tools/kernel_list.py <kernel_trace.csv> [min_us=200] [last=120]"""
import csv
import sys


def main():
  rows = list(csv.DictReader(open(sys.argv[1])))
  min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
  last = int(sys.argv[3]) if len(sys.argv) > 3 else 120
  rows.sort(key=lambda r: int(r["Start_Timestamp"]))
  t0 = int(rows[0]["Start_Timestamp"])
  out = []
  for r in rows:
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if dur < min_us:
      continue
    name = r["Kernel_Name"].replace("void ", "").replace("tnh::", "")
    grid = r.get("Grid_Size_X") or r.get("Grid_Size") or ""
    wg = r.get("Workgroup_Size_X") or r.get("Workgroup_Size") or ""
    out.append(f"t={(int(r['Start_Timestamp']) - t0) / 1e6:10.2f} ms  {dur / 1e3:9.3f} ms  grid={grid:>10s} wg={wg:>4s}  {name[:110]}")
  for line in out[-last:]:
    print(line)


if __name__ == "__main__":
  main()
