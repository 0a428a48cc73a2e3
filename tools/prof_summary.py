"""Summarise rocprofv3 rocpd databases (gpurun_out/prof_*) into a text file for profiles/.

  python tools/prof_summary.py gpurun_out profiles/r01_bench_D256.txt
"""
import glob
import os
import sqlite3
import sys


def main(src, dst):
  lines = []
  for db in sorted(glob.glob(os.path.join(src, "prof_stats*", "*.db"))):
    c = sqlite3.connect(db)
    lines.append(f"# rocprofv3 --kernel-trace --stats   ({os.path.relpath(db, src)})")
    lines.append(f"{'calls':>6} {'total_ms':>12} {'avg_ms':>12} {'pct':>7}  kernel")
    for name, calls, total, avg, pct in c.execute("select * from top_kernels"):
      lines.append(f"{calls:6d} {total / 1e3:12.3f} {avg / 1e3:12.3f} {pct:7.2f}  {name}")
    lines.append("")
  for db in sorted(glob.glob(os.path.join(src, "prof_pmc_*", "*.db"))):
    c = sqlite3.connect(db)
    lines.append(f"# rocprofv3 --pmc   ({os.path.relpath(db, src)})  per-launch averages")
    q = ("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
         "group by kernel_name, counter_name order by kernel_name, counter_name")
    for name, ctr, n, avg in c.execute(q):
      lines.append(f"{ctr:32s} n={n:3d} avg={avg:20.1f}  {name[:90]}")
    lines.append("")
  with open(dst, "w") as f:
    f.write("\n".join(lines) + "\n")
  print("\n".join(lines))


if __name__ == "__main__":
  main(sys.argv[1], sys.argv[2])
