"""Kernel table of one rocprofv3 --kernel-trace --stats run (rocpd database under <dir>): top_kernels, microseconds.
  python tools/kernel_stats.py <dir> "<title>" > profiles/rNN_....txt"""
import glob
import os
import sqlite3
import sys

dbs = glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True)
if not dbs:
  sys.exit("no rocpd database under " + sys.argv[1])
rows = list(sqlite3.connect(dbs[0]).execute("select * from top_kernels"))
print("# " + (sys.argv[2] if len(sys.argv) > 2 else os.path.basename(dbs[0])))
print(f"{'calls':>7} {'total_ms':>10} {'avg_us':>9} {'pct':>6}  kernel")
for name, calls, total, avg, pct in rows:
  if pct >= 0.2:
    print(f"{calls:7d} {total / 1e3:10.3f} {avg:9.2f} {pct:6.2f}  {name[:120]}")
print(f"# all kernels: {sum(r[1] for r in rows)} launches, {sum(r[2] for r in rows) / 1e3:.1f} ms of kernel time")
