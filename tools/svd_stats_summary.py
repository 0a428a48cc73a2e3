"""Kernel tables of the band SVD from the rocprofv3 rocpd databases gpurun_out/prof_svd_{f32,f64}/*.db
(tools/r4_final.sh) -> <dst>/svd_band_{f32,f64}_kernel_stats.txt.  top_kernels: microseconds."""
import glob
import os
import sqlite3
import sys

src, dst = sys.argv[1], sys.argv[2]
for dt in ("f32", "f64"):
  dbs = glob.glob(os.path.join(src, f"prof_svd_{dt}", "*.db"))
  if not dbs:
    continue
  rows = list(sqlite3.connect(dbs[0]).execute("select * from top_kernels"))
  lines = [f"# rocprofv3 --kernel-trace --stats -- python tools/svd_stats_run.py {dt}   (3 calls of be.svd(4096 x 4096 {dt}, "
           "max_singular_values=256); MI355X, round 6, tools/r6_final.sh)",
           f"{'calls':>7} {'total_ms':>10} {'avg_us':>9} {'pct':>6}  kernel"]
  tot, ncalls = sum(r[2] for r in rows), sum(r[1] for r in rows)
  for name, calls, total, avg, pct in rows:
    if pct >= 0.3:
      lines.append(f"{calls:7d} {total / 1e3:10.3f} {avg:9.2f} {pct:6.2f}  {name[:110]}")
  lines.append(f"# all kernels: {ncalls} launches and {tot / 1e3:.1f} ms of kernel time over 3 calls = {ncalls // 3} launches and "
               f"{tot / 3e3:.1f} ms of kernel time per call (under the profiler)")
  with open(os.path.join(dst, f"svd_band_{dt}_kernel_stats.txt"), "w") as f:
    f.write("\n".join(lines) + "\n")
  print("\n".join(lines[:10] + lines[-1:]))
