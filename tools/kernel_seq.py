"""The LAST `--slices`-th part of a rocprofv3 kernel trace (rocpd database under <dir>) as a sequence: kernels of at
least --min-us microseconds in launch order, with grid size and duration -- which permute, which product.
  python tools/kernel_seq.py <dir> --slices 8 --min-us 20"""
import argparse, glob, os, sqlite3, sys

ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--slices", type=int, default=8)
ap.add_argument("--min-us", type=float, default=20.0)
a = ap.parse_args()
dbs = glob.glob(os.path.join(a.dir, "**", "*.db"), recursive=True)
rows = list(sqlite3.connect(dbs[0]).execute("select name, start, duration, grid_x, workgroup_x from kernels order by start"))
n = len(rows) // a.slices
tail = rows[-n:]
t0 = tail[0][1]
total = sum(r[2] for r in tail) / 1e3
print(f"# last {n} of {len(rows)} dispatches: {total:.0f} us of kernel time, {(tail[-1][1] + tail[-1][2] - t0) / 1e3:.0f} us wall")
for name, start, dur, gx, wx in tail:
  if dur / 1e3 >= a.min_us:
    short = name.replace("void tnh::", "").split("(")[0][:70]
    print(f"{(start - t0) / 1e3:9.0f} us  {dur / 1e3:8.1f} us  grid {gx // max(wx, 1):6d}  {short}")
