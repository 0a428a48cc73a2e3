"""HBM traffic per launch of the headline GEMM from the rocprofv3 PMC passes of bench.py
(tools/profile.sh) -> profiles/<tag>_traffic.json, which bench.py reports as roofline.traffic.

  python tools/traffic_json.py gpurun_out gpurun_out/bench.json profiles/r01_traffic.json

Corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE reports half of the bytes of wide (16 B/lane) coalesced reads -- this kernel's loads
are all `global_load_lds_dwordx4` -- so it is doubled; WRITE_SIZE is taken as is (uncalibrated)."""
import glob, json, os, sqlite3, sys


def avg_counter(src, counter, kernel_substr):
  for db in glob.glob(os.path.join(src, "prof_pmc_*", "*.db")):
    c = sqlite3.connect(db)
    try:
      rows = list(c.execute("select avg(value), count(*) from counters_collection where counter_name=? "
                            "and kernel_name like ?", (counter, f"%{kernel_substr}%")))
    except sqlite3.Error:
      continue
    if rows and rows[0][0] is not None:
      return float(rows[0][0]), int(rows[0][1])
  return None, 0


def main(src, bench_json, dst, kernel_substr="gemm_nt"):
  with open(bench_json) as f:
    bench = json.loads(f.read().strip().splitlines()[-1])
  fetch_kib, n1 = avg_counter(src, "FETCH_SIZE", kernel_substr)
  write_kib, n2 = avg_counter(src, "WRITE_SIZE", kernel_substr)
  if fetch_kib is None or write_kib is None:
    raise SystemExit("PMC databases with FETCH_SIZE / WRITE_SIZE not found under " + src)
  wl = bench["config"]["workload"]
  dims = wl[wl.index("GEMM ") + 5:].split(",")[0].split("x")
  rec = {
      "kernel": bench["roofline"]["kernel"], "shape": [int(d) for d in dims],
      "fetch_kib_raw": fetch_kib, "write_kib_raw": write_kib, "launches": [n1, n2],
      "fetch_bytes": 2.0 * fetch_kib * 1024.0, "write_bytes": write_kib * 1024.0,
      "hbm_bytes": 2.0 * fetch_kib * 1024.0 + write_kib * 1024.0,
      "correction": "FETCH_SIZE x2 (gfx950 wide-load half count), KiB -> bytes; WRITE_SIZE KiB -> bytes",
      "command": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 3 "
                 "--warmup 1 --no-cpu-baseline --svd-n 0 --rr-bond 0",
  }
  with open(dst, "w") as f:
    json.dump(rec, f, indent=1)
  print(json.dumps(rec))


if __name__ == "__main__":
  main(*sys.argv[1:])
