#!/bin/bash
# Round-2 trip 7: GPU suite (dtype aliases, tail split), drop-in rerun, bond-sweep with tail split,
# rocprofv3 kernel tables of the secondary workloads (SVD 4096^2, sliced D=12 network, MERA chi=32).
set -u
export TMPDIR=/tmp
R=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT
export TN_REFERENCE_DIR=$PWD/_reference_scratch
echo "== pytest gpu"
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 --deselect tests/test_gpu_reference_dropin.py > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_gpu.log
echo "== reference drop-in"
OUT=$OUT/refdropin PER_FILE_TIMEOUT=600 timeout 1500 bash tools/reference_dropin/run_reference_tests.sh 2>&1 | tail -14
echo "== bond sweep (tail split on / off)"
QUICK="--steps 3 --warmup 1 --svd-n 0 --rr-bond 0 --mera-chi 0 --no-extras --no-cpu-baseline --no-verify"
for arm in "tail_on:TNH_GEMM_TAIL_SPLIT=1" "tail_off:TNH_GEMM_TAIL_SPLIT=0"; do
  name=${arm%%:*}; envs=${arm#*:}
  env $envs timeout 300 python bench.py $QUICK > $OUT/ab_$name.json 2> $OUT/ab_$name.err; echo "$name rc=$?"
  python - "$OUT/ab_$name.json" "$name" <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "headline", round(r["value"]), r["roofline"]["kernel"])
for row in r.get("bond_sweep", []):
  print("   ", row["D"], row["layout"][:2], round(row["tflops"]), row["kernel"], row["permute_launches"])
PY
done
echo "== secondary kernel tables"
rm -rf $OUT/prof2_*
cd /tmp
run() { tag=$1; shift; timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof2_$tag -o p -- "$@" > $OUT/prof2_$tag.log 2>&1; echo "$tag rc=$?"; }
run svd4096 python $R/tools/svd_probe.py --check 0 --sizes 4096 --reps 1
run rr12 python $R/tools/rr64_probe.py --D 12 --min-slices 64 --max-slices 16
run mera32 python $R/tools/mera_probe.py --chi 32 --reps 1
cd $R
python - <<'PY'
import sqlite3, glob, os
out = []
for d in sorted(glob.glob('gpurun_out/prof2_*/')):
  dbs = glob.glob(d + '*.db')
  if not dbs: continue
  c = sqlite3.connect(dbs[0])
  out.append(f"# rocprofv3 --kernel-trace --stats   ({os.path.basename(d.rstrip('/'))})")
  out.append(f"{'calls':>7} {'total_ms':>10} {'avg_ms':>9} {'pct':>6}  kernel")
  for name, calls, total, avg, pct in list(c.execute("select * from top_kernels"))[:14]:
    out.append(f"{calls:7d} {total/1e3:10.2f} {avg/1e3:9.4f} {pct:6.2f}  {name[:120]}")
  out.append("")
open('gpurun_out/kernel_stats.txt', 'w').write("\n".join(out) + "\n")
print("\n".join(out))
PY
