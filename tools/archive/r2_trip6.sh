#!/bin/bash
# Round-2 trip 6: rocprofv3 evidence (headline stats + PMC, full-bench kernel table, helper / view PMC),
# and the N>1 code path of bench.py exercised with one rank (RCCL through K8, and the torch fallback).
set -u
export TMPDIR=/tmp
R=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT
rm -rf $OUT/prof_stats $OUT/prof_pmc_* $OUT/prof2_* $OUT/prof3_*
echo "== bench, 1 rank through the distributed path (RcclComm)"
TNH_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 600 python bench.py --steps 3 --warmup 1 --no-sweep --no-extras --no-cpu-baseline --svd-n 0 --mera-chi 0 > $OUT/bench_dist1.json 2> $OUT/bench_dist1.err; echo "rc=$?"; tail -3 $OUT/bench_dist1.err; python -c "
import json; r=json.loads(open('$OUT/bench_dist1.json').read().strip().splitlines()[-1]); print(r['value'], r['config'], json.dumps(r.get('sliced_network'))[:600])"
echo "== bench, torch fallback communicator"
TNH_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29534 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 600 python bench.py --comm torch --steps 3 --warmup 1 --no-sweep --no-extras --no-cpu-baseline --svd-n 0 --mera-chi 0 --no-verify > $OUT/bench_dist1_torch.json 2> $OUT/bench_dist1_torch.err; echo "rc=$?"; tail -3 $OUT/bench_dist1_torch.err; python -c "
import json; r=json.loads(open('$OUT/bench_dist1_torch.json').read().strip().splitlines()[-1]); print(r['value'], r['config']['comm'], json.dumps(r.get('sliced_network'))[:300])"
echo "== torchrun 1 proc (the driver's launcher)"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29535 bench.py --gpus 1 --steps 2 --warmup 1 --no-sweep --no-extras --no-cpu-baseline --svd-n 0 --mera-chi 0 --rr-bond 0 --no-verify > $OUT/bench_torchrun1.json 2> $OUT/bench_torchrun1.err; echo "rc=$?"; tail -2 $OUT/bench_torchrun1.err; tail -c 400 $OUT/bench_torchrun1.json
echo "== headline profile (stats + PMC)"
bash tools/profile.sh > $OUT/profile.log 2>&1; tail -8 $OUT/profile.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sweep --no-extras --no-verify --mera-chi 0 --svd-n 0 --rr-bond 0 > $OUT/bench_prof.json 2>/dev/null
python tools/prof_summary.py $OUT $OUT/prof_summary.txt > /dev/null; head -30 $OUT/prof_summary.txt
python tools/traffic_json.py $OUT $OUT/bench_prof.json $OUT/traffic.json gemm_nt_pp; cat $OUT/traffic.json
echo "== full bench kernel table"
cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof2_fullbench -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify > $OUT/prof2_fullbench.log 2>&1; echo "rc=$?"; cd $R
echo "== helper / view PMC"
cd /tmp
for ctr in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $ctr | tr ' ' '_' | cut -c1-30)
  timeout 300 rocprofv3 --pmc $ctr -d $OUT/prof3_$tag -o h -- python $R/tools/helper_one.py > $OUT/prof3_$tag.log 2>&1; echo "pmc $tag rc=$?"
done
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
  for name, calls, total, avg, pct in list(c.execute("select * from top_kernels"))[:40]:
    out.append(f"{calls:7d} {total/1e3:10.2f} {avg/1e3:9.4f} {pct:6.2f}  {name[:120]}")
  out.append("")
for db in sorted(glob.glob('gpurun_out/prof3_*/*.db')):
  c = sqlite3.connect(db)
  out.append(f"# rocprofv3 --pmc ({os.path.relpath(db, 'gpurun_out')})  per-launch averages, tools/helper_one.py")
  q = ("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
       "group by kernel_name, counter_name order by kernel_name, counter_name")
  for name, ctr, n, avg in c.execute(q):
    out.append(f"{ctr:28s} n={n:3d} avg={avg:18.1f}  {name[:100]}")
  out.append("")
open('gpurun_out/secondary_stats.txt', 'w').write("\n".join(out) + "\n")
print("\n".join(out)[:9000])
PY
