#!/bin/bash
# K-loop gather kernel: parity, the bench leg with its K = 1728 row, the D = 12 network (timing, then the partials check).
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4t23; mkdir -p $O
timeout 100 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k gemm_gather --timeout 90 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 150 python tools/gather_probe.py > $O/gather_gemm_rows.jsonl 2> $O/gather_probe.err; echo "probe rc=$?"; python - <<'PY'
import json
for line in open("gpurun_out/r4t23/gather_gemm_rows.jsonl"):
  r = json.loads(line)
  if "case" in r:
    print(r["case"], r["box"], {o: [round(r[o]["classic_us"]), round(r[o]["gather_us"]), r[o].get("max_abs_difference"), r[o].get("rel_difference"), r[o]["gather_kernel"]] for o in ("small_first", "long_first")})
  else:
    print(r)
PY
tail -3 $O/gather_probe.err
for g in 0 1; do
  TNH_GATHER_GEMM=$g timeout 100 python tools/rr64_probe.py --D 12 --max-slices 8 > $O/rr64_gather$g.json 2> $O/rr64_gather$g.err; echo "rr64 gather=$g rc=$?"; python -c "import json;r=json.load(open('$O/rr64_gather$g.json'));print(r['sec_per_slice'],r['tflops'])"; tail -2 $O/rr64_gather$g.err
done
timeout 100 python tools/rr64_check.py > $O/rr64_check.json 2> $O/rr64_check.err; echo "check rc=$?"; cat $O/rr64_check.json; tail -3 $O/rr64_check.err
