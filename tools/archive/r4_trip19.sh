#!/bin/bash
# Gather GEMM with the incremental chunk decode: parity, per-shape timings, the D = 12 network (timing + partials check).
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4t19; mkdir -p $O
timeout 100 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k gemm_gather --timeout 90 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 120 python tools/gather_probe.py > $O/gather_probe.jsonl 2> $O/gather_probe.err; echo "probe rc=$?"; python - <<'PY'
import json
for line in open("gpurun_out/r4t19/gather_probe.jsonl"):
  r = json.loads(line)
  print(r["case"], r["plan"], {k: v["us"] for k, v in r.items() if isinstance(v, dict) and "us" in v})
PY
run() {  # tag, env...
  tag=$1; shift
  env "$@" timeout 100 python tools/rr64_probe.py --D 12 --max-slices 8 > $O/rr64_$tag.json 2> $O/rr64_$tag.err; echo "rr64 $tag rc=$?"; python -c "import json;r=json.load(open('$O/rr64_$tag.json'));print(r['sec_per_slice'],r['tflops'])"; tail -2 $O/rr64_$tag.err
}
run classic TNH_GATHER_GEMM=0
run gather TNH_GATHER_GEMM=1
run gather_p256 TNH_GATHER_GEMM=1 TNH_GATHER_MIN_PIECE=256
TNH_GATHER_GEMM=1 timeout 150 python tools/rr64_check.py > $O/rr64_check_gather.json 2> $O/rr64_check_gather.err; echo "check rc=$?"; cat $O/rr64_check_gather.json; tail -3 $O/rr64_check_gather.err
