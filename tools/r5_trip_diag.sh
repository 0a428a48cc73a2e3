#!/bin/bash
# round-5 diagnostic trip: where the time of the D = 96 step goes (tools/alloc_gc_probe.py: default policy vs
# the amortised collector passes on / off vs bare tensordot, before and after the sliced-network leg), then the GPU tests that reach
# tensornetwork_amd/distributed.py (changed after tools/r5_final.sh ran: grid partitions)
O=gpurun_out/r5diag; mkdir -p $O
timeout 300 python tools/alloc_gc_probe.py --after-sliced 96 64 > $O/alloc_gc_probe.jsonl 2> $O/alloc_gc_probe.err; echo "probe rc=$?"
python - <<'PY'
import json
for line in open("gpurun_out/r5diag/alloc_gc_probe.jsonl"):
  r = json.loads(line)
  if "variant" in r:
    print(r["when"][:5], r["D"], r["layout"], r["variant"].ljust(18), "wall %.3f/%.3f host %.3f gemm %s tf %.0f skipped %d full %d pass_ms %.3f objs %d pool %s" % (
      r["wall_ms_best"], r["wall_ms_median"], r["host_ms_best"], r["gemm_event_ms"] and round(r["gemm_event_ms"], 3), r["tflops_best"],
      r["passes_skipped_for_slack"], r["full_passes"], r["full_pass_ms_now"], r["tracked_objects"], r["pool_after"]))
  else:
    print(r)
PY
tail -3 $O/alloc_gc_probe.err
timeout 600 python -m pytest tests/test_gpu_workloads.py -m gpu -q --timeout 600 -k "sliced or distributed or collector or gc" > $O/pytest_gpu_sliced.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu_sliced.txt
