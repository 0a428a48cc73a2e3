#!/bin/bash
set -u
python - <<'PY'
import json, sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import tensornetwork_amd as ta
import bench
be = ta.get_hip_backend()
for order in ("16", "32", "64", "128"):
  os.environ["TNH_PERMUTE_ORDER"] = order
  for na in ("", "1"):
    if na: os.environ["TNH_PERMUTE_NA2"] = "1"
    else: os.environ.pop("TNH_PERMUTE_NA2", None)
    rows = bench.helpers_bench(ta, be)[:3]
    print("order", order, "NA2" if na else "NA1", " ".join("%6.0f" % r["gbps"] for r in rows), flush=True)
PY
