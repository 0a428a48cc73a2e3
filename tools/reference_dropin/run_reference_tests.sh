#!/bin/bash
# Runs the reference's own pytest files through the drop-in boundary.
#   TN_REFERENCE_DIR  directory that contains the `tensornetwork` package (default /root/reference)
#   TNH_REF_BACKENDS  comma list for the `backend` fixture (default hip)
#   OUT               log directory (default gpurun_out/refdropin)
# One pytest process per reference file so that a crash in one cannot hide the others;
# writes <file>.log + summary.txt (pass/fail/skip counts per file).
set -u
REPO="$(cd "$(dirname "$0")/../.." && pwd)"
REF="${TN_REFERENCE_DIR:-/root/reference}"
OUT="${OUT:-$REPO/gpurun_out/refdropin}"
mkdir -p "$OUT"
export PYTHONPATH="$REPO/tests/golden/_stubs:$REPO/tools/reference_dropin/_stubs:$REF:$REPO:$REPO/tools/reference_dropin:${PYTHONPATH:-}"
export TNH_REF_BACKENDS="${TNH_REF_BACKENDS:-hip}"
FILES="${FILES:-tensornetwork/tests/split_node_test.py
tensornetwork/tests/network_operations_test.py
tensornetwork/tests/ncon_interface_test.py
tensornetwork/contractors/opt_einsum_paths/path_contractors_node_test.py
tensornetwork/tests/tensornetwork_test.py
tensornetwork/tests/network_test.py
tensornetwork/tests/network_components_free_test.py
tensornetwork/tests/tensor_test.py
tensornetwork/linalg/tests/test_operations.py
tensornetwork/linalg/tests/test_linalg.py
tensornetwork/linalg/tests/initialization_test.py
tensornetwork/linalg/tests/node_linalg_test.py}"
: > "$OUT/summary.txt"
echo "reference: $REF   backends: $TNH_REF_BACKENDS   $(date -u +%FT%TZ)" >> "$OUT/summary.txt"
cd "$REF" || exit 2
for f in $FILES; do
  name="$(echo "$f" | tr '/' '_')"
  timeout "${PER_FILE_TIMEOUT:-900}" python -m pytest --noconftest -p tnh_ref_plugin -p no:cacheprovider \
      -q --maxfail=1000 -o addopts="" --timeout=300 -rfEs "$f" > "$OUT/$name.log" 2>&1
  rc=$?
  tail -1 "$OUT/$name.log" | sed "s|^|$f  rc=$rc  |" >> "$OUT/summary.txt"
done
cat "$OUT/summary.txt"
