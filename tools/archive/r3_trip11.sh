#!/bin/bash
# QR on 16-wide panels + split guard + band SVD regression
set -u
O=gpurun_out/${1:-r3t11}
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_linalg.py tests/test_gpu_svd_band.py -q --timeout 600 > $O/pytest_a.log 2>&1; echo "pytest linalg+band rc=$?" | tee $O/trip.log
tail -12 $O/pytest_a.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -q --timeout 600 -k "split or f32" > $O/pytest_b.log 2>&1; echo "pytest split rc=$?" | tee -a $O/trip.log
tail -8 $O/pytest_b.log
timeout 300 python tools/qr_sizes_probe.py > $O/qr_new.jsonl 2>> $O/probe.err; echo "qr probe rc=$?"; cat $O/qr_new.jsonl
TNH_QR_PANEL16=0 timeout 300 python tools/qr_sizes_probe.py > $O/qr_old.jsonl 2>> $O/probe.err; echo "qr old rc=$?"; cat $O/qr_old.jsonl
tail -3 $O/probe.err
