#!/bin/bash
# Round-2 trip 1: reference drop-in on the MI355X, tr_b16 semantics, ld-pad probe, power telemetry.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
echo "== tr_probe"; timeout 60 tools/tr_probe > $OUT/tr_probe.txt 2>&1; echo "rc=$?"; head -20 $OUT/tr_probe.txt
echo "== hwmon"; ls /sys/class/drm/ 2>&1 | head; for f in /sys/class/drm/card*/device/hwmon/hwmon*/; do echo $f; ls $f | tr '\n' ' '; echo; done 2>&1 | head -20
ls /opt/rocm/bin | grep -i smi
echo "== reference drop-in"
export TN_REFERENCE_DIR=$PWD/_reference_scratch
OUT=$OUT/refdropin PER_FILE_TIMEOUT=600 timeout 1500 bash tools/reference_dropin/run_reference_tests.sh 2>&1 | tail -20
OUT=$PWD/gpurun_out
echo "== power probe"
timeout 400 python tools/power_probe.py --seconds 2.0 > $OUT/power_probe.jsonl 2> $OUT/power_probe.err; echo "rc=$?"; cat $OUT/power_probe.jsonl | cut -c1-400; tail -3 $OUT/power_probe.err
echo "== ld_pad probe"
timeout 300 python tools/ld_pad_probe.py --k 65536 --pads 0,64,128,576,4160 --iters 3 > $OUT/ld_pad.jsonl 2> $OUT/ld_pad.err; echo "rc=$?"; cat $OUT/ld_pad.jsonl; tail -3 $OUT/ld_pad.err
timeout 300 python tools/ld_pad_probe.py --k 262144 --pads 0,64,576 --iters 2 >> $OUT/ld_pad.jsonl 2>> $OUT/ld_pad.err; echo "rc=$?"; tail -6 $OUT/ld_pad.jsonl
