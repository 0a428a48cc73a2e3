#!/bin/bash
# The GPU trips of round 6 in one file (each was one `gpurun -- bash tools/r6_tripN.sh` call; profiles/README.md cites them
# by number).  usage: gpurun --timeout S -- 'bash tools/r6_trips.sh N'
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
case "${1:-}" in
1)   # Round 6, trip 1: GEMM kernel tests under the tightened bf16 / f16 rule + the BASELINE-size tests, the band SVD's
  timeout 1500 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 900 -x -k "gemm or config2 or d512" > $OUT/t1_pytest_gemm.log 2>&1; echo "pytest rc=$?"
  tail -15 $OUT/t1_pytest_gemm.log
  rm -rf $OUT/prof_svd_f32 $OUT/prof_mps
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_svd_f32 -o svd -- python $OUT/../tools/svd_stats_run.py f32 > $OUT/t1_svd_f32.log 2>&1; echo "svd prof rc=$?")
  find $OUT/prof_svd_f32 -name "*kernel_trace.csv" -delete
  python tools/svd_stats_summary.py $OUT $OUT
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_mps -o mps -- python $OUT/../tests/perf_mps_chain.py --D 512 --d 2 > $OUT/t1_mps.log 2>&1; echo "mps prof rc=$?")
  cat $OUT/t1_mps.log | tail -3
  python tools/kernel_stats.py $OUT/prof_mps | head -25
  find $OUT/prof_mps -name "*kernel_trace.csv" | head -2
  ;;
2)   # Round 6, trip 2: the fast band reduction -- band SVD tests, the A/B probe, a kernel table.
  timeout 600 python tools/svd_fast_probe.py > $OUT/t2_fast_probe.jsonl 2> $OUT/t2_fast_probe.err; echo "probe rc=$?"
  cat $OUT/t2_fast_probe.jsonl | cut -c1-400; tail -5 $OUT/t2_fast_probe.err
  timeout 1500 python -m pytest tests/test_gpu_svd_band.py tests/test_gpu_linalg.py -m gpu -q --timeout 900 > $OUT/t2_pytest_svd.log 2>&1; echo "pytest rc=$?"
  tail -15 $OUT/t2_pytest_svd.log
  rm -rf $OUT/prof_svd_f32
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_svd_f32 -o svd -- python $OUT/../tools/svd_stats_run.py f32 > $OUT/t2_svd_f32.log 2>&1; echo "svd prof rc=$?")
  find $OUT/prof_svd_f32 -name "*kernel_trace.csv" -delete
  python tools/svd_stats_summary.py $OUT $OUT
  cat $OUT/svd_band_f32_kernel_stats.txt
  ;;
3)   # Round 6, trip 3: fast stage 1 with the scalar-operand reduce kernels; tile knobs; new boundary tests; MPS chain shapes.
  timeout 300 python tools/svd_fast_probe.py --sizes 4096x4096,2048x2048,1024x1024,512x512 --spectra 1 > $OUT/t3_fast_probe.jsonl 2> $OUT/t3_fast_probe.err; echo "probe rc=$?"
  cut -c1-230 $OUT/t3_fast_probe.jsonl; tail -3 $OUT/t3_fast_probe.err
  for knob in "TNH_SVDB_FAST_CW=64" "TNH_SVDB_FAST_CW=128" "TNH_SVDB_FAST_WGS=512" "TNH_SVDB_FAST_WGS=2048" "TNH_SVDB_FAST_SWITCH=64" "TNH_SVDB_FAST_SWITCH=256"; do
    echo "== $knob"; env $knob timeout 200 python tools/svd_fast_probe.py --sizes 4096x4096,2048x2048 --spectra 0 --check 0 2>&1 | grep '"fast_env": 1' | cut -c1-200
  done | tee $OUT/t3_knobs.txt
  timeout 900 python -m pytest tests/test_gpu_svd_band.py tests/test_gpu_linalg.py -m gpu -q --timeout 900 > $OUT/t3_pytest_svd.log 2>&1; echo "pytest svd rc=$?"; tail -4 $OUT/t3_pytest_svd.log
  timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 600 -k "signature or high_rank" > $OUT/t3_pytest_boundary.log 2>&1; echo "pytest boundary rc=$?"; tail -8 $OUT/t3_pytest_boundary.log
  rm -rf $OUT/prof_svd_f32
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_svd_f32 -o svd -- python $OUT/../tools/svd_stats_run.py f32 > $OUT/t3_svd_f32.log 2>&1; echo "svd prof rc=$?")
  find $OUT/prof_svd_f32 -name "*kernel_trace.csv" -delete
  python tools/svd_stats_summary.py $OUT $OUT | tail -25
  timeout 300 python tools/mps_chain_shapes.py > $OUT/t3_mps_shapes.jsonl 2>&1; echo "mps shapes rc=$?"; cat $OUT/t3_mps_shapes.jsonl | cut -c1-200
  ;;
4)   # Round 6, trip 4: reduce kernels shared by four waves (ds_read_b128 operands).
  timeout 300 python tools/svd_fast_probe.py --sizes 4096x4096,2048x2048,1024x1024,512x512,3072x1024,1000x600 --spectra 1 > $OUT/t4_fast_probe.jsonl 2> $OUT/t4_fast_probe.err; echo "probe rc=$?"
  cut -c1-230 $OUT/t4_fast_probe.jsonl; tail -3 $OUT/t4_fast_probe.err
  timeout 900 python -m pytest tests/test_gpu_svd_band.py tests/test_gpu_linalg.py -m gpu -q --timeout 900 > $OUT/t4_pytest_svd.log 2>&1; echo "pytest svd rc=$?"; tail -4 $OUT/t4_pytest_svd.log
  rm -rf $OUT/prof_svd_f32
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_svd_f32 -o svd -- python $OUT/../tools/svd_stats_run.py f32 > $OUT/t4_svd_f32.log 2>&1; echo "svd prof rc=$?")
  find $OUT/prof_svd_f32 -name "*kernel_trace.csv" -delete
  python tools/svd_stats_summary.py $OUT $OUT | tail -22
  ;;
5)   # Round 6, trip 5: reduce kernels with batched partial loads; skinny split-K on the MPS chain.
  timeout 300 python tools/svd_fast_probe.py --sizes 4096x4096,2048x2048,1024x1024,512x512 --spectra 0 > $OUT/t5_fast_probe.jsonl 2> $OUT/t5_fast_probe.err; echo "probe rc=$?"
  cut -c1-200 $OUT/t5_fast_probe.jsonl; tail -3 $OUT/t5_fast_probe.err
  timeout 900 python -m pytest tests/test_gpu_svd_band.py -m gpu -q --timeout 900 -x > $OUT/t5_pytest_svd.log 2>&1; echo "pytest svd rc=$?"; tail -4 $OUT/t5_pytest_svd.log
  timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 900 -k "split_k" > $OUT/t5_pytest_splitk.log 2>&1; echo "pytest splitk rc=$?"; tail -8 $OUT/t5_pytest_splitk.log
  rm -rf $OUT/prof_svd_f32
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_svd_f32 -o svd -- python $OUT/../tools/svd_stats_run.py f32 > $OUT/t5_svd_f32.log 2>&1; echo "svd prof rc=$?")
  find $OUT/prof_svd_f32 -name "*kernel_trace.csv" -delete
  python tools/svd_stats_summary.py $OUT $OUT | tail -22
  timeout 300 python tests/perf_mps_chain.py --D 512 --d 2,4 > $OUT/t5_mps.log 2>&1; echo "mps rc=$?"; tail -3 $OUT/t5_mps.log | cut -c1-300
  timeout 300 python tools/mps_chain_shapes.py > $OUT/t5_mps_shapes.jsonl 2>&1; tail -12 $OUT/t5_mps_shapes.jsonl | cut -c1-200
  ;;
6)   # Round 6, trip 6: tiny-output kernel, host overhead of small steps.
  timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 900 -k "split_k or tiny" > $OUT/t6_pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/t6_pytest.log
  timeout 300 python tests/perf_mps_chain.py --D 512 --d 2,4 > $OUT/t6_mps.log 2>&1; echo "mps rc=$?"; tail -3 $OUT/t6_mps.log | cut -c1-300
  timeout 600 python tools/host_overhead_probe.py > $OUT/t6_host.txt 2>&1; echo "host rc=$?"; head -70 $OUT/t6_host.txt | cut -c1-200; grep -n "mps chain" -A60 $OUT/t6_host.txt | cut -c1-200
  ;;
7)   # Round 6, trip 7: index_update with a tensor assignee on the GPU; the Sturm kernel choice for small rounds.
  timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 600 -k "index_update or tiny or signature or high_rank" > $OUT/t7_pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/t7_pytest.log
  timeout 300 python tools/svd_fast_probe.py --sizes 512x512,768x768,1024x1024,1000x600,4096x512,4096x4096 --spectra 0 > $OUT/t7_fast_probe.jsonl 2> $OUT/t7_fast_probe.err; echo "probe rc=$?"
  cut -c1-200 $OUT/t7_fast_probe.jsonl; tail -3 $OUT/t7_fast_probe.err
  echo "== TNH_SVDB_LANE_AUTO=0"; TNH_SVDB_LANE_AUTO=0 timeout 300 python tools/svd_fast_probe.py --sizes 512x512,768x768,1024x1024,1000x600 --spectra 0 --check 0 2>&1 | grep '"fast_env": 1' | cut -c1-160
  timeout 900 python -m pytest tests/test_gpu_svd_band.py tests/test_gpu_linalg.py tests/test_gpu_mps.py -m gpu -q --timeout 900 > $OUT/t7_pytest_svd.log 2>&1; echo "pytest svd rc=$?"; tail -4 $OUT/t7_pytest_svd.log
  ;;
8)   # Round 6, trip 8: the skinny f32 / f64 kernel; MPS chain; the whole kernel test file.
  timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 900 -k "skinny or split_k or tiny" > $OUT/t8_pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $OUT/t8_pytest.log
  timeout 300 python tests/perf_mps_chain.py --D 512 --d 2,4 > $OUT/t8_mps.log 2>&1; echo "mps rc=$?"; tail -3 $OUT/t8_mps.log | cut -c1-300
  timeout 300 python tools/mps_chain_shapes.py > $OUT/t8_mps_shapes.jsonl 2>&1; tail -48 $OUT/t8_mps_shapes.jsonl | cut -c1-160
  timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_mps.py tests/test_gpu_workloads.py tests/test_gpu_graph.py -m gpu -q --timeout 900 > $OUT/t8_pytest_all.log 2>&1; echo "pytest all rc=$?"; tail -12 $OUT/t8_pytest_all.log
  ;;
9)   # Round 6, trip 9: tiny kernel with loads in flight; the d = 4 chain's shapes.
  timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 600 -k "tiny or split_k" > $OUT/t9_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/t9_pytest.log
  timeout 300 python tests/perf_mps_chain.py --D 512 --d 2,4 > $OUT/t9_mps.log 2>&1; echo "mps rc=$?"; tail -3 $OUT/t9_mps.log | cut -c1-300
  timeout 300 python tools/mps_chain_shapes.py --d 4 > $OUT/t9_mps_shapes_d4.jsonl 2>&1; tail -48 $OUT/t9_mps_shapes_d4.jsonl | cut -c1-160
  ;;
10)   # Round 6, trip 10: what the driver runs -- smoke, the whole -m gpu suite, bench.py --gpus 1 --steps 20 --warmup 5.
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/t10_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/t10_smoke.log | cut -c1-300
  timeout 2400 python -m pytest tests -m gpu -q --timeout 900 > $OUT/t10_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/t10_pytest_gpu.log
  timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/t10_bench.json 2> $OUT/t10_bench.err; echo "bench rc=$?"
  tail -1 $OUT/t10_bench.json | cut -c1-4000; tail -3 $OUT/t10_bench.err
  cp bench_detail.json $OUT/t10_bench_detail.json 2>/dev/null
  ;;
11)   # Round 6, trip 11: the fast band reduction for f64 input.
  timeout 600 python tools/svd_fast_probe.py --dtype f64 --sizes 1024x1024,2048x2048,4096x4096,512x512,3072x1024,1000x600 --spectra 1 --reps 3 > $OUT/t11_fast_probe_f64.jsonl 2> $OUT/t11_fast_probe_f64.err; echo "probe rc=$?"
  cut -c1-330 $OUT/t11_fast_probe_f64.jsonl; tail -5 $OUT/t11_fast_probe_f64.err
  timeout 300 python tools/svd_fast_probe.py --sizes 4096x4096,1024x1024 --spectra 0 --reps 3 2>&1 | cut -c1-200
  timeout 1200 python -m pytest tests/test_gpu_svd_band.py tests/test_gpu_linalg.py tests/test_gpu_mps.py -m gpu -q --timeout 900 > $OUT/t11_pytest_svd.log 2>&1; echo "pytest svd rc=$?"; tail -15 $OUT/t11_pytest_svd.log
  ;;
12)   # Round 6, trip 12: kernel table of the f64 band SVD with the fast stage.
  rm -rf $OUT/prof_svd_f32 $OUT/prof_svd_f64
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_svd_f64 -o svd -- python $OUT/../tools/svd_stats_run.py f64 > $OUT/t12_svd_f64.log 2>&1; echo "svd prof rc=$?")
  find $OUT/prof_svd_f64 -name "*kernel_trace.csv" -delete
  python tools/svd_stats_summary.py $OUT $OUT | tail -30
  ;;
*) echo "usage: $0 <trip number 1..12>"; exit 2;;
esac
