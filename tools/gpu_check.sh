#!/bin/bash
# Runs on the GPU box via gpurun: GPU test-suite, native parity sweep, smoke and a short bench.
# A step that times out (124/137) stops the chain: no further GPU step after a killed one.
set -u
mkdir -p gpurun_out
TAG=${TAG:-r03}
step() {  # name timeout cmd...
  local name=$1 tmo=$2; shift 2
  echo "=== $name" | tee -a gpurun_out/summary.log
  timeout -k 10 "$tmo" "$@" > "gpurun_out/$name.log" 2>&1
  local rc=$?
  echo "rc=$rc" | tee -a gpurun_out/summary.log
  tail -n 12 "gpurun_out/$name.log" | tee -a gpurun_out/summary.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a gpurun_out/summary.log; exit 1; fi
  return 0
}
: > gpurun_out/summary.log
for s in "$@"; do
  case $s in
    smoke)   step smoke 300 python -c "import __graft_entry__ as g; g.smoke()" ;;
    pytest)  step pytest 900 python -m pytest tests -x -q -m gpu ;;
    parity2k) step parity2k 600 tests/native/_build/parity_driver 2048 1 ;;
    parity4k) step parity4k 900 tests/native/_build/parity_driver 4096 1 ;;
    bench)   step bench 600 python bench.py --steps 10 --warmup 3 ;;
    bench_serial) step bench_serial 400 python bench.py --steps 5 --warmup 2 --no-overlap --cpu-sample 0 --no-node-baseline ;;
    bench_k2) step bench_k2 300 python bench.py --config K2 --steps 10 --warmup 3 --cpu-sample 0 ;;
    bench_k4) step bench_k4 400 python bench.py --config K4 --steps 3 --warmup 1 --cpu-sample 0 ;;
    bench_k5) step bench_k5 600 python bench.py --config K5 --steps 2 --warmup 1 --cpu-sample 1 ;;
    bench_8192) step bench_8192 300 python bench.py --config K5 --n 8192 --batch 4 --steps 2 --warmup 1 --cpu-sample 1 ;;
    prof_k5) export TMPDIR=/tmp; step prof_k5 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_k5 -o $TAG -- python3 bench.py --config K5 --steps 1 --warmup 1 --cpu-sample 0 ;;
    coop_stamps) LAPWARM_HIP_LIB=$PWD/gnn-accelerated-lap-warm-start-pipeline_amd/liblapwarm_hip_coopstamps.so DIAG_CHECK=0 step coop_stamps 300 python tools/diag_coop.py 8192 1 uniform 2 ;;
    bench_inflight) step bench_inflight 400 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-node-baseline --inflight 8 ;;
    bench256) step bench256 300 python bench.py --steps 2 --warmup 1 --threads-hint 256 --cpu-sample 0 ;;
    bench1024) step bench1024 300 python bench.py --steps 2 --warmup 1 --threads-hint 1024 --cpu-sample 0 ;;
    diag)    step diag 600 python tools/diag_k3.py 32 2048 1024 256 ;;
    prof)    export TMPDIR=/tmp; step prof 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o $TAG -- python3 bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-node-baseline ;;
    hostapi) step hostapi 600 python tools/diag_hostapi.py ;;
    pmc)     export TMPDIR=/tmp
             step pmc_fetch 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o $TAG -- python3 bench.py --steps 1 --warmup 1 --cpu-sample 0 --no-node-baseline --no-overlap
             step pmc_write 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o $TAG -- python3 bench.py --steps 1 --warmup 1 --cpu-sample 0 --no-node-baseline --no-overlap ;;
    pmc_sq)  export TMPDIR=/tmp
             step pmc_sq 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_sq -o $TAG -- python3 bench.py --steps 1 --warmup 1 --cpu-sample 0 --no-node-baseline --no-overlap ;;
    pmc_k5)  export TMPDIR=/tmp
             step pmc_k5_fetch 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_k5_fetch -o $TAG -- python3 bench.py --config K5 --n 8192 --batch 1 --steps 1 --warmup 1 --cpu-sample 0 --no-overlap
             step pmc_k5_sq 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d gpurun_out/pmc_k5_sq -o $TAG -- python3 bench.py --config K5 --n 8192 --batch 1 --steps 1 --warmup 1 --cpu-sample 0 --no-overlap ;;
    *) echo "unknown step $s" ;;
  esac
done
