#!/bin/bash
# Fast GPU iteration for the solver: native parity sweep (n <= 256 and 2048), then the K3 anatomy.
# A step that times out stops the chain (no GPU step after a killed one).
set -u
mkdir -p gpurun_out
: > gpurun_out/iter_summary.log
step() {
  local name=$1 tmo=$2; shift 2
  echo "=== $name" | tee -a gpurun_out/iter_summary.log
  timeout -k 10 "$tmo" "$@" > "gpurun_out/$name.log" 2>&1
  local rc=$?
  echo "rc=$rc" | tee -a gpurun_out/iter_summary.log
  tail -n "${TAILN:-8}" "gpurun_out/$name.log" | tee -a gpurun_out/iter_summary.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a gpurun_out/iter_summary.log; exit 1; fi
  if [ $rc -ne 0 ] && [ "${STOP_ON_FAIL:-1}" = "1" ]; then echo "FAILED $name: stopping" | tee -a gpurun_out/iter_summary.log; exit 1; fi
  return 0
}
for s in "$@"; do
  case $s in
    p256)  step it_p256 300 tests/native/_build/parity_driver 256 2 ;;
    p2k)   step it_p2k 400 tests/native/_build/parity_driver 2048 1 ;;
    diag)  TAILN=48 step it_diag 300 python tools/diag_k3.py 32 2048 0 ;;
    bench) step it_bench 400 python bench.py --steps 5 --warmup 2 ;;
    benchq) step it_benchq 300 python bench.py --steps 5 --warmup 2 --cpu-sample 0 ;;
    k2)    step it_k2 300 python bench.py --config K2 --steps 5 --warmup 2 --cpu-sample 0 ;;
    feat)  step it_feat 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "feature or golden or wrappers or k2_config or k3_config" ;;
    prof)  cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; step it_prof 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_iter -o r02 -- python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-node-baseline --no-overlap ;;
    dense) step it_dense 300 python tools/diag_dense.py uniform && step it_dense_sparse 300 python tools/diag_dense.py sparse && step it_dense_ties 300 python tools/diag_dense.py ties ;;
    pytest) step it_pytest 1100 python -m pytest tests -x -q -m gpu ;;
    smoke) step it_smoke 300 python -c "import __graft_entry__ as g; g.smoke()" ;;
    *) echo "unknown step $s" ;;
  esac
done
