#!/bin/bash
# The N > 1 route of bench.py with ONE rank (one-rank RCCL communicator) on a one-GPU box: default (3 in flight), one frame at a
# time, entries of 2 frames with 3 and with 1 in flight; then the comm and pipeline tests.
cd "$GRAFT_REPO_ROOT"
for extra in "" "--frames-in-flight 1" "--frames-per-entry 2" "--frames-per-entry 2 --frames-in-flight 1"; do
  BENCH_REHEARSE_CABI=1 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 3 $extra > gpurun_out/s8.json 2> gpurun_out/s8.err || { tail -5 gpurun_out/s8.err; exit 1; }
  echo "[$extra] $(python3 scripts/print_line.py gpurun_out/s8.json)"
done
timeout -k 10 500 python -u -m pytest tests/test_gpu_comm.py tests/test_gpu_pipeline.py -x -q --timeout 180 --timeout-method thread 2>&1 | tail -3
