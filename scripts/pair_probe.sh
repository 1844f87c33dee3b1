#!/bin/bash
# Two independent single-rank bench processes on ONE GPU at the same time (co-resident kernels of two
# processes): used to tell a multi-process artefact of a shared GPU from a bug in the N > 1 code path.
for i in 1 2 3 4 5 6; do
  BENCH_DEBUG=1 timeout -k 10 200 python bench.py --steps 20 --warmup 2 --no-cpu-baseline > gpurun_out/pair_a_$i.json 2> gpurun_out/pair_a_$i.err &
  pa=$!
  BENCH_DEBUG=1 timeout -k 10 200 python bench.py --steps 20 --warmup 2 --no-cpu-baseline > gpurun_out/pair_b_$i.json 2> gpurun_out/pair_b_$i.err &
  pb=$!
  wait $pa || exit 1
  wait $pb || exit 1
  for s in a b; do python -c "import json; d=json.load(open('gpurun_out/pair_${s}_$i.json')); print('$i$s', d['value'], d['ms_per_step'], d['config']['rays_per_frame'], d['roofline']['avg_launch_ms'])"; done
done
