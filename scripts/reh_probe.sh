#!/bin/bash
# N = 2 rehearsal on one GPU (gloo), repeated; prints value, ms/step, rays/frame and the assembled-frame check per run.
# mode: overlap (default bench path) | sync | pack
mode=${1:-overlap}; n=${2:-8}
for i in $(seq 1 $n); do
  if [ "$mode" != overlap ]; then export BENCH_XCHG=$mode; fi
  BENCH_DEBUG=1 BENCH_DIST_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
     --master-addr 127.0.0.1 --master-port $((29700 + i)) bench.py --gpus 2 --steps 5 --warmup 1 2>gpurun_out/reh_${mode}_$i.err | tail -1 | \
     python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$mode', $i, d['value'], d['ms_per_step'], d['config']['rays_per_frame'], d['config']['assembled_frames_equal_whole_frames'])" || exit 1
done
