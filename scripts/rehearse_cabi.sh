BENCH_REHEARSE_CABI=1 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 5 > gpurun_out/s8.json 2> gpurun_out/s8.err; tail -5 gpurun_out/s8.err
python3 scripts/print_line.py gpurun_out/s8.json
BENCH_REHEARSE_CABI=1 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 3 --frames-in-flight 1 > gpurun_out/s8b.json 2> gpurun_out/s8b.err
python3 scripts/print_line.py gpurun_out/s8b.json
timeout -k 10 500 python -u -m pytest tests/test_gpu_pipeline.py -x -q --timeout 180 --timeout-method thread 2>&1 | tail -2
