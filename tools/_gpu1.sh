set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "fused" 2>&1 | tail -25 > gpurun_out/g1_fused_ops.log
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "fused_engine" 2>&1 | tail -25 > gpurun_out/g1_fused_engine.log
timeout 600 python tools/microbench.py --only fused --iters 5 > gpurun_out/g1_micro.log 2>&1
timeout 600 python tools/microbench.py --only hidden --iters 5 >> gpurun_out/g1_micro.log 2>&1
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/g1_bench_fused.json 2> gpurun_out/g1_bench_fused.err
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --fused off > gpurun_out/g1_bench_off.json 2> gpurun_out/g1_bench_off.err
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/g1_all.log
tail -5 gpurun_out/g1_fused_ops.log gpurun_out/g1_fused_engine.log gpurun_out/g1_all.log; cat gpurun_out/g1_micro.log
