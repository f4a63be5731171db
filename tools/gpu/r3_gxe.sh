#!/bin/bash
# g_xe chunk layout: op test, A/B of the step (exp/libx_head.so = same tree built with SNET_CODEGEN_OPTS=gxestd=1), HBM write bytes
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_native_model_gpu.py -x -q -k "fused_matches or native_model_vs_engine" 2>&1 | tail -3 | tee gpurun_out/r3_gxe_tests.log
bash tools/gpu/ab_bench.sh
ROOT=$PWD; cd /tmp && export TMPDIR=/tmp
for v in head new; do
if [ $v = head ]; then export SNET_HIP_LIB=$ROOT/exp/libx_head.so; else unset SNET_HIP_LIB; fi
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $ROOT/gpurun_out/gxe_write_$v -o r -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob('$ROOT/gpurun_out/gxe_write_$v/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'conv_bwdf' in r['Kernel_Name'] or 'segment_sum' in r['Kernel_Name']:
        acc[r['Kernel_Name'][:60]].append(float(r['Counter_Value']))
for k, v in sorted(acc.items()):
    print('$v WRITE_SIZE', k, len(v), 'launches, mean', round(sum(v) / len(v) * 1024 / 1e9, 3), 'GB (counter in KB)')
PY
rm -rf $ROOT/gpurun_out/gxe_write_$v
done 2>&1 | tee $ROOT/gpurun_out/r3_gxe_write.log
