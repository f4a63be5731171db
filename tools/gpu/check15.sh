#!/bin/bash
mkdir -p gpurun_out
ROOT=$GRAFT_REPO_ROOT
timeout 200 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "fused_matches" 2>&1 | tail -3
timeout 200 python tools/microbench.py --only "conv_bwd_fused[" --iters 3 --terms 2 2>&1 | grep -v amdgpu.ids
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $ROOT/gpurun_out/g15_w -o r -- python $ROOT/tools/microbench.py --only "conv_bwd_fused[" --iters 1 --terms 2 > $ROOT/gpurun_out/g15_w.log 2>&1
cd $ROOT
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/g15_w/**/*counter_collection.csv', recursive=True)[0]
tot = {}
for r in csv.DictReader(open(f)):
    if 'conv_bwdf' in r['Kernel_Name'] and r['Counter_Name'] == 'WRITE_SIZE':
        tot.setdefault(r['Kernel_Name'][:60], []).append(float(r['Counter_Value']) * 1024 / 1e9)
print({k: [round(x, 2) for x in v] for k, v in tot.items()})
PY
rm -rf gpurun_out/g15_w
