#!/bin/bash
mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "fused_matches" 2>&1 | tail -3
timeout 300 python tools/microbench.py --only "conv_bwd_fused[" --iters 3 \
   --fv "8,0,2;8,1,2;4,0,2;4,1,2;4,1,1;8,1,2,1;8,1,2,2;8,1,2,4;8,1,2,8;8,1,2,16;8,1,2,6;8,1,2,15;8,1,2,31" > gpurun_out/g3_micro_bwd.log 2>&1
cat gpurun_out/g3_micro_bwd.log | grep -v amdgpu.ids
cd /tmp && export TMPDIR=/tmp
for v in "8,1,2"; do
  tag=$(echo $v | tr ',' '_')
  SNET_FV_BWD=$v timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES \
     --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/g3_pmc_$tag -o r -- python $GRAFT_REPO_ROOT/tools/microbench.py --only "conv_bwd_fused[" --iters 1 > $GRAFT_REPO_ROOT/gpurun_out/g3_pmc_$tag.log 2>&1
  python $GRAFT_REPO_ROOT/tools/pmc_mfma_reduce.py $(find $GRAFT_REPO_ROOT/gpurun_out/g3_pmc_$tag -name "*counter_collection.csv" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/g3_pmc_$tag.json
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/g3_pmc_$tag
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import json
for t in ('8_1_2',):
    try:
        d = json.load(open(f'gpurun_out/g3_pmc_{t}.json'))
        for k, v in d.items():
            if 'conv_' in k:
                print(t, k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items()})
    except Exception as e:
        print(t, 'FAILED', e)
PY
