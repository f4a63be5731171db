#!/bin/bash
# stochastic PC sampling of one fused kernel (gfx950: every sample carries the wave's issue / stall reason):
#   gpurun --timeout 900 -- bash tools/gpu/pc_sampling.sh "conv_bwd_fused[22" sevennet_0 1
ONLY=${1:-conv_bwd_fused[22}; MODEL=${2:-sevennet_0}; LAYER=${3:-1}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pcs; rm -rf $OUT; mkdir -p $OUT
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
for method in stochastic host_trap; do
  unit=cycles; iv=65536; [ $method = host_trap ] && { unit=time; iv=100; }
  timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $method --pc-sampling-unit $unit --pc-sampling-interval $iv \
    --kernel-trace --output-format csv -d $OUT/$method -- python $R/tools/microbench.py --model $MODEL --layer $LAYER --terms 4 --iters 3 --only "$ONLY" > $OUT/$method.log 2>&1
  echo "$method rc=$?"; tail -3 $OUT/$method.log | cut -c1-200
  find $OUT/$method -name "*pc_sampling*" | head -5
  [ -n "$(find $OUT/$method -name '*pc_sampling*csv' -size +1k | head -1)" ] && break
done
# keep the merged-back data small: per-(kernel, offset, reason) histogram instead of raw samples
python - <<PY
import csv, glob, collections, os, sys
for f in glob.glob('$OUT/*/**/*pc_sampling*.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    print(f, len(rows), 'samples; columns', list(rows[0].keys()) if rows else None)
    if not rows: continue
    keyf = [k for k in rows[0].keys()]
    h = collections.Counter()
    for r in rows:
        h[tuple(r.get(k, '') for k in keyf if k.lower() not in ('sample_timestamp', 'timestamp', 'exec_mask', 'dispatch_id', 'correlation_id', 'wave_in_group', 'chiplet', 'hw_id', 'workgroup_id_x', 'workgroup_id_y', 'workgroup_id_z', 'wave_id', 'cu_id', 'simd_id', 'exec', 'wave_count'))] += 1
    with open(f.replace('.csv', '_hist.tsv'), 'w') as o:
        o.write('\t'.join(k for k in keyf if k.lower() not in ('sample_timestamp', 'timestamp', 'exec_mask', 'dispatch_id', 'correlation_id', 'wave_in_group', 'chiplet', 'hw_id', 'workgroup_id_x', 'workgroup_id_y', 'workgroup_id_z', 'wave_id', 'cu_id', 'simd_id', 'exec', 'wave_count')) + '\tcount\n')
        for k, v in h.most_common():
            o.write('\t'.join(k) + f'\t{v}\n')
    os.remove(f)
PY
du -sh $OUT; ls -R $OUT | head -30
