#!/bin/bash
mkdir -p gpurun_out
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for h in python native; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/g14_$h -o r -- python $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --host $h > $ROOT/gpurun_out/g14_$h.log 2>&1
  cp $(find $ROOT/gpurun_out/g14_$h -name "*kernel_stats.csv" | head -1) $ROOT/gpurun_out/g14_${h}_kernel_stats.csv
  rm -rf $ROOT/gpurun_out/g14_$h
done
cd $ROOT
python - <<'PY'
import csv
def load(f):
    d = {}
    for r in csv.DictReader(open(f)):
        d[r['Name'].split('(anonymous namespace)::')[-1][:70]] = (int(r['Calls']), float(r['AverageNs']) / 1e6, float(r['TotalDurationNs']) / 1e6)
    return d
a, b = load('gpurun_out/g14_python_kernel_stats.csv'), load('gpurun_out/g14_native_kernel_stats.csv')
tot_a = sum(v[2] for v in a.values()); tot_b = sum(v[2] for v in b.values())
print('total kernel ms python', round(tot_a, 1), 'native', round(tot_b, 1))
for k in sorted(set(a) | set(b), key=lambda k: -(a.get(k, (0, 0, 0))[2])):
    va, vb = a.get(k, (0, 0, 0)), b.get(k, (0, 0, 0))
    if max(va[2], vb[2]) > 3.0:
        print(f'{k:72s} py calls {va[0]:4d} avg {va[1]:7.3f} | nat calls {vb[0]:4d} avg {vb[1]:7.3f}')
PY
