#!/bin/bash
# every kernel dispatch of ONE step in order with its duration (rocprofv3 --kernel-trace, csv), native host (one stream)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/tl && rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --host ${HOST:-python} > /dev/null 2>&1
F=$(find /tmp/tl -name "*kernel_trace.csv" | head -1)
python - "$F" <<'PY' > $R/gpurun_out/step_timeline.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last step = from the last edge_vectors_kernel / edge_fwd_kernel on
starts = [i for i, r in enumerate(rows) if 'edge_fwd_kernel' in r['Kernel_Name']]
i0 = starts[-1]
t0 = int(rows[i0]['Start_Timestamp'])
prev_end = t0
tot = 0
for r in rows[i0:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    print(f"{(s - t0) / 1e3:10.1f} us  gap {(s - prev_end) / 1e3:7.1f}  dur {(e - s) / 1e3:8.1f}  {name[:70]}  grid {r.get('Grid_Size_X', r.get('Grid_Size', ''))} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size', ''))}")
    prev_end = max(prev_end, e); tot += e - s
print('sum of durations', tot / 1e6, 'ms; span', (prev_end - t0) / 1e6, 'ms')
PY
tail -3 $R/gpurun_out/step_timeline.txt
