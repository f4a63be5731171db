#!/usr/bin/env python3
"""Reduce a rocprofv3 counter pass (SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE, SQ_WAVE_CYCLES ...)
to per-kernel matrix-pipe utilisation.

    python tools/pmc_mfma_reduce.py <counter_collection.csv> > out.json

Per kernel (averaged over its launches): the raw counters, the dispatch duration from the same rows'
timestamps, and
    mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (duration_s * clock_hz * 1024 SIMDs)
SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe busy cycles summed over all SIMDs of the chip
(/opt/skills/guides/MI355X_MICROARCH.md: = 32 x N_mfma for v_mfma_f32_32x32x16_bf16, 16 x N for the
16x16x32 form).  The shader clock is taken from GRBM_GUI_ACTIVE / duration when that counter is in the
pass (it is reported per XCD-sum or per device depending on the rocprofv3 build: both readings are
emitted, `clock_ghz_if_per_xcd_sum` = value / 8 / duration), else 2.4 GHz (max clock; DVFS typically
runs these kernels at 1.9-2.3 GHz, so the fraction is then a lower bound).
"""
import csv
import json
import sys
from collections import defaultdict

N_SIMD = 256 * 4


def main():
    vals = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    dur = defaultdict(float)
    seen = defaultdict(set)
    with open(sys.argv[1], newline='') as f:
        for row in csv.DictReader(f):
            name = row['Kernel_Name'].split('(anonymous namespace)::')[-1].split('(')[0][:60]
            c = row['Counter_Name']
            vals[name][c] += float(row['Counter_Value'])
            cnt[name][c] += 1
            did = row['Dispatch_Id']
            if did not in seen[name]:
                seen[name].add(did)
                dur[name] += (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) * 1e-9
    out = {}
    for k in sorted(vals):
        n = len(seen[k])
        d = dur[k] / n
        rec = {c: vals[k][c] / cnt[k][c] for c in vals[k]}
        rec['launches'] = n
        rec['avg_duration_ms'] = d * 1e3
        clock = 2.4e9
        if 'GRBM_GUI_ACTIVE' in rec and d > 0:
            rec['clock_ghz_if_device'] = rec['GRBM_GUI_ACTIVE'] / d / 1e9
            rec['clock_ghz_if_per_xcd_sum'] = rec['GRBM_GUI_ACTIVE'] / 8 / d / 1e9
            for key in ('clock_ghz_if_device', 'clock_ghz_if_per_xcd_sum'):
                if 1.0 <= rec[key] <= 2.6:
                    clock = rec[key] * 1e9
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in rec and d > 0:
            rec['mfma_busy_frac'] = rec['SQ_VALU_MFMA_BUSY_CYCLES'] / (d * clock * N_SIMD)
            rec['clock_ghz_used'] = clock / 1e9
        if 'SQ_BUSY_CYCLES' in rec and rec['SQ_BUSY_CYCLES'] > 0 and 'SQ_VALU_MFMA_BUSY_CYCLES' in rec:
            rec['mfma_busy_over_sq_busy'] = rec['SQ_VALU_MFMA_BUSY_CYCLES'] / rec['SQ_BUSY_CYCLES']
        out[k] = rec
    json.dump(out, sys.stdout, indent=1)


if __name__ == '__main__':
    main()
