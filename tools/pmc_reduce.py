#!/usr/bin/env python3
"""Reduce two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE) to HBM bytes per launch per kernel.

    python tools/pmc_reduce.py <fetch_counter_collection.csv> <write_counter_collection.csv> > out.json

Units and the gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3
section): both counters are in KB; FETCH_SIZE under-counts 128-B read requests as 64 B on gfx950,
so reads are multiplied by 2; WRITE_SIZE needs no correction (it reproduces the known w / g_w byte
counts exactly).  Kernel names are cut at the first '(' and long template names at 60 characters.
"""
import csv
import json
import sys
from collections import defaultdict


def per_kernel(path, counter):
    tot, n = defaultdict(float), defaultdict(int)
    with open(path, newline='') as f:
        for row in csv.DictReader(f):
            if row['Counter_Name'] != counter:
                continue
            name = row['Kernel_Name'].split('(anonymous namespace)::')[-1].split('(')[0][:60]
            tot[name] += float(row['Counter_Value'])
            n[name] += 1
    return {k: (tot[k] / n[k], n[k]) for k in tot}


def main():
    fetch = per_kernel(sys.argv[1], 'FETCH_SIZE')
    write = per_kernel(sys.argv[2], 'WRITE_SIZE')
    out = {}
    for k in sorted(set(fetch) | set(write)):
        fr = fetch.get(k, (0.0, 0))[0] * 1024.0
        wr = write.get(k, (0.0, 0))[0] * 1024.0
        out[k] = dict(fetch_bytes_raw=fr, fetch_bytes_corrected=2.0 * fr, write_bytes=wr,
                      hbm_bytes_per_launch=2.0 * fr + wr, launches=max(fetch.get(k, (0, 0))[1], write.get(k, (0, 0))[1]))
    json.dump(out, sys.stdout, indent=1)


if __name__ == '__main__':
    main()
