#!/usr/bin/env python3
"""Wall time of one SevenNetCalculator evaluation at the benchmark size, host positions in -> host
energy / forces / stress out (GPU neighbor list + graph + pair map + model + D2H).

    python tools/calculator_cost.py [--reps 23]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=23)
    a = ap.parse_args()
    from sevennet_amd.calculator import SevenNetCalculator
    from sevennet_amd.model_spec import sevennet_0_config
    from sevennet_amd.neighbor import diamond_cubic
    from sevennet_amd.synthetic import random_state_dict
    cfg = sevennet_0_config()
    cfg['_type_map'] = {14: 0}
    calc = SevenNetCalculator((cfg, random_state_dict(cfg, 0)), file_type='model_instance', device='cuda:0')
    pos, cell = diamond_cubic(5.431, (a.reps,) * 3, 0.05, 2)
    numbers = np.full(len(pos), 14)
    calc.compute(numbers, pos, cell, [True] * 3)
    ts = []
    for k in range(5):
        p = pos + 1e-3 * k            # new positions every call, as in MD
        t0 = time.perf_counter()
        res = calc.compute(numbers, p, cell, [True] * 3)
        ts.append(time.perf_counter() - t0)
    print(f'{len(pos)} atoms, {res["num_edges"]} edges, energy {res["energy"]:.4f}')
    print(f'calculator.compute per call (host positions -> host E/F/stress): median {np.median(ts) * 1e3:.1f} ms, '
          f'min {min(ts) * 1e3:.1f} ms')


if __name__ == '__main__':
    main()
