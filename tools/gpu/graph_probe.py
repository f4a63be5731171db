#!/usr/bin/env python3
"""Does capturing one evaluation in a HIP graph pay?  eager compute() vs torch.cuda.CUDAGraph replay of the same call
(kernel launches through ctypes on the capturing stream, buffers from the graph's private pool), at three system sizes."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    from bench import model_config
    from sevennet_amd.engine import HipForceEngine, build_graph
    from sevennet_amd.neighbor import diamond_cubic, neighbor_list
    from sevennet_amd.synthetic import random_state_dict
    cfg = model_config('sevennet_0')
    eng = HipForceEngine(cfg, random_state_dict(cfg, 0))
    for reps in (4, 6, 12, 23):
        pos, cell = diamond_cubic(5.431, (reps,) * 3, 0.05, 2)
        ei, ev, _ = neighbor_list(pos, cell, [True] * 3, cfg['cutoff'])
        g = build_graph(np.zeros(len(pos), np.int64), ei, ev)
        for _ in range(3):
            ref = eng.compute(g)
        torch.cuda.synchronize()
        n = 20

        def timed(fn):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            t_host = time.perf_counter() - t0
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3, t_host / n * 1e3
        eager, eager_host = timed(lambda: eng.compute(g))
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            eng.compute(g)
        torch.cuda.current_stream().wait_stream(s)
        cg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(cg):
            out = eng.compute(g)
        cg.replay()
        torch.cuda.synchronize()
        same = float((out['forces'] - ref['forces']).abs().max()), float(out['energy'] - ref['energy'])
        graph, graph_host = timed(cg.replay)
        print(f'{len(pos):6d} atoms {g.n_edges:8d} edges: eager {eager:7.3f} ms/step (host enqueue {eager_host:6.3f}), '
              f'graph replay {graph:7.3f} ms/step (host {graph_host:6.3f}); replay vs eager: max|dF| {same[0]:.1e}, dE {same[1]:.1e}',
              flush=True)


if __name__ == '__main__':
    main()
