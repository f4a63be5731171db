#!/usr/bin/env python3
"""bench.py -- atom-steps/s of one SevenNet-0 energy+force evaluation on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one energy+forces evaluation of the hot path (edge embedding -> 5
interaction blocks -> readout -> analytic reverse pass -> forces/virial) on a
synthetic periodic diamond-Si cell with seeded synthetic weights (no checkpoints
exist offline; throughput is weight-independent).  Workload at every N: the
~100k-atom cell BASELINE.json's metric is quoted on (97 336 atoms, Si x 23^3,
sigma = 0.05 A, cutoff 5.0 A); with N > 1 the SAME cell is split into N spatial
bricks (strong scaling) with an RCCL ghost-feature halo exchange per layer.
Inputs (graph + weights) are resident in HBM before the timed region.

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  roofline     -- dominant kernel class, achieved vs gfx950 peak, from HIP events
                  recorded around every launch of that class inside the timed steps
  cpu_baseline -- the CPU oracle (reference-equivalent PyTorch path) timed on this
                  box's host cores on a bounded sample (N=1 only)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E peak
MFMA_F32_PEAK_TF = 157.3   # dense fp32-input MFMA peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--reps', type=int, default=23, help='diamond cells per axis (23 -> 97 336 atoms)')
    ap.add_argument('--model', default='sevennet_0', choices=['sevennet_0', 'sevennet_l3i5', 'sevennet_mf_ompa'])
    ap.add_argument('--mlp-mode', default='bf16x6', choices=['bf16x6', 'fp32'],
                    help='radial-MLP MFMA mode: bf16 x6 split products (fp32-class accuracy) or exact fp32 MFMA')
    ap.add_argument('--host', default='python', choices=['python', 'native'],
                    help="who sequences the kernels: the Python host (per-kernel HIP-event timers) or the native "
                         "snet_model_eval sequencer (same kernels; kernel timers then come from an extra untimed pass)")
    ap.add_argument('--fused', default='auto', choices=['auto', 'off', 'fwd', 'bwd'],
                    help="radial-MLP last layer inside the tensor-product kernels (w / g_w never materialised): "
                         "'auto' = forward and reverse (default), 'off' = separate kernels")
    ap.add_argument('--terms', type=int, default=3, choices=[1, 2, 3],
                    help='bf16 terms per operand of the fused in-kernel products (3 = bf16x6, fp32-rounding class)')
    ap.add_argument('--no-overlap', action='store_true', help='radial MLPs on the main stream (no second stream)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-reps', type=int, default=5, help='CPU-baseline sample: cells per axis (5 -> 1000 atoms)')
    return ap.parse_args()


def model_config(name):
    from sevennet_amd.model_spec import sevennet_0_config, sevennet_l3i5_config, sevennet_mf_ompa_config
    return {'sevennet_0': sevennet_0_config, 'sevennet_l3i5': sevennet_l3i5_config,
            'sevennet_mf_ompa': sevennet_mf_ompa_config}[name]()


def species_of(cfg, n_atoms):
    """single species, or (119-species models) a seeded 4-species decoration Z in {3, 8, 14, 22}"""
    if int(cfg.get('_number_of_species', 1)) < 23:
        return np.zeros(n_atoms, np.int64)
    return np.random.default_rng(4).choice(np.array([3, 8, 14, 22]), size=n_atoms).astype(np.int64)


def kernel_model(ls, n_nodes, n_edges):
    """Algorithmic bytes / flops per launch of each kernel class of one layer
    (SURVEY.md §8(d) conventions: dst rows once per node, src rows once per edge,
    radial weights NOT counted for the tensor-product kernels, no cache credit)."""
    dx, dmid, wn = ls.conv.irreps_x.dim, ls.conv.irreps_out.dim, ls.conv.weight_numel
    nsh = ls.conv.irreps_sh.dim
    h = ls.mlp_dims
    mlp_flops = 2.0 * n_edges * sum(h[i] * h[i + 1] for i in range(len(h) - 1))
    return {
        f'conv_fwd[{ls.conv.tag}]': dict(bound='hbm', bytes=n_edges * (4 * dx + 4 * nsh + 8) + n_nodes * 4 * dmid),
        f'conv_fwd_fused[{ls.conv.tag}]': dict(bound='hbm', bytes=n_edges * (4 * dx + 4 * nsh + 8) + n_nodes * 4 * dmid),
        f'conv_bwd_edge[{ls.conv.tag}]': dict(bound='hbm', bytes=n_edges * (4 * dx + 2 * 4 * nsh + 8) + n_nodes * 4 * dmid),
        f'conv_bwd_node[{ls.conv.tag}]': dict(bound='hbm', bytes=n_edges * (4 * dmid + 4 * nsh + 12) + n_nodes * 4 * dx),
        f'radial_mlp_fwd[wn={wn}]': dict(bound='mfma', flops=mlp_flops),
        f'radial_mlp_bwd[wn={wn}]': dict(bound='mfma', flops=mlp_flops),
    }


def cpu_baseline(cfg, sd, reps):
    """The oracle = this repo's CPU restatement of the reference's e3nn/PyTorch path, fp32,
    on a bounded sample (about 10-20 s of CPU work)."""
    from oracle.model import OracleModel
    from sevennet_amd.neighbor import diamond_cubic, neighbor_list
    pos, cell = diamond_cubic(5.431, (reps,) * 3, 0.05, 2)
    ei, ev, _ = neighbor_list(pos, cell, [True] * 3, cfg['cutoff'])
    types = species_of(cfg, len(pos))
    m = OracleModel(cfg, sd, dtype=torch.float32, modal='mpa' if cfg.get('use_modality') else None)
    best = None
    default_threads = torch.get_num_threads()
    # eager PyTorch on many small ops does not scale to every core: try the default and 32 threads
    for threads in sorted({default_threads, min(32, default_threads)}):
        torch.set_num_threads(threads)
        t0 = time.perf_counter()
        m.forward(types, ei, ev)  # warm-up
        warm = time.perf_counter() - t0
        n_max = int(max(1, min(8, 6.0 / max(warm, 1e-3))))
        t0 = time.perf_counter()
        for _ in range(n_max):
            m.forward(types, ei, ev)
        dt = time.perf_counter() - t0
        rate = len(pos) * n_max / dt
        if best is None or rate > best[0]:
            best = (rate, threads, n_max, dt)
    torch.set_num_threads(default_threads)
    rate, threads, n, dt = best
    return dict(value=rate, unit='atom-steps/s', cores=threads, kind='port',
                sample=f'SevenNet-0 shape, {len(pos)}-atom Si cell ({ei.shape[1]} edges), {n} energy+force '
                       f'evaluations in {dt:.1f} s after one warm-up, fp32 torch CPU oracle (oracle/model.py), '
                       f'best of {{default, 32}} torch threads = {threads} on {os.cpu_count()} logical CPUs')


def main():
    a = parse()
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if a.gpus != world and world > 1:
        raise SystemExit(f'--gpus {a.gpus} but WORLD_SIZE={world}')
    if a.gpus > 1 and world == 1:
        raise SystemExit('launch N>1 with torch.distributed.run (one process per GPU)')
    # SNET_DIST_BACKEND=gloo lets a 1-GPU box exercise the N>1 code path (all ranks share cuda:0, transfers
    # staged by gloo): a functional dry run, never a measurement
    backend = os.environ.get('SNET_DIST_BACKEND', 'nccl')
    dev_id = local_rank % max(torch.cuda.device_count(), 1) if backend != 'nccl' else local_rank
    torch.cuda.set_device(dev_id)
    dev = f'cuda:{dev_id}'
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device(dev))
        else:
            dist.init_process_group(backend)

    from sevennet_amd.engine import HipForceEngine, build_graph
    from sevennet_amd.neighbor import diamond_cubic, neighbor_list
    from sevennet_amd.synthetic import random_state_dict

    cfg = model_config(a.model)
    sd = random_state_dict(cfg, seed=0)
    modal = 'mpa' if cfg.get('use_modality') else None
    eng = HipForceEngine(cfg, sd, device=dev, mlp_mode=a.mlp_mode, fused=(False if a.fused == 'off' else a.fused), fused_terms=a.terms, modal=modal, overlap=not a.no_overlap)

    pos, cell = diamond_cubic(5.431, (a.reps,) * 3, 0.05, 2)
    n_atoms = len(pos)
    t0 = time.perf_counter()
    halo = None
    if world == 1:
        ei, ev, _ = neighbor_list(pos, cell, [True] * 3, cfg['cutoff'])
        types = species_of(cfg, n_atoms)
        graph = build_graph(types, ei, ev, device=dev, num_species=eng.spec.num_species)
        n_edges_total = graph.n_edges
    else:
        from sevennet_amd.parallel import HaloExchange, build_brick_graph
        bg = build_brick_graph(pos, cell, species_of(cfg, n_atoms), cfg['cutoff'], world, rank)
        graph = build_graph(bg.types, bg.edge_index, bg.edge_vec, n_local=bg.n_local, device=dev,
                            num_species=eng.spec.num_species)
        halo = HaloExchange(bg.send_lists, bg.recv_counts, dev)
        ne = torch.tensor([graph.n_edges], device=dev, dtype=torch.int64)
        dist.all_reduce(ne)
        n_edges_total = int(ne.item())
    t_graph = time.perf_counter() - t0

    nat = None
    if a.host == 'native':
        if a.mlp_mode != 'bf16x6':
            raise SystemExit('--host native always uses the bf16x6 radial MLP')
        from sevennet_amd.native_model import NativeModel
        nat = NativeModel(cfg, sd, device=dev, modal=modal)
        nat.set_halo(halo)

    def step():
        return nat.compute(graph) if nat is not None else eng.compute(graph, halo=halo)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    eng.events = []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    t_enq = time.perf_counter() - t0  # host time to enqueue K steps (kernels run asynchronously)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if nat is not None:  # per-kernel timers live in the Python host: one extra untimed pass of the same kernels
        eng.events = []
        for _ in range(a.steps):
            eng.compute(graph, halo=halo)
        fence()
    times = eng.kernel_times_ms()
    eng.events = None

    # ---- per-kernel-class time over the timed region -> dominant kernel + roofline
    totals = {k: float(np.sum(v)) for k, v in times.items()}
    counts = {k: len(v) for k, v in times.items()}
    models = {}
    for ls in eng.spec.layers:
        models.update(kernel_model(ls, graph.n_local, graph.n_edges))
    # classes tagged '@side' ran on the second stream, overlapped with main-stream kernels: their event brackets
    # are not exclusive time, so the dominant class is chosen among the main-stream ones
    dominant = max((k for k in totals if k in models), key=lambda k: totals[k])
    avg_ms = totals[dominant] / counts[dominant]
    km = models[dominant]
    if km['bound'] == 'hbm':
        ach = km['bytes'] / (avg_ms * 1e-3) / 1e9
        roof = dict(bound='hbm', kernel=dominant, achieved=ach, peak=HBM_PEAK_GBS, unit='GB/s',
                    frac=ach / HBM_PEAK_GBS, traffic=None, avg_ms=avg_ms,
                    algorithmic_bytes_per_launch=km['bytes'])
    else:
        ach = km['flops'] / (avg_ms * 1e-3) / 1e12
        roof = dict(bound='mfma', kernel=dominant, achieved=ach, peak=MFMA_F32_PEAK_TF, unit='TFLOP/s',
                    frac=ach / MFMA_F32_PEAK_TF, traffic=None, avg_ms=avg_ms,
                    algorithmic_flops_per_launch=km['flops'])
    # HBM bytes per launch from the committed PMC passes of the same workload (profiles/README.md)
    try:
        import glob
        pmc = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_hbm_traffic.json')))[-1]
        with open(pmc) as f:
            tr = json.load(f)
        key = dominant.replace('conv_bwd_edge[', 'conv_bwd_edge_vec_').replace('conv_fwd_fused[', 'conv_ffwd_').replace('conv_fwd[', 'conv_fwd_') \
            .replace('conv_bwd_node[', 'conv_bwd_node_').rstrip(']')
        # a kernel class may be several kernels (one per x irrep block: <name>_k0, _k1, ...)
        parts = [v for k, v in tr.items() if k == key or k.startswith(key + '_k')]
        if world == 1 and a.reps == 23 and parts:
            roof['traffic'] = sum(v['hbm_bytes_per_launch'] for v in parts)
            roof['traffic_frac_of_peak'] = roof['traffic'] / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
            roof['traffic_source'] = os.path.basename(pmc) + ' (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, gfx950-corrected)'
    except Exception:  # noqa: BLE001
        pass
    if any(k.endswith('@side') for k in totals):
        roof['note'] = ("classes tagged '@side' (radial MLPs) run on a second HIP stream concurrently with the "
                        "main-stream kernels: their event brackets overlap the others and are not exclusive time; "
                        "avg_ms of the dominant kernel is measured with that concurrent work present")
    step_ms = dt / a.steps * 1e3
    e_total = out['energy'].clone()
    if world > 1:  # ranks hold partial energies of their bricks
        dist.all_reduce(e_total)
    roof['kernel_ms_per_step'] = {k: round(v / a.steps, 4) for k, v in sorted(totals.items(), key=lambda kv: -kv[1])}

    if rank == 0:
        res = {
            'metric': 'atom-steps/sec (energy+forces), SevenNet-0 100k-atom cell, 1/2/4/8 MI355X' if a.model == 'sevennet_0'
            else f'atom-steps/sec (energy+forces), {a.model} shape',
            'value': n_atoms * a.steps / dt, 'unit': 'atom-steps/s', 'n_gpus': world, 'steps': a.steps,
            'warmup': a.warmup, 'ms_per_step': step_ms, 'higher_is_better': True, 'scaling': 'strong',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'{a.model} shape (5 interaction layers), {n_atoms}-atom periodic diamond-Si '
                                   f'cell (a=5.431 A x {a.reps}^3, sigma=0.05 A), cutoff {cfg["cutoff"]} A, '
                                   f'{n_edges_total} directed edges, seeded synthetic weights',
                       'atoms': n_atoms, 'edges': n_edges_total,
                       'parallelism': 'single GPU' if world == 1 else f'spatial decomposition x{world}, RCCL halo',
                       'graph_build_s': round(t_graph, 3),
                       'host': a.host, 'host_enqueue_ms_per_step': round(t_enq / a.steps * 1e3, 3),
                       'energy': float(e_total.cpu())},
            'roofline': roof,
        }
        if world == 1 and not a.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline(cfg, sd, a.cpu_reps)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
