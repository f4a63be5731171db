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
Weights and the graph TOPOLOGY (CSR index arrays, pair map) are resident in HBM before the timed
region; the per-step inputs of an MD step -- the edge vectors r_j - r_i [E,3] fp32, i.e. what positions
turn into -- are copied host -> device INSIDE every timed step ("graph already built;
H2D of positions/edges included"; `--no-h2d` times the resident-input step instead).

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
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16 MFMA peak (the split-precision products run there)
# fp32 vector peak: 256 CUs x 4 SIMD-32 x (64 lanes / 2 cycles) x 2 flop x 2.4 GHz (MI355X_MICROARCH.md: a wave64 VALU instruction
# issues over 2 cycles).  Round 4 priced the vector pipe at 4 cycles per instruction (78.6 TF): corrected.
# What a kernel can reach depends on its occupancy: ONE wave issues a vector instruction every ~7.5 cycles whatever its kind, so the
# pipe needs four waves per SIMD for its rate; measured with s_memtime inside the kernel (tools/gpu/issue_probe,
# profiles/r05_issue_rate_probe.txt): 2.5 cycles per SIMD-instruction at 4 waves per SIMD (with the clock down at 1.6 GHz: 103 TF),
# 4.0 at the two waves per SIMD the fused reverse kernels run at, 7.5 at one.
VALU_F32_PEAK_TF = 157.3
VALU_F32_MEASURED_TF = {4: 103.0, 2: 66.0, 1: 39.0}   # waves per SIMD -> v_fma_f32 TFLOP/s of the whole chip (issue probe, round 5)
# algorithmic work per atom-step of the named shapes (bytes, flops), fwd + reverse
STEP_WORK = {'sevennet_0': (0.992e6, 44.5e6), 'sevennet_l3i5': (1.584e6, 119e6), 'sevennet_mf_ompa': (3.225e6, 305e6)}


# what the path computes in: fp32 storage, accumulation, tensor product, gate, force reduction everywhere; the dense
# contractions run on the matrix cores with every fp32 operand split into low-precision terms (fp32 accumulate)
DTYPE_LABEL = {
    0: 'f32 (storage / accumulate / tensor product; dense products as bf16x6 splits: fp32-rounding class)',
    4: 'f32 (storage / accumulate / tensor product; in-kernel W2 products f16x3 = two fp16 terms per operand, '
       'three matrix-core products: fp32-rounding class; node linears bf16x6)',
    3: 'f32 (storage / accumulate / tensor product; in-kernel W2 products bf16x6: fp32-rounding class)',
    2: 'f32 storage / accumulate; in-kernel W2 products bf16x3 (NOT fp32 class: up to 3.5e-4 eV/A at max|F| = 8 eV/A)',
    1: 'f32 storage / accumulate; in-kernel W2 products plain bf16 (1e-2 relative)',
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--reps', type=int, default=0, help='diamond cells per axis; default per model: sevennet_0 23 (97 336 atoms), '
                    'sevennet_l3i5 19 (54 872, BASELINE config 4), sevennet_mf_ompa 15 (27 000)')
    ap.add_argument('--model', default='sevennet_0', choices=['sevennet_0', 'sevennet_l3i5', 'sevennet_mf_ompa'])
    ap.add_argument('--mlp-mode', default='bf16x6', choices=['bf16x6', 'fp32'],
                    help='radial-MLP MFMA mode: bf16 x6 split products (fp32-class accuracy) or exact fp32 MFMA')
    ap.add_argument('--host', default='python', choices=['python', 'native'],
                    help="who sequences the kernels: the Python host (per-kernel HIP-event timers) or the native "
                         "snet_model_eval sequencer (same kernels; kernel timers then come from an extra untimed pass)")
    ap.add_argument('--fused', default='auto', choices=['auto', 'off', 'fwd', 'bwd'],
                    help="radial-MLP last layer inside the tensor-product kernels (w / g_w never materialised): "
                         "'auto' = forward and reverse (default), 'off' = separate kernels")
    ap.add_argument('--terms', type=int, default=4, choices=[1, 2, 3, 4],
                    help='precision mode of the fused in-kernel products: 4 = f16x3 (two fp16 terms per operand, three products; '
                         'fp32-rounding class; engine default), 3 = bf16x6 (fp32-rounding class, six products), 2 = bf16x3 '
                         '(outside the 1e-4 eV/A bar at MD-scale forces), 1 = plain bf16')
    ap.add_argument('--no-overlap', action='store_true', help='radial MLPs on the main stream (no second stream)')
    ap.add_argument('--no-transposed', action='store_true', help='last layer: per-edge g_xe rows + segment sum instead of the '
                    'transposed scalar convolution (A/B switch, Python host)')
    ap.add_argument('--halo', default='auto', choices=['auto', 'native', 'torch'],
                    help="N > 1: ghost exchange by libsnet_hip.so's own RCCL send/recv groups ('native', default with the "
                         "nccl backend) or by torch.distributed.all_to_all_single ('torch'; the only choice over gloo)")
    ap.add_argument('--no-halo-overlap', action='store_true', help='N > 1, native halo: run the forward ghost exchange on the '
                    'compute stream instead of a second stream beside the self-connection / hidden radial layers')
    ap.add_argument('--dist-path', action='store_true', help='take the N > 1 code path (process group, RCCL unique-id broadcast, '
                    'brick graph, native halo with its communicators) even at world size 1: what tools/gpu/rccl_world1_soak.sh runs')
    ap.add_argument('--h2d', default='positions', choices=['positions', 'edges', 'none'],
                    help="per-step input copied host -> device inside every timed step: 'positions' (default: fp64 positions "
                         "[n,3], edge vectors formed on the GPU from the resident topology + image offsets -- what an MD host "
                         "hands over), 'edges' (the fp32 edge vectors [E,3], round-2 behaviour), 'none' (inputs resident)")
    ap.add_argument('--no-h2d', action='store_true', help="same as --h2d none")
    ap.add_argument('--brick-proxy', type=int, default=0, metavar='W',
                    help='N = 1 only: after the timed region also time rank 0\'s brick of a W-way spatial decomposition of the same cell '
                         'on this GPU (ghost rows, interior / boundary split, a self-loop halo that packs, copies and accumulates the '
                         'same rows the RCCL exchange would, on the halo stream) and report it as `brick_proxy` beside the ideal N = 1 step / W')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-reps', type=int, default=11, help='CPU-baseline sample: cells per axis (11 -> the 10 648-atom cell of BASELINE '
                    'config 2, SURVEY.md 8(d), ~4 min; 6 -> only the 1728-atom cell that picks the thread count, ~1 min)')
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


def kernel_model(ls, n_nodes, n_edges, mlp_tail=False, nb=8):
    """Algorithmic bytes / flops per launch of each kernel class of one layer.
    `bytes` is SURVEY.md section 8(d)'s count, literally: per edge fwd = 4 dx + 12 + 8, reverse = 2 * 4 dx + 12 + 8 + 12,
    plus the destination rows once per node (4 dmid); radial weights NOT counted, no cache credit.  `bytes_incl_staging`
    adds what the fused kernels really move besides (spherical harmonics and their Jacobian, h2, emb / g_emb rows)."""
    dx, dmid, wn = ls.conv.irreps_x.dim, ls.conv.irreps_out.dim, ls.conv.weight_numel
    nsh = ls.conv.irreps_sh.dim
    h = ls.mlp_dims
    mlp_flops = 2.0 * n_edges * sum(h[i] * h[i + 1] for i in range(len(h) - 1))
    # reverse kernel with the MLP's hidden layers reversed inside: h2 in, emb in, g_emb read + written (no g_h2)
    bwd_mlp_bytes = (256 + 3 * 4 * nb) if mlp_tail else 512
    bwd_tail_flops = 2.0 * n_edges * (2 * nb * 64 + 2 * 64 * 64) if mlp_tail else 0.0
    return {
        f'conv_fwd[{ls.conv.tag}]': dict(bound='hbm', bytes=n_edges * (4 * dx + 4 * nsh + 8) + n_nodes * 4 * dmid),
        # fused kernels: w = h2 @ W2 and g_h2 = g_w @ W2^T run inside (bf16 x terms products on the matrix cores);
        # per edge they move the source row, Y (+ its Jacobian), src / w_row, h2[64] (+ g_xe, g_h2, g_vec on the way back)
        f'conv_fwd_fused[{ls.conv.tag}]': dict(bound='hbm', bytes=n_edges * (4 * dx + 12 + 8) + n_nodes * 4 * dmid,
                                               bytes_incl_staging=n_edges * (4 * dx + 4 * nsh + 8 + 256) + n_nodes * 4 * dmid,
                                               flops=2.0 * n_edges * 64 * wn),
        f'conv_bwd_fused[{ls.conv.tag}]': dict(bound='hbm', bytes=n_edges * (2 * 4 * dx + 12 + 8 + 12) + n_nodes * 4 * dmid,
                                               bytes_incl_staging=n_edges * (2 * 4 * dx + 4 * 4 * nsh + 8 + bwd_mlp_bytes + 24) + n_nodes * 4 * dmid,
                                               flops=2.0 * 2.0 * n_edges * 64 * wn + bwd_tail_flops),
        f'conv_bwd_edge[{ls.conv.tag}]': dict(bound='hbm', bytes=n_edges * (4 * dx + 2 * 4 * nsh + 8) + n_nodes * 4 * dmid),
        f'conv_bwd_node[{ls.conv.tag}]': dict(bound='hbm', bytes=n_edges * (4 * dmid + 4 * nsh + 12) + n_nodes * 4 * dx),
        f'radial_mlp_fwd[wn={wn}]': dict(bound='mfma', flops=mlp_flops),
        f'radial_mlp_bwd[wn={wn}]': dict(bound='mfma', flops=mlp_flops),
    }


def scheduled_work(eng, graph):
    """Arithmetic one evaluation of THIS rank's graph schedules, from the engine's own tables: matrix-core flops as issued (each
    low-precision product of a split counted) and the vector-pipe flops of the sparse tensor products (forward + reverse)."""
    from sevennet_amd.codegen import _path_terms
    E, N, NT = graph.n_edges, graph.n_local, graph.n_total
    P = graph.n_pairs if graph.w_row is not None else E
    terms = {1: 1, 2: 3, 3: 6, 4: 3}[eng.fused_terms]
    mfma = valu = 0.0
    for t, L in enumerate(eng.layers):
        ls = L.spec
        wn, d = ls.conv.weight_numel, ls.mlp_dims
        mfma += 2.0 * P * (d[0] * d[1] + d[1] * d[2]) * 6                      # hidden radial layers, forward (bf16x6)
        mfma += 2.0 * E * 64 * wn * terms                                      # w = h2 W2 inside the forward kernel
        mfma += 2.0 * 2.0 * E * 64 * wn * terms                                # reverse: w again and g_h2 += g_w W2^T
        mfma += 2.0 * E * (2 * d[0] * d[1] + 2 * d[1] * d[2]) * terms          # hidden-layer tail of the reverse kernel
        if getattr(L, 'tplan', None) is not None:
            mfma += 2.0 * E * 64 * wn * terms                                  # transposed scalar convolution (last layer)
        for lin in (L.sc, L.si1, L.si2):
            if lin is None or (t == 0 and lin is not L.si2 and eng.h0_table is not None):
                continue                                                       # (layer 0: species tables, no GEMM)
            rows = N
            f = sum((2 * b.l + 1) * b.mul_in * b.mul_out for b in lin.spec.blocks if b.species <= 0)
            mfma += 2.0 * rows * f * 6 * (1 if t == 0 else 2)                  # forward + transposed (layer 0: forward only)
        for p in ls.conv.paths:
            tr = _path_terms(p)
            nnz, pab, pac = len(tr), len({(a_, b_) for a_, b_, _, _ in tr}), len({(a_, c_) for a_, _, c_, _ in tr})
            d1, d3 = 2 * p.l1 + 1, 2 * p.l3 + 1
            valu += E * p.mul * (2.0 * nnz + pab + 2.0 * d3)                   # forward body per (edge, channel)
            valu += E * (2.0 * nnz + p.mul * (4.0 * pac + 5.0 * d1))           # reverse body: V per edge, P / s per (a, c), 3 per a
            if getattr(L, 'tplan', None) is not None:
                valu += E * p.mul * (2.0 * nnz + pab + 2.0 * d1)               # transposed convolution of the same paths
    return dict(mfma=mfma, valu=valu)


def fused_source_hashes():
    """sha1 (12 hex digits) of the generated source of every ahead-of-time fused tensor-product kernel: what a PMC profile is stamped
    with (tools/collect_profiles.sh) and what `roofline.traffic` is checked against"""
    import hashlib
    from sevennet_amd import codegen_fused
    from sevennet_amd.shapes import aot_conv_specs
    out = {}
    for tag, spec in aot_conv_specs(()).items():
        if codegen_fused.fusable(spec):
            out[tag] = hashlib.sha1(codegen_fused.gen_conv_fused(spec).encode()).hexdigest()[:12]
    return out


def cpu_model_name():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown CPU'


def physical_cores():
    """distinct (package, core) pairs of /proc/cpuinfo (SMT siblings counted once); os.cpu_count() if unreadable"""
    try:
        seen, pkg = set(), None
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('physical id'):
                    pkg = line.split(':')[1].strip()
                elif line.startswith('core id'):
                    seen.add((pkg, line.split(':')[1].strip()))
        if seen:
            return len(seen)
    except OSError:
        pass
    return os.cpu_count() or 1


def cpu_baseline(cfg, sd, reps, small_reps=6):
    """The oracle = this repo's CPU restatement of the reference's e3nn/PyTorch path, fp32, on a bounded sample.
    Two legs:
      * the small cell (`small_reps`^3 cells = 1728 atoms) at BOTH thread counts SURVEY.md 8(d) names -- every physical core of the
        box, and the 32 threads that were the best setting found for this eager many-small-ops workload -- three evaluations each
        after a warm-up: picks the thread count and keeps the rounds 1-5 figure comparable (`small_sample`);
      * the cell SURVEY.md 8(d) asks the CPU baseline on -- `reps` = 11: the 10 648-atom cell of BASELINE config 2 -- at the faster
        thread count, three evaluations after a warm-up (~4 min on the GPU box's EPYC): this is `value`.
    `--cpu-reps 6` skips the second leg (value = the small cell's figure)."""
    from oracle.model import OracleModel
    from sevennet_amd.neighbor import diamond_cubic, neighbor_list
    m = OracleModel(cfg, sd, dtype=torch.float32, modal='mpa' if cfg.get('use_modality') else None)
    default_threads = torch.get_num_threads()
    phys = physical_cores()
    n_eval = 3

    def cell(r):
        pos, cell_ = diamond_cubic(5.431, (r,) * 3, 0.05, 2)
        ei, ev, _ = neighbor_list(pos, cell_, [True] * 3, cfg['cutoff'])
        return pos, ei, ev, species_of(cfg, len(pos))

    def timed(threads, types, ei, ev):
        torch.set_num_threads(threads)
        m.forward(types, ei, ev)  # warm-up
        ts = []
        for _ in range(n_eval):
            t0 = time.perf_counter()
            m.forward(types, ei, ev)
            ts.append(time.perf_counter() - t0)
        return ts

    pos_s, ei_s, ev_s, types_s = cell(min(small_reps, reps))
    runs = {}
    for threads in sorted({min(32, phys), phys}):
        ts = timed(threads, types_s, ei_s, ev_s)
        runs[threads] = (len(pos_s) * n_eval / sum(ts), ts)
    best = max(runs, key=lambda k: runs[k][0])
    small = dict(atoms=len(pos_s), edges=int(ei_s.shape[1]), value=runs[best][0], cores=best,
                 by_threads={str(k): round(v[0], 1) for k, v in runs.items()},
                 evaluation_s={str(k): [round(t, 2) for t in v[1]] for k, v in runs.items()})
    if reps > small_reps:
        pos, ei, ev, types = cell(reps)
        ts = timed(best, types, ei, ev)
        rate, n_at, n_ed = len(pos) * n_eval / sum(ts), len(pos), int(ei.shape[1])
    else:
        ts, rate, n_at, n_ed = runs[best][1], runs[best][0], len(pos_s), int(ei_s.shape[1])
    torch.set_num_threads(default_threads)
    return dict(value=rate, unit='atom-steps/s', cores=best, kind='port', cpu=cpu_model_name(),
                logical_cpus=os.cpu_count(), physical_cores=phys, evaluations=n_eval, evaluation_s=[round(t, 2) for t in ts],
                by_threads=small['by_threads'], small_sample=small,
                sample=f'SevenNet-0 shape, {n_at}-atom Si cell ({n_ed} edges), {n_eval} energy+force '
                       f'evaluations in {sum(ts):.1f} s after one warm-up, fp32 torch CPU oracle (oracle/model.py), '
                       f'{best} torch threads on {cpu_model_name()} ({phys} physical cores, {os.cpu_count()} logical CPUs); '
                       f'thread count chosen on the {len(pos_s)}-atom cell, atom-steps/s by thread count there: '
                       + ', '.join(f'{k}: {v[0]:.0f}' for k, v in runs.items()))


class SelfLoopHalo:
    """Halo of ONE brick measured alone (bench.py --brick-proxy): every exchange moves the rows the real one would -- the send rows
    are packed (index_select over the brick's send lists), copied device to device into the ghost rows / added into the owners'
    rows on the way back -- on a second stream with the same start / finish protocol as NativeHalo, so the interior / boundary
    split of the hosts runs; only the wire (RCCL over xGMI) and the peers are missing.  The ghost rows receive copies of OWNED rows
    (finite features of the right magnitude): the numbers of the step are not those of the decomposed cell, its cost is."""

    def __init__(self, send_lists, n_ghost, device):
        idx = np.concatenate([np.asarray(s_, np.int64) for s_ in send_lists]) if len(send_lists) else np.zeros(0, np.int64)
        if len(idx) == 0:
            idx = np.zeros(1, np.int64)
        self.n_ghost = int(n_ghost)
        self.n_send = int(sum(len(s_) for s_ in send_lists))
        take = np.resize(idx, self.n_ghost)                    # as many rows as arrive from the peers
        self.idx = torch.as_tensor(take, dtype=torch.long, device=device)
        self.side = torch.cuda.Stream(device=device)
        self.overlap = True

    def forward_start(self, x, n_local):
        cur = torch.cuda.current_stream(x.device)
        self.side.wait_stream(cur)
        with torch.cuda.stream(self.side):
            if self.n_ghost:
                x[n_local:n_local + self.n_ghost] = x.index_select(0, self.idx)     # pack + "wire" + unpack
        done = torch.cuda.Event()
        done.record(self.side)
        return done, x

    def forward_finish(self, handle):
        torch.cuda.current_stream(handle[1].device).wait_event(handle[0])

    def forward(self, x, n_local):
        self.forward_finish(self.forward_start(x, n_local))

    def reverse_start(self, gx, n_local):
        cur = torch.cuda.current_stream(gx.device)
        self.side.wait_stream(cur)
        with torch.cuda.stream(self.side):
            staged = gx[n_local:n_local + self.n_ghost].clone()                     # ghost rows leave; what the peers return is staged
        done = torch.cuda.Event()
        done.record(self.side)
        return done, gx, staged, n_local

    def reverse_finish(self, handle, gx=None):
        done, gx0, staged, n_local = handle
        gx = gx0 if gx is None else gx
        torch.cuda.current_stream(gx.device).wait_event(done)
        if self.n_ghost:
            gx[:n_local].index_add_(0, self.idx, staged)                            # deterministic order is not the point here

    def reverse(self, gx, n_local):
        self.reverse_finish(self.reverse_start(gx, n_local), gx)


def brick_proxy(a, cfg, sd, eng, pos, cell, dev, modal, n1_step_ms):
    """rank 0's brick of an a.brick_proxy-way decomposition of the benchmark cell, alone on this GPU: ms per step of both hosts,
    dispatches, kernel time, gaps, against the ideal (the measured N = 1 step / W)."""
    from sevennet_amd.engine import build_graph
    from sevennet_amd.parallel import build_brick_graph
    W = a.brick_proxy
    bg = build_brick_graph(pos, cell, species_of(cfg, len(pos)), cfg['cutoff'], W, 0)
    g = build_graph(bg.types, bg.edge_index, bg.edge_vec, n_local=bg.n_local, n_interior=bg.n_interior, device=dev,
                    num_species=eng.spec.num_species)
    n_ghost = g.n_total - g.n_local
    halo = SelfLoopHalo(bg.send_lists, n_ghost, dev)

    def timed(fn, steps):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        t_enq = time.perf_counter() - t0
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3, t_enq / steps * 1e3
    steps = max(a.steps, 20)
    eng.events, eng.event_filter = None, None
    ms_py, enq_py = timed(lambda: eng.compute(g, halo=halo), steps)
    eng.events = []
    eng.compute(g, halo=halo)
    torch.cuda.synchronize()
    times = eng.kernel_times_ms()
    eng.events = None
    kern = {k: float(np.sum(v)) for k, v in times.items() if not k.startswith('halo') and not k.endswith('@side')}
    n_disp = int(sum(len(v) for k, v in times.items() if not k.startswith('halo')))
    out = dict(world=W, rank=0, atoms_local=int(g.n_local), interior_atoms=int(bg.n_interior or 0), ghost_rows=int(n_ghost),
               edges=int(g.n_edges), send_rows=halo.n_send, ms_per_step=round(ms_py, 3), host_enqueue_ms_per_step=round(enq_py, 3),
               ideal_ms=round(n1_step_ms / W, 3), ratio_to_ideal=round(ms_py / (n1_step_ms / W), 3),
               dispatches_per_step=n_disp, kernel_ms_per_step=round(sum(kern.values()), 3),
               gap_ms_per_step=round(ms_py - sum(kern.values()), 3),
               kernel_ms_top={k: round(v, 3) for k, v in sorted(kern.items(), key=lambda kv: -kv[1])[:8]},
               halo='self-loop: pack + device-to-device copy + accumulate of the rows the RCCL exchange moves '
                    f'({2 * 4} feature exchanges + 1 force fold per step), second stream, interior / boundary split on; no wire, no peers',
               exchange_mb=round(n_ghost * 480 * 4 / 1e6, 2))
    try:   # the native sequencer on the same brick (its own second stream; the halo through callbacks)
        from sevennet_amd.native_model import NativeModel
        nat = NativeModel(cfg, sd, device=dev, modal=modal)
        nat.set_halo(halo)
        ms_nat, enq_nat = timed(lambda: nat.compute(g), steps)
        out['ms_per_step_native_host'] = round(ms_nat, 3)
        out['host_enqueue_ms_per_step_native_host'] = round(enq_nat, 3)
    except Exception as exc:  # noqa: BLE001
        out['native_host_error'] = repr(exc)[:200]
    return out


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
    dist_mode = world > 1 or a.dist_path
    if dist_mode:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29531')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
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
    eng = HipForceEngine(cfg, sd, device=dev, mlp_mode=a.mlp_mode, fused=(False if a.fused == 'off' else a.fused), fused_terms=a.terms, modal=modal, overlap=not a.no_overlap,
                         transposed_conv=not a.no_transposed)

    # workloads of SURVEY.md section 8(d): config 3 (SevenNet-0: sigma 0.05 A, seed 2), config 4 (l3i5: "amorphous",
    # sigma 0.35 A with a 1.8 A minimum-distance reject, seed 3), config 5's per-GPU share (MF-ompa: 4-species
    # decoration, sigma 0.1 A, seed 4, cutoff 6 A)
    a.reps = a.reps or {'sevennet_0': 23, 'sevennet_l3i5': 19, 'sevennet_mf_ompa': 15}[a.model]
    if a.model == 'sevennet_l3i5':
        from sevennet_amd.neighbor import amorphous_cell
        pos, cell = amorphous_cell(5.431, (a.reps,) * 3, 0.35, 3, 1.8)
        wl_note = 'sigma=0.35 A with a 1.8 A minimum-distance reject ("amorphous", BASELINE config 4)'
    elif a.model == 'sevennet_mf_ompa':
        pos, cell = diamond_cubic(5.431, (a.reps,) * 3, 0.1, 4)
        wl_note = 'sigma=0.1 A, 4-species decoration Z in {3, 8, 14, 22} (per-GPU share of BASELINE config 5)'
    else:
        pos, cell = diamond_cubic(5.431, (a.reps,) * 3, 0.05, 2)
        wl_note = 'sigma=0.05 A'
    n_atoms = len(pos)
    t0 = time.perf_counter()
    halo = None
    if not dist_mode:
        ei, ev, _ = neighbor_list(pos, cell, [True] * 3, cfg['cutoff'])
        types = species_of(cfg, n_atoms)
        graph = build_graph(types, ei, ev, device=dev, num_species=eng.spec.num_species)
        n_edges_total = graph.n_edges
    else:
        from sevennet_amd.parallel import HaloExchange, build_brick_graph
        bg = build_brick_graph(pos, cell, species_of(cfg, n_atoms), cfg['cutoff'], world, rank)
        graph = build_graph(bg.types, bg.edge_index, bg.edge_vec, n_local=bg.n_local, n_interior=bg.n_interior, device=dev,
                            num_species=eng.spec.num_species)
        use_native = a.halo == 'native' or (a.halo == 'auto' and backend == 'nccl')
        rccl_info = None
        if use_native:
            from sevennet_amd.parallel import NativeHalo, RcclComm
            # Agree on the transport BEFORE the collective communicator construction: RcclComm() broadcasts the unique
            # id and runs ncclCommInitRank -- a rank that failed alone in there would leave the others blocked inside it.  The
            # local probe (library loads, every symbol binds) is what can differ between ranks; it is reduced first.
            ok = 1 if RcclComm.available() else 0
            if world > 1:
                flag = torch.tensor([ok], device=dev, dtype=torch.int32)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                ok = int(flag.item())
            if not ok:
                if a.halo == 'native':
                    raise SystemExit('--halo native: librccl.so cannot be bound by libsnet_hip.so on every rank')
                if rank == 0:
                    print('[bench] native RCCL halo unavailable on some rank; every rank uses the torch.distributed exchange', file=sys.stderr)
                use_native = False
            else:
                rccl_comm = RcclComm(world, rank)   # collective; a failure past the probe is fatal on every rank alike
                rccl_info = rccl_comm.info()
                if rccl_info != (world, rank):
                    raise SystemExit(f'[bench rank {rank}] RCCL reports communicator {rccl_info}, expected ({world}, {rank})')
                halo = NativeHalo(rccl_comm, bg.send_lists, bg.recv_counts, overlap=not a.no_halo_overlap)
        if not use_native:
            halo = HaloExchange(bg.send_lists, bg.recv_counts, dev)
        ne = torch.tensor([graph.n_edges], device=dev, dtype=torch.int64)
        dist.all_reduce(ne)
        n_edges_total = int(ne.item())
        # start-up self-check of the decomposition, so that a first real multi-GPU run is diagnosable from its
        # JSON line alone: owned atoms partition the cell; what every rank expects to RECEIVE from a peer is what that peer will
        # SEND to it (the two count matrices are transposes); ghost rows = rows received
        sc = torch.zeros(world, world, device=dev, dtype=torch.int64)
        rc_ = torch.zeros(world, world, device=dev, dtype=torch.int64)
        sc[rank] = torch.tensor([len(s_) for s_ in bg.send_lists], device=dev, dtype=torch.int64)
        rc_[rank] = torch.tensor([int(c) for c in bg.recv_counts], device=dev, dtype=torch.int64)
        own = torch.tensor([graph.n_local, graph.n_total - graph.n_local, getattr(bg, 'n_interior', 0) or 0], device=dev, dtype=torch.int64)
        owns = torch.zeros(world, 3, device=dev, dtype=torch.int64)
        owns[rank] = own
        for t_ in (sc, rc_, owns):
            dist.all_reduce(t_)
        sc, rc_, owns = sc.cpu().numpy(), rc_.cpu().numpy(), owns.cpu().numpy()
        problems = []
        if int(owns[:, 0].sum()) != n_atoms:
            problems.append(f'owned atoms sum to {int(owns[:, 0].sum())}, cell has {n_atoms}')
        if not (sc.T == rc_).all():
            problems.append('send / receive count matrices are not transposes of each other')
        if not (rc_.sum(1) == owns[:, 1]).all():
            problems.append('ghost rows differ from the rows received')
        if np.diag(sc).any():
            problems.append('a rank sends to itself')
        if problems:
            raise SystemExit(f'[bench rank {rank}] decomposition self-check failed: ' + '; '.join(problems))
        selfcheck = dict(ok=True, owned_atoms_by_rank=owns[:, 0].tolist(), ghost_rows_by_rank=owns[:, 1].tolist(),
                         interior_atoms_by_rank=owns[:, 2].tolist(), send_rows_by_rank=sc.sum(1).tolist(),
                         peers_by_rank=(sc > 0).sum(1).tolist(),
                         rccl_comm=(None if rccl_info is None else dict(nranks=rccl_info[0], rank0_user_rank=rccl_info[1])))
    t_graph = time.perf_counter() - t0

    nat = None
    if a.host == 'native':
        if a.mlp_mode != 'bf16x6':
            raise SystemExit('--host native always uses the bf16x6 radial MLP')
        from sevennet_amd.native_model import NativeModel
        nat = NativeModel(cfg, sd, device=dev, modal=modal)
        nat.set_halo(halo)

    # per-step input: the edge vectors (what an MD host derives from the new positions) come from pinned host
    # memory every step, on the compute stream, inside the timed region
    ev_host = pos_host = None
    if a.no_h2d:
        a.h2d = 'none'
    if a.h2d == 'edges':
        ev_host = graph.edge_vec.cpu().pin_memory()
    elif a.h2d == 'positions':
        # resident: topology (center, src) and the periodic-image offset of every edge; per step: the positions of this
        # rank's local + ghost atoms (fp64, as ASE / LAMMPS hold them)
        import ctypes as C
        from sevennet_amd import _lib
        pos_rank = np.ascontiguousarray(pos if not dist_mode else pos[bg.global_ids], np.float64)
        pos_host = torch.from_numpy(pos_rank).pin_memory()
        pos_dev = pos_host.to(dev)
        ev64 = torch.from_numpy(np.ascontiguousarray(ev if not dist_mode else bg.edge_vec, np.float64)).to(dev)
        if graph.order is not None:
            ev64 = ev64[graph.order]
        shift_dev = (ev64 - (pos_dev[graph.src.long()] - pos_dev[graph.center.long()])).contiguous()
        del ev64
        lib_ = _lib.load()

    def step():
        if ev_host is not None:
            graph.edge_vec.copy_(ev_host, non_blocking=True)
        elif pos_host is not None:
            pos_dev.copy_(pos_host, non_blocking=True)
            _lib.check(lib_.snet_edge_vectors(C.c_void_p(pos_dev.data_ptr()), C.c_void_p(graph.center.data_ptr()),
                                              C.c_void_p(graph.src.data_ptr()), C.c_void_p(shift_dev.data_ptr()), graph.n_edges,
                                              C.c_void_p(graph.edge_vec.data_ptr()),
                                              C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'snet_edge_vectors')
        return nat.compute(graph) if nat is not None else eng.compute(graph, halo=halo)

    def fence():
        torch.cuda.synchronize()
        if dist_mode:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    # Per-kernel HIP events cost time on this platform (a timestamp packet per record: the step with ~120 records
    # ran 15 % slower than without).  So: (1) one untimed pass with every class bracketed picks the dominant class,
    # (2) the TIMED steps bracket only that class (6 records per step) -- its average launch duration is measured
    # live inside the timed region, as the contract asks --, (3) a last untimed pass gives the full breakdown.
    eng.events, eng.event_filter = [], None
    eng.compute(graph, halo=halo)
    fence()
    probe = {}
    for name, t in eng.kernel_times_ms().items():
        probe[name] = float(np.sum(t))
    models0 = {}
    for ls, L in zip(eng.spec.layers, eng.layers):
        models0.update(kernel_model(ls, graph.n_local, graph.n_edges, getattr(L, 'mlp_tail', False), eng.spec.n_basis))
    dominant0 = max((k for k in probe if k in models0), key=lambda k: probe[k])
    eng.events, eng.event_filter = [], {dominant0}
    # per-step durations for the MEDIAN ("median of >= 20 steps"): one HIP event between consecutive steps on the
    # compute stream (K + 1 records per K steps), read after the closing fence
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(a.steps):
        out = step()
        marks[i + 1].record()
    t_enq = time.perf_counter() - t0  # host time to enqueue K steps (kernels run asynchronously)
    fence()
    dt = time.perf_counter() - t0
    # Shader clock / socket power / temperature this line was measured under (sevennet_amd/telemetry.py: librocm_smi64 on a side
    # thread): the two dominant kernels run at the socket's power cap, so a 5 % swing between boxes is a clock swing unless shown
    # otherwise.  Sampled over a REPEAT of the same K steps right behind the timed bracket, not inside it: the first version sampled
    # inside and one step of twenty took 61 instead of 39.8 ms (the SMU query contends with command submission).
    from sevennet_amd.telemetry import Sampler
    with Sampler(dev_id) as tele:
        t0t = time.perf_counter()
        for i in range(a.steps):
            step()
        fence()
        tele_ms = (time.perf_counter() - t0t) / a.steps * 1e3
    per_step = torch.tensor([marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps)], device=dev, dtype=torch.float64)
    if dist_mode:   # a step is over when the slowest rank is
        dist.all_reduce(per_step, op=dist.ReduceOp.MAX)
    per_step = per_step.cpu().numpy()
    step_ms_median = float(np.median(per_step))
    if dist_mode:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    timed_dom = eng.kernel_times_ms().get(dominant0, [])   # recorded by the Python host only
    eng.events, eng.event_filter = [], None
    for _ in range(min(a.steps, 3)):
        eng.compute(graph, halo=halo)
    fence()
    n_break = min(a.steps, 3)
    times = eng.kernel_times_ms()
    eng.events = None

    # ---- per-kernel-class time over the timed region -> dominant kernel + roofline
    totals = {k: float(np.sum(v)) for k, v in times.items()}
    counts = {k: len(v) for k, v in times.items()}
    models = {}
    for ls, L in zip(eng.spec.layers, eng.layers):
        models.update(kernel_model(ls, graph.n_local, graph.n_edges, getattr(L, 'mlp_tail', False), eng.spec.n_basis))
    # classes tagged '@side' ran on the second stream, overlapped with main-stream kernels: their event brackets
    # are not exclusive time, so the dominant class is chosen among the main-stream ones
    dominant = dominant0
    # average launch duration of the dominant class: from the timed steps when the Python host sequenced them,
    # from the breakdown pass when the native sequencer did (its launches are not bracketed)
    avg_ms = float(np.mean(timed_dom)) if len(timed_dom) else totals[dominant] / counts[dominant]
    km = models[dominant]
    if km['bound'] == 'hbm':
        ach = km['bytes'] / (avg_ms * 1e-3) / 1e9
        roof = dict(bound='hbm', kernel=dominant, achieved=ach, peak=HBM_PEAK_GBS, unit='GB/s',
                    frac=ach / HBM_PEAK_GBS, traffic=None, avg_ms=avg_ms,
                    algorithmic_bytes_per_launch=km['bytes'],
                    byte_model='SURVEY.md section 8(d): E (2*4*dx + 12 + 8 + 12) + N 4 dmid for a reverse launch, '
                               'E (4 dx + 12 + 8) + N 4 dmid for a forward launch')
        if 'bytes_incl_staging' in km:   # what the kernel really moves besides (Y, dY, h2, emb, g_emb): a second view
            roof['frac_incl_staging'] = km['bytes_incl_staging'] / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
            roof['bytes_incl_staging_per_launch'] = km['bytes_incl_staging']
        if 'flops' in km:  # fused kernels also carry the radial MLP's last layer on the matrix cores
            n_prod = {4: 3, 3: 6, 2: 3, 1: 1}[a.terms]
            roof['mfma'] = dict(algorithmic_flops_per_launch=km['flops'], bf16_products_per_flop=n_prod,
                                achieved_tflops=km['flops'] * n_prod / (avg_ms * 1e-3) / 1e12, peak_tflops=MFMA_BF16_PEAK_TF,
                                frac=km['flops'] * n_prod / (avg_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TF)
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
        key = dominant.replace('conv_bwd_edge[', 'conv_bwd_edge_vec_').replace('conv_fwd_fused[', 'conv_fwdf_') \
            .replace('conv_bwd_fused[', 'conv_bwdf_').replace('conv_fwd[', 'conv_fwd_') \
            .replace('conv_bwd_node[', 'conv_bwd_node_').rstrip(']')
        # a kernel class may be several kernels (one per x irrep block: <name>_k0, _k1, ...)
        parts = [v for k, v in tr.items() if k == key or k.startswith(key + '_k') or k.startswith(key + '<')]
        if world == 1 and a.reps == 23 and parts:
            roof['traffic'] = sum(v['hbm_bytes_per_launch'] for v in parts)
            roof['traffic_frac_of_peak'] = roof['traffic'] / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
            roof['traffic_source'] = os.path.basename(pmc) + ' (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, gfx950-corrected)'
            # the counters belong to the kernel SOURCE they were collected on: tools/collect_profiles.sh stamps the profile with
            # the sha1 of every generated fused kernel; a kernel edited since then ships its figure marked stale, not silently
            tag_ = dominant.split('[')[-1].rstrip(']')
            now = fused_source_hashes().get(tag_)
            then = (tr.get('__kernel_sources__') or {}).get(tag_)
            roof['traffic_kernel_sha1'] = then
            roof['traffic_stale'] = (then is None) or (now != then)
            if roof['traffic_stale']:
                roof['traffic_note'] = (f'PMC pass predates the current kernel source (profile: {then or "unstamped"}, tree: {now}); '
                                        're-run tools/collect_profiles.sh')
    except Exception:  # noqa: BLE001
        pass
    if any(k.endswith('@side') for k in totals):
        roof['note'] = ("classes tagged '@side' (radial MLPs) run on a second HIP stream concurrently with the "
                        "main-stream kernels: their event brackets overlap the others and are not exclusive time; "
                        "avg_ms of the dominant kernel is measured with that concurrent work present")
    step_ms = dt / a.steps * 1e3
    e_total = out['energy'].clone()
    if dist_mode:  # ranks hold partial energies of their bricks
        dist.all_reduce(e_total)
    roof['kernel_ms_per_step'] = {k: round(v / n_break, 4) for k, v in sorted(totals.items(), key=lambda kv: -kv[1])}
    roof['avg_ms_source'] = 'HIP events inside the timed steps' if len(timed_dom) else 'HIP events of an untimed pass of the same kernels (native host)'
    # whole step: SURVEY.md 8(d)'s algorithmic BYTES against the HBM peak, and the arithmetic the engine actually SCHEDULES (pruned
    # paths, sparse Clebsch-Gordan tensors, split-precision products counted once per matrix-core product issued) against the
    # pipes it runs on.  SURVEY's dense-CG fp32 FLOP count is kept as information only: divided into a step that runs the dense
    # contractions as low-precision splits on the 2.5-PF pipe and prunes unread paths it gave "fractions" above 1.
    if a.model in STEP_WORK:
        sb, sf = (v * n_atoms for v in STEP_WORK[a.model])
        t = dt / a.steps * world   # GPU-seconds per step
        sched = scheduled_work(eng, graph)
        valu_peak = VALU_F32_PEAK_TF * 1e12
        roof['step'] = dict(bytes=sb, frac_hbm=sb / t / (HBM_PEAK_GBS * 1e9), floor_ms_hbm=sb / (HBM_PEAK_GBS * 1e9) * 1e3,
                            mfma_flops_issued=sched['mfma'], valu_flops=sched['valu'],
                            frac_mfma_issued=sched['mfma'] * world / t / (MFMA_BF16_PEAK_TF * 1e12),
                            frac_valu=sched['valu'] * world / t / valu_peak,
                            valu_peak_tflops=VALU_F32_PEAK_TF, valu_measured_tflops_by_waves_per_simd=VALU_F32_MEASURED_TF,
                            flops_survey_dense_fp32=sf,
                            note='bytes: SURVEY.md 8(d) per atom-step (fwd + reverse, no cache credit, radial weights not '
                                 'materialised) over 8 TB/s.  mfma_flops_issued: every 16x16x32 / 32x32x16 product the kernels issue '
                                 '(in-kernel W2 products x3 for f16x3, node linears and hidden radial layers x6 for bf16x6) over the '
                                 '2.5 PFLOP/s dense f16 / bf16 peak.  valu_flops: the sparse tensor-product arithmetic as generated '
                                 '(forward: nnz(C) FMAs + pair products per channel; reverse: Clebsch-Gordan tensor contracted with the '
                                 'harmonics per edge, two FMAs per nonzero (a, c) pair and three per x component per channel) over '
                                 'the 157.3 TFLOP/s fp32 vector peak (SIMD-32: a wave64 instruction issues over 2 cycles); '
                                 'valu_measured_tflops_by_waves_per_simd = what plain v_fma_f32 streams reach on this chip by occupancy '
                                 '(one wave issues a vector instruction every ~7.5 cycles: tools/gpu/issue_probe, '
                                 'profiles/r05_issue_rate_probe.txt).  flops_survey_dense_fp32 = SURVEY.md 8(d)\'s dense-CG count, for '
                                 'reference only.')

    # ghost exchange of this rank per step: L-1 forward (width dx_t) + L-1 reverse exchanges + one force fold; bytes = rows
    # sent + received; time = HIP-event brackets around the exchange calls (with the split exchange the brackets cover
    # only the enqueue + the final wait, i.e. the part that is NOT hidden behind compute)
    halo_ms, halo_n, halo_bytes = 0.0, 0, 0
    if dist_mode:
        halo_ms = sum(v for k, v in totals.items() if k.startswith('halo')) / n_break
        rows = int(sum(len(s_) for s_ in bg.send_lists)) + int(graph.n_total - graph.n_local)
        dims = [ls.si1.dim_out for ls in eng.spec.layers[1:]]
        halo_n = 2 * len(dims) + 1
        halo_bytes = rows * 4 * (2 * sum(dims) + 3)
    if rank == 0:
        res = {
            'metric': 'atom-steps/sec (energy+forces), SevenNet-0 100k-atom cell, 1/2/4/8 MI355X' if a.model == 'sevennet_0'
            else f'atom-steps/sec (energy+forces), {a.model} shape',
            # value: the contract's bracket -- atoms x K / wall time of the K steps between two fences (max over ranks), = atoms /
            # ms_per_step (rounds 1-4 and the driver's own clock use this statistic).  value_median: SURVEY.md 8(d)'s
            # statistic -- atoms / MEDIAN step time (per-step HIP events, max over ranks) -- beside it
            'value': n_atoms * a.steps / dt, 'unit': 'atom-steps/s', 'n_gpus': world, 'steps': a.steps,
            'warmup': a.warmup, 'ms_per_step': step_ms, 'ms_per_step_median': step_ms_median, 'value_median': n_atoms / (step_ms_median * 1e-3),
            'ms_per_step_min_max': [float(per_step.min()), float(per_step.max())],
            'higher_is_better': True, 'scaling': 'strong',
            'vs_baseline': None, 'dtype': DTYPE_LABEL[a.terms if a.fused != 'off' else 0], 'data': 'synthetic',
            'config': {'workload': f'{a.model} shape (5 interaction layers), {n_atoms}-atom periodic diamond-Si '
                                   f'cell (a=5.431 A x {a.reps}^3, {wl_note}), cutoff {cfg["cutoff"]} A, '
                                   f'{n_edges_total} directed edges, seeded synthetic weights',
                       'atoms': n_atoms, 'edges': n_edges_total,
                       'parallelism': 'single GPU' if not dist_mode else f'spatial decomposition x{world}, RCCL halo',
                       'graph_build_s': round(t_graph, 3),
                       'peak_device_memory_gb': round(torch.cuda.max_memory_allocated(dev) / 1e9, 2),
                       'host': a.host, 'host_enqueue_ms_per_step': round(t_enq / a.steps * 1e3, 3),
                       'h2d_in_step': (f'edge_vec [E,3] fp32 = {ev_host.numel() * 4 / 1e6:.1f} MB from pinned host memory every step'
                                       if ev_host is not None else
                                       f'positions [n,3] fp64 = {pos_host.numel() * 8 / 1e6:.2f} MB from pinned host memory every step; '
                                       'edge vectors formed on the GPU (snet_edge_vectors) from the resident topology'
                                       if pos_host is not None else None),
                       'fused': a.fused, 'terms': a.terms,
                       'halo': (None if not dist_mode else ('libsnet_hip RCCL send/recv groups' if type(halo).__name__ == 'NativeHalo'
                                                          else f'torch.distributed all_to_all_single ({backend})')),
                       'halo_overlap': (None if not dist_mode else bool(getattr(halo, 'overlap', True))),
                       'ghost_rows_rank0': (None if not dist_mode else int(graph.n_total - graph.n_local)),
                       'decomposition_selfcheck': (None if not dist_mode else selfcheck),
                       'halo_exposed_ms': (None if not dist_mode else round(halo_ms, 3)),   # main-stream time inside the exchange calls (start: enqueue only; finish: the wait)
                       'halo_ms_per_step_rank0': (None if not dist_mode else round(halo_ms, 4)),
                       'halo_exchanges_per_step': (None if not dist_mode else halo_n),
                       'halo_bytes_per_step_rank0': (None if not dist_mode else halo_bytes),
                       'halo_gbs_rank0': (None if not dist_mode or halo_ms <= 0 else round(halo_bytes / (halo_ms * 1e-3) / 1e9, 2)),
                       'kernel_ms_per_step_rank0': round(sum(v for k, v in totals.items() if not k.startswith('halo')) / n_break, 3),
                       # what the step costs beyond its kernels' own durations: dependent-dispatch gaps, the per-step input copy,
                       # torch fill kernels (main-stream classes only: '@side' brackets overlap the others)
                       'non_kernel_ms_per_step_rank0': round(step_ms_median - sum(v for k, v in totals.items() if not k.startswith('halo') and not k.endswith('@side')) / n_break, 3),
                       'dispatches_per_step_rank0': int(sum(c for k, c in counts.items() if not k.startswith('halo')) / n_break),
                       'energy': float(e_total.cpu()),
                       # the box this line was measured on: medians over the timed region (rank 0's GPU)
                       **tele.summary(), 'telemetry_pass_ms_per_step': round(tele_ms, 3)},
            'roofline': roof,
        }
        if world == 1 and not dist_mode and a.brick_proxy > 1 and a.model == 'sevennet_0':
            res['brick_proxy'] = brick_proxy(a, cfg, sd, eng, pos, cell, dev, modal, step_ms)
        if world == 1 and not a.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline(cfg, sd, a.cpu_reps)
        # RCCL writes a version banner through C stdio, which a pipe buffers until exit: flush it now so that the JSON
        # line is the last thing on stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:  # noqa: BLE001
            pass
        print(json.dumps(res), flush=True)
    if dist_mode:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
