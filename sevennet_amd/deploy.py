"""`python -m sevennet_amd.deploy <checkpoint.pth> [-o model.snet]`

Engine-side counterpart of `sevenn get_model` (sevenn/main/sevenn_get_model.py,
sevenn/scripts/deploy.py:16-76): turns a reference checkpoint into the `.snet` file that
`pair_style e3gnn` / `e3gnn/parallel` of lammps/pair_e3gnn_hip.cpp (or any `snet_model_load`
caller) consumes.  One file serves both styles -- there are no per-layer segments to deploy
(`get_parallel`, deploy.py:80-170), the ghost exchange points are inside `snet_model_eval`.
Runs without a GPU.
"""
from __future__ import annotations

import argparse
import sys


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog='python -m sevennet_amd.deploy', description=__doc__.split('\n\n')[1])
    ap.add_argument('checkpoint', help='reference checkpoint (.pth with config + model_state_dict)')
    ap.add_argument('-o', '--output', default='deployed_model.snet')
    ap.add_argument('-m', '--modal', default=None, help='fidelity channel of a multi-modal checkpoint (sevenn get_model -m)')
    a = ap.parse_args(argv)
    from .calculator import load_reference_checkpoint
    from .model_file import write_model_file
    from .model_spec import build_model_spec
    cfg, sd = load_reference_checkpoint(a.checkpoint)
    if cfg.get('use_modality') and a.modal is None:
        print(f"Modal is not given. It has: {list((cfg.get('_modal_map') or {}).keys())}", file=sys.stderr)
        return 2
    out = a.output if a.output.endswith('.snet') else a.output + '.snet'
    write_model_file(out, cfg, sd, modal=a.modal)
    sp = build_model_spec(cfg)
    tags = ', '.join(ls.conv.tag for ls in sp.layers)
    print(f'wrote {out}: {len(sp.layers)} interaction layers, cutoff {sp.cutoff}, tensor-product shapes [{tags}]')
    print('the shapes must be compiled into libsnet_hip.so (sevennet_amd/shapes.py::aot_configs); '
          'snet_model_load names a missing one')
    return 0


if __name__ == '__main__':
    raise SystemExit(main())
