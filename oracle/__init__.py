"""CPU oracle for the SevenNet energy/force hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain-PyTorch (CPU, fp32/fp64 switchable) restatement of the
arithmetic the reference runs per MD step: e3nn's lowered
`index -> per-path einsum -> scatter_add -> autograd` pipeline, written from the
published e3nn semantics because e3nn itself is not vendored under
/root/reference (pyproject.toml:24 pins only `e3nn>=0.5.0`).

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import it, and only as the checker / the timed CPU baseline --
never as part of the shipped force engine (`sevennet_amd`), which must fail
loudly when its HIP library is missing.

Parity pins (see tests/test_oracle_golden.py, oracle/tools/make_golden.py):
  * real Wigner-3j for l<=2: the 8 `_w3j_*` buffers stored by e3nn inside the
    reference's own test checkpoint tests/data/checkpoints/cp_0.pth;
  * whole-model arithmetic (Bessel, poly cutoff, SH l<=1, radial MLP incl.
    normalize2mom, uvu tensor product, scatter, o3.Linear, FCTP self-connection,
    Gate, rescale, edge-gradient forces, segment/ghost plumbing): outputs of the
    reference's deployed TorchScript example models
    (example_inputs/md_{serial,parallel}_example) run in the build container.
  * UNPINNED here (no data, no executable): l=3 Wigner-3j / spherical
    harmonics, normalised-SH path, XPLOR cutoff, `linear` self-connection.
    Those follow the same generators validated on l<=2 and are checked by
    equivariance + fp64 finite differences only.  Multi-modal linears and
    modal-wise rescale (sevenn/nn/linear.py:66-92, scale.py:196-363) restate the
    in-tree Python literally; their STRUCTURE is pinned by the reference's
    parameter counts (tests/unit_tests/test_model.py:185-212), their numerics
    are unpinned.
"""
