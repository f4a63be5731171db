#!/bin/bash
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys; sys.path.insert(0,'tests')
import numpy as np
from helpers import load_ts_golden
from sevennet_amd.model_file import write_model_file
from test_lammps_glue_gpu import _write_structure
d,cfg,sd=load_ts_golden('hfo2_96')
write_model_file('/tmp/m.snet',cfg,sd)
_write_structure('/tmp/s.txt', d['types']+1, d['pos'], d['cell'], 2, 1.0)
PY
cd /tmp
timeout 300 /opt/rocm/bin/rocgdb -batch -ex run -ex bt --args $GRAFT_REPO_ROOT/tests/lammps_mock/run_pair e3gnn/parallel /tmp/s.txt /tmp/o.json -- '*' '*' /tmp/m.snet Hf O 2>&1 | tail -40
