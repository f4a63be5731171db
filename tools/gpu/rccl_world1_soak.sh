#!/bin/bash
# The exact entry path of `bench.py --gpus N` (torch.distributed.run launch, nccl process group, RCCL unique id made by
# libsnet_hip.so and broadcast through the process group, the halo's communicator, the brick graph, every per-layer
# exchange as an ncclGroup with zero-size sends, the energy all-reduce) on REAL RCCL at world size 1 -- the only size
# a one-GPU box offers.  Python host and native sequencer, with and without the second halo stream.
set -o pipefail
for extra in "" "--no-halo-overlap" "--host native"; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 1 --dist-path --halo native --no-cpu-baseline --steps 5 --warmup 2 $extra 2> gpurun_out/soak.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('soak [$extra]', round(d['ms_per_step'],2), 'ms/step', c['parallelism'], c['halo'], 'exchanges/step', c['halo_exchanges_per_step'], 'halo ms', c['halo_ms_per_step_rank0'], 'E', c['energy'])" || { echo "soak [$extra] FAILED"; tail -5 gpurun_out/soak.err; }
done 2>&1 | tee gpurun_out/rccl_world1_soak.log
timeout 300 python bench.py --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('single-process reference', round(d['ms_per_step'],2), 'ms/step  E', d['config']['energy'])" | tee -a gpurun_out/rccl_world1_soak.log
