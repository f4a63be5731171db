#!/bin/bash
# socket power / shader clock while ONE kernel runs back to back (is the dominant kernel at the power cap?):
#   gpurun --timeout 900 -- bash tools/gpu/power_probe.sh ["name|microbench flags" ...]
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/power_probe.txt
python -c "import torch; torch.zeros(1, device='cuda')" >/dev/null 2>&1
if [ $# -eq 0 ]; then set -- "conv_bwd_fused[22|" "conv_fwd_fused[22|" "conv_bwd_fused[22|--zeros" "conv_fwd_fused[22|--zeros" "segment_sum_rows|" "si2_fwd|"; fi
{
echo "== idle"; rocm-smi --showpower --showmaxpower --showclocks 2>&1 | grep -E "Power \(W\)|sclk|mclk|fclk" | head -8
for SPEC in "$@"; do
  K=${SPEC%%|*}; FL=${SPEC#*|}
  echo "== while running $K $FL back to back"
  timeout 50 python tools/microbench.py --layer 1 --model sevennet_0 --terms 4 --iters 1000000 --only "$K" $FL > gpurun_out/power_mb.log 2>&1 &
  PID=$!
  sleep 28
  for i in 1 2 3 4; do rocm-smi --showpower --showclocks 2>&1 | grep -E "Power \(W\)|sclk" | sed 's/GPU\[0\]\t*: //' | tr '\n' ' '; echo; sleep 1.5; done
  wait $PID
done
} > $OUT 2>&1
cat $OUT
