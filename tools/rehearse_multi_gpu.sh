#!/bin/bash
# The N > 1 branches of bench.py rehearsed on ONE GPU: two ranks, gloo collectives on the host, both on device 0 (the
# launcher starts before anything touches the GPU).  Weak scaling (dyads) and strong scaling (window ranges of one
# dyad).  The rates mean nothing (two processes share one GPU); what is checked is that the code path runs end to end:
# process group, shards, band sums, gather, max-reduce of the time, cpu_baseline on rank 0.
#     gpurun --timeout 900 -- 'bash tools/rehearse_multi_gpu.sh r03'
set -eo pipefail
TAG=${1:-r03}
mkdir -p gpurun_out
for mode in dyads windows; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
      bench.py --gpus 2 --steps 3 --warmup 1 --backend gloo --shard $mode 2> gpurun_out/rehearsal_${TAG}_$mode.err \
      | grep '^{' > gpurun_out/rehearsal_${TAG}_$mode.json
  python - <<PY
import json
d = json.load(open("gpurun_out/rehearsal_${TAG}_$mode.json"))
print("$mode", d["n_gpus"], d["scaling"], round(d["value"]), "cpu_baseline" in d and d["cpu_baseline"]["value"], d["config"]["parallelism"], d["config"]["gather"])
PY
done
