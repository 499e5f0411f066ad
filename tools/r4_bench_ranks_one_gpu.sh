#!/bin/bash
# Dev tool (round 4): bench.py's N > 1 path with the rank processes SHARING the one leased GPU (TUTEL_AMD_BENCH_SHARE_GPU=1: gloo
# rendezvous, IPC transport between the processes), N = 8 / 4 / 2, and the same with the HIP runtime limited to fewer hardware
# queues per process (GPU_MAX_HW_QUEUES) -- to tell a protocol problem from hardware-queue oversubscription of the shared device.
#     bash tools/r4_bench_ranks_one_gpu.sh [quick]      -> gpurun_out/r4n/
mkdir -p gpurun_out/r4n
export TUTEL_AMD_BENCH_SHARE_GPU=1
run() {  # n, label, extra env...
  local n=$1 label=$2; shift 2
  local s=$(date +%s)
  env "$@" timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) \
      bench.py --gpus $n --steps 20 --warmup 5 --settle 20 > gpurun_out/r4n/bench_n${n}_$label.json 2> gpurun_out/r4n/bench_n${n}_$label.err
  local rc=$?
  python - <<PY
import json
try:
    d = json.loads([l for l in open("gpurun_out/r4n/bench_n${n}_$label.json") if l.startswith("{")][0])
    print("N=$n $label rc=$rc wall=$(( $(date +%s) - s ))s  replay %.4f ms/step  eager %.4f ms/step  (%s)" % (d["ms_per_step"], d["launch_modes"]["other"]["ms_per_step"], d["launch_modes"]["timed"]))
except Exception as ex:
    print("N=$n $label rc=$rc: no line (%s)" % ex)
PY
}
if [ "$1" != "quick" ]; then
  run 8 default A=1
  run 4 default A=1
  run 2 default A=1
fi
run 4 hwq2 GPU_MAX_HW_QUEUES=2
run 4 hwq1 GPU_MAX_HW_QUEUES=1
run 4 hwq1_onestream GPU_MAX_HW_QUEUES=1 TUTEL_AMD_EP_STREAMS=1
run 8 hwq2 GPU_MAX_HW_QUEUES=2
run 8 hwq1 GPU_MAX_HW_QUEUES=1
run 8 hwq1_onestream GPU_MAX_HW_QUEUES=1 TUTEL_AMD_EP_STREAMS=1
