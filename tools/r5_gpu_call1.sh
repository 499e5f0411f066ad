#!/bin/bash
# round 5, GPU call 1: the new multi-rank safety work on the one GPU (ranks sharing the device) + the visibility reproducer + bench lines
cd "$(dirname "$0")/.." || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5c1
mkdir -p $O
python -c "import torch; print(torch.cuda.get_device_name(0))" > $O/device.txt 2>&1
echo "== selfcheck / canary / capacity / stress tests" 
timeout 900 python -m pytest tests/test_ep_ipc_one_gpu.py -x -q --durations=25 -k "selfcheck or canaries or capacity_changes or bounded_wait or stress_worker or ranks_sharing" > $O/pytest_ipc_new.log 2>&1; echo "rc=$?" >> $O/pytest_ipc_new.log
tail -5 $O/pytest_ipc_new.log
echo "== two-process visibility reproducer"
GPU_MAX_HW_QUEUES=2 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 tools/r5_two_process_visibility.py $O/r05_two_process_visibility.txt > $O/visibility.log 2>&1; echo "rc=$?" >> $O/visibility.log
tail -15 $O/visibility.log
echo "== bench N=1"
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; echo "rc=$?"
tail -c 600 $O/bench_n1.err
echo "== bench N=2 sharing the GPU, bare python"
TUTEL_AMD_BENCH_SHARE_GPU=1 GPU_MAX_HW_QUEUES=2 timeout 900 python bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_n2_share.json 2> $O/bench_n2_share.err; echo "rc=$?"
tail -c 1500 $O/bench_n2_share.err
python - <<'PY'
import json
for f in ("bench_n1.json", "bench_n2_share.json"):
    try:
        d = json.loads([l for l in open("gpurun_out/r5c1/" + f) if l.startswith("{")][-1])
        print(f, d.get("value"), d.get("ms_per_step"), d.get("parity"), [ (m["transport"], m["a2a_ffn_overlap_degree"], m["value"], (m["parity"] or {}).get("ok")) for m in d.get("ep_modes") or []])
        if "extra" in d: print({k: (v if len(str(v)) < 400 else str(v)[:400]) for k, v in d["extra"].items()})
    except Exception as ex:
        print(f, "unreadable:", ex)
PY
