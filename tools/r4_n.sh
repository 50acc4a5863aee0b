#!/bin/bash
# the driver's launch shapes for N > 1, as far as one GPU allows: torchrun with one nccl rank; two ranks over gloo sharing the GPU; dr_group in-process with two ranks on the device
mkdir -p gpurun_out
export TMPDIR=/tmp
A="--steps 20 --warmup 5 --repeats 3 --no-traffic --no-cpu-baseline --no-extras"
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 $A > gpurun_out/r4n_torchrun1.json 2> gpurun_out/r4n_torchrun1.err; echo "torchrun 1 rank rc=$?"
DOGERAY_BENCH_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 $A > gpurun_out/r4n_torchrun2_gloo.json 2> gpurun_out/r4n_torchrun2_gloo.err; echo "torchrun 2 ranks gloo rc=$?"
DOGERAY_GROUP_DEVICES=0,0 timeout -k 10 300 python3 bench.py --gpus 2 $A > gpurun_out/r4n_group2.json 2> gpurun_out/r4n_group2.err; echo "group 2 ranks rc=$?"
python3 - <<'PY'
import json
for f in ("r4n_torchrun1", "r4n_torchrun2_gloo", "r4n_group2"):
    try:
        j = json.loads(open("gpurun_out/%s.json" % f).read().strip().split("\n")[-1])
        print(f, j["n_gpus"], round(j["value"], 1), round(j["ms_per_step"], 4), j.get("transport"), j.get("assembled_frame_identical_to_one_context", j.get("gathered_identical")), {k: v for k, v in j.items() if "identical" in k})
    except Exception as e:
        print(f, "FAILED", e)
PY
tail -3 gpurun_out/r4n_torchrun2_gloo.err
