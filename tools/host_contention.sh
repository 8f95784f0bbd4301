#!/bin/bash
# The 8-GPU host risk measured on a 1-GPU box: N bench.py ranks (one process each, gloo transport, ALL on GPU 0 through
# DMC_FORCE_DEVICE=0, a small per-rank batch so that the shared GPU is not what is measured, DMC_BENCH_REPLICAS=1 = no gradient
# exchange: gloo would move 45 MB per rank and step through host memory and measure itself) -- each rank reports its CLEAN
# host cost per step (bench.py: every step enqueued on an empty launch queue, all ranks at the same time behind a barrier),
# i.e. the Python + launch cost of a rank while N - 1 others contend for the same host cores.
#   tools/host_contention.sh <out-subdir-of-gpurun_out> [ranks=8] [batch=5]
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT
N=${2:-8}; B=${3:-5}
cd $GRAFT_REPO_ROOT
python bench.py --no-cpu-baseline --batch $B --steps 20 --warmup 5 > $OUT/host_1rank.json 2> $OUT/host_1rank.err
DMC_BENCH_REPLICAS=1 DMC_FORCE_DEVICE=0 DMC_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N \
    --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus $N --batch $B --steps 20 --warmup 5 --no-cpu-baseline \
    > $OUT/host_${N}ranks.json 2> $OUT/host_${N}ranks.err
python - <<PY
import json
def last_json(path):        # gloo prints its rendezvous lines to stdout: take the bench line
    for line in reversed(open(path).read().splitlines()):
        i = line.find('{"metric"')
        if i >= 0:
            return json.loads(line[i:])
a = last_json("$OUT/host_1rank.json"); b = last_json("$OUT/host_${N}ranks.json")
json.dump({"one_rank": {k: a[k] for k in ("host_clean_ms_per_step", "host_enqueue_ms_per_step", "ms_per_step")},
           "ranks": $N, "batch_per_rank": $B, "ms_per_step": b["ms_per_step"], "comm": b["comm"]}, open("$OUT/host_contention.json", "w"), indent=1)
print("1 rank : clean host %.3f ms/step, GPU step %.3f ms (batch $B)" % (a["host_clean_ms_per_step"], a["ms_per_step"]))
c = b["comm"]
print("$N ranks: clean host per rank", c["host_clean_ms_per_step_by_rank"], "cores usable", c["host_cores_usable"], "step", b["ms_per_step"])
PY
