#!/bin/bash
# The 8-GPU host risk measured on a 1-GPU box: N bench.py ranks (one process each, gloo transport, ALL on GPU 0 through
# DMC_FORCE_DEVICE=0, a small per-rank batch so that the shared GPU is not what is measured).  Each rank reports its CLEAN host
# cost per step (bench.py: every step enqueued on an empty launch queue, all ranks at the same time behind a barrier), i.e. the
# Python + launch cost of a rank while N - 1 others contend for the same host cores.  Three runs of the N ranks:
#   replicas   DMC_BENCH_REPLICAS=1: independent replicas, no reducer at all (the round-4 figure: the floor)
#   stub       the gradient reducer ON (74 post-accumulate hooks, bucket copies, side-stream joins, waits) with all_reduce
#              returning a completed Work at once: the reducer's HOST cost without gloo's host-memory transport
#   gloo       the reducer ON over gloo (45 MB per rank and step through host memory: an upper bound no RCCL run pays)
#   tools/host_contention.sh <out-subdir-of-gpurun_out> [ranks=8] [batch=2] [further bench.py arguments, e.g. --config i3d --clip-length 64]
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT
N=${2:-8}; B=${3:-2}
shift; shift; shift
EXTRA="$@"
cd $GRAFT_REPO_ROOT
python bench.py --no-cpu-baseline --batch $B --steps 20 --warmup 5 $EXTRA > $OUT/host_1rank.json 2> $OUT/host_1rank.err
run() {   # name, extra environment
  env $2 DMC_FORCE_DEVICE=0 DMC_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N \
      --master-addr 127.0.0.1 --master-port $3 bench.py --gpus $N --batch $B --steps 20 --warmup 5 --no-cpu-baseline $EXTRA \
      > $OUT/host_${N}ranks_$1.json 2> $OUT/host_${N}ranks_$1.err
}
run replicas DMC_BENCH_REPLICAS=1 29611
run stub "DMC_BENCH_STUB_ALLREDUCE=1 DMC_BENCH_TEST_HOOKS=1" 29612
run gloo DMC_NOTHING=1 29613
python - <<PY
import json
def last_json(path):        # gloo prints its rendezvous lines to stdout: take the bench line
    for line in reversed(open(path).read().splitlines()):
        i = line.find('{"metric"')
        if i >= 0:
            return json.loads(line[i:])
a = last_json("$OUT/host_1rank.json")
out = {"one_rank": {k: a[k] for k in ("host_clean_ms_per_step", "host_enqueue_ms_per_step", "ms_per_step")}, "ranks": $N, "batch_per_rank": $B, "bench_args": "$EXTRA"}
print("1 rank : clean host %.3f ms/step, GPU step %.3f ms (batch $B)" % (a["host_clean_ms_per_step"], a["ms_per_step"]))
for name in ("replicas", "stub", "gloo"):
    b = last_json("$OUT/host_${N}ranks_%s.json" % name)
    if b is None:
        print(name, ": no bench line"); continue
    c = b["comm"]
    out[name] = {"ms_per_step": b["ms_per_step"], "comm": c}
    print("$N ranks, %-8s: clean host per rank %s  cores usable %s  step %.3f  exposed wait %s" % (
        name, c["host_clean_ms_per_step_by_rank"], c["host_cores_usable"], b["ms_per_step"], c.get("exposed_wait_ms_per_step_rank_max")))
json.dump(out, open("$OUT/host_contention.json", "w"), indent=1)
PY
