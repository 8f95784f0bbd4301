"""One rank of the two-ranks-on-one-GPU parity test (tests/test_hip_parity_full.py).

Launched twice with RANK=0/1, WORLD_SIZE=2: both processes use cuda:0, exchange gradients over a
gloo group (RCCL needs one GPU per rank) and run the real ``Model`` + HIP autograd + the
per-optimizer ``GradBucketReducer`` + fused Adam on the bucket views, i.e. the code path of
``bench.py --gpus N`` -- code/dmcnet/train.py:117,221-266 and code/dmcnet_GAN/train.py:261-371 with
DataParallel replaced by one process per rank.  The helper functions are also imported by the test
itself to evaluate the two shards one after the other in a single process (the expected result).
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import dmcnet_amd                                # noqa: E402
from dmcnet_amd import ddp, train as T           # noqa: E402
from oracle import dmc_oracle as O               # noqa: E402  (seeded fills / synthetic inputs only)

DEV = "cuda:0"
HP = dict(lr=0.01, weight_decay=1e-4, lr_cls_mult=0.01, lr_mse_mult=1.0)
KW = dict(base_model="resnet18", use_databn=0, gen_flow_or_delta=1, arch_estimator="DenseNetTiny")


def build(gan):
    m = dmcnet_amd.Model(51, 3, "mv", arch_d="Discriminator3" if gan else None, **KW)
    o = O.OracleModel(51, 3, "mv", arch_d="Discriminator3" if gan else None, **KW)
    m.load_state_dict(O.seeded_state_fill(o, 171).state_dict())
    return m


def shard_batches(rank):
    """Clips [2 rank, 2 rank + 2) of two seeded 4-clip batches (first step, second step)."""
    out = []
    for seed, ds in ((172, 16), (173, 0)):
        b = O.synthetic_batch(seed=seed, batch=4, num_segments=3, num_class=51, flow_ds_factor=ds)
        out.append(tuple(t[2 * rank:2 * rank + 2].contiguous() for t in b))
    return out


def _grads(model):
    return {k: p.grad.detach().clone().cpu() for k, p in model.named_parameters() if p.grad is not None}


def run_phases(model, batches, gan, reducer, do_step, shard, state_after_first=None, local=None):
    """dmcnet: one step ('step').  gan: a D step then a G step ('D', 'G').  ``do_step=False`` leaves
    the weights alone (gradients only); ``state_after_first`` is then loaded before the second phase.
    ``local``: dict filled by hooks with this rank's gradients BEFORE the exchange."""
    res = {"grads": {}, "bytes": {}, "where": {}, "local": {}}
    if gan:
        step = T.GanTrainStep(model, 3, 1.0, 1.0, 0.01, 10.0, lr_d_mult=1.0, reducer=reducer, **HP)
        opts = (step.optimizer_cls, step.optimizer_gf, step.optimizer_d)
    else:
        step = T.DmcnetTrainStep(model, 3, 1.0, 10.0, reducer=reducer, **HP)
        opts = (step.optimizer_cls, step.optimizer_gf)
    if not do_step:
        for o in opts:
            o.step = lambda *a, **k: None
    phases = (("D", 0), ("G", 1)) if gan else (("step", 0),)
    for tag, i in phases:
        if i == 1 and state_after_first is not None:
            model.load_state_dict(state_after_first)
        batch = tuple(t.to(DEV) for t in batches[i])
        if gan:
            # masks are a function of (shard, phase) only, so that the single-process evaluation of a
            # shard sees the same Dropout2d draw as the rank that owned it
            n = 12 if i == 0 else 6
            model.discriminator.forced_masks = O.seeded_dropout_masks(300 + 2 * shard + i, model.discriminator, n)
            step.step(batch, i)
        else:
            step.step(batch)
        res["grads"][tag] = _grads(model)
        if local is not None:
            res["local"][tag] = dict(local)
            local.clear()
        if reducer is not None:
            res["bytes"][tag] = reducer.reduced_bytes(by_set=True)
            res["where"][tag] = [w for _, _, _, w in reducer.last_reduced]
        if i == 0:
            res["state_after_first"] = {k: v.detach().clone().cpu() for k, v in model.state_dict().items()}
    return res


def main():
    import torch.distributed as dist
    phase, out_dir = sys.argv[1], sys.argv[2]
    rank = int(os.environ["RANK"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=int(os.environ["WORLD_SIZE"]))
    gan = phase == "gan"
    model = build(gan).to(DEV).train()
    # hooks registered BEFORE the reducer's run first: they keep this rank's own gradients
    local = {}
    for k, p in model.named_parameters():
        p.register_post_accumulate_grad_hook(lambda q, k=k: local.__setitem__(k, q.grad.detach().clone().cpu()))
    reducer = ddp.for_model(model)
    res = run_phases(model, shard_batches(rank), gan, reducer, do_step=True, shard=rank, local=local)
    res["params"] = {k: p.detach().clone().cpu() for k, p in model.named_parameters()}
    torch.save(res, os.path.join(out_dir, "%s_r%d.pt" % (phase, rank)))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
