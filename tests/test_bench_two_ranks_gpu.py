"""``bench.py --gpus 2`` launched as the driver launches it (torch.distributed.run, one process per rank), both ranks on GPU 0
over gloo (the test hooks DMC_FORCE_DEVICE / DMC_DIST_BACKEND; RCCL needs one GPU per rank): every --config prints ONE JSON
line whose ``comm`` object has the same keys, names the communicator and reports what the last step exchanged.  Replaces
``torch.nn.DataParallel`` of code/dmcnet/train.py:117 / code/dmcnet_GAN/train.py:118 and the reference I3D's DataParallel wrap."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = {"backend", "world_size", "reduce_op", "last_step_bytes_by_set", "last_step_launched_from", "exposed_wait_ms_per_step",
          "ms_per_step_by_rank", "ms_per_step_rank_min", "ms_per_step_rank_max", "exposed_wait_ms_per_step_rank_max",
          "host_clean_ms_per_step_by_rank", "host_cores_usable", "device_of_rank0"}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(extra, env_extra=None):
    env = dict(os.environ, DMC_FORCE_DEVICE="0", DMC_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(env_extra or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline"] + extra
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l[l.find('{"metric"'):] for l in p.stdout.splitlines() if '{"metric"' in l]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.slow
@pytest.mark.parametrize("config,extra,sets", [
    ("dmcnet", ["--batch", "2"], {"base_model", "gen_flow_model"}),
    ("gan", ["--batch", "2"], {"base_model", "discriminator"}),          # the last step bench.py runs (a clean-host probe, index 8) is a D step
    ("i3d", ["--batch", "1", "--clip-length", "16"], None)])
def test_two_ranks_print_the_same_comm_object(config, extra, sets):
    line = _run(["--config", config] + extra)
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "dp2" and line["scaling"] == "weak"
    c = line["comm"]
    assert COMMON <= set(c), sorted(COMMON - set(c))
    assert c["backend"] == "gloo" and c["world_size"] == 2 and len(c["ms_per_step_by_rank"]) == 2
    assert sum(c["last_step_bytes_by_set"].values()) > 0
    if sets is not None:
        assert set(c["last_step_bytes_by_set"]) == sets, c["last_step_bytes_by_set"]
    assert "allreduce" not in c


def test_stubbed_allreduce_is_recorded_and_refused_without_the_test_flag():
    line = _run(["--batch", "2"], {"DMC_BENCH_STUB_ALLREDUCE": "1", "DMC_BENCH_TEST_HOOKS": "1"})
    assert line["comm"]["allreduce"] == "stubbed" and "STUBBED" in line["config"]["parallelism"]
    env = dict(os.environ, DMC_FORCE_DEVICE="0", DMC_DIST_BACKEND="gloo", DMC_BENCH_STUB_ALLREDUCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("DMC_BENCH_TEST_HOOKS", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
           "--batch", "2", "--no-cpu-baseline"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and "DMC_BENCH_TEST_HOOKS" in (p.stderr + p.stdout)
