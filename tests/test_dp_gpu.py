"""GPU: the N-rank flow of bench.py on the one GPU of the test box - two ranks SHARING the device (gloo collectives), spawned as
a child process group: process-group init, per-rank graph capture next to a live process group, the flat-buffer gradient
all-reduce + averaging, barrier / MAX-over-ranks timing and the rank-0 JSON line.  Round 1 ended with this configuration
aborting with HSA_STATUS_ERROR_EXCEPTION 0x1016; the out-of-bounds access behind that signature (a NaN matching cost left the
device LSAP's arg-min at its sentinel, which the mask-loss kernels then used as a row index) is closed in csrc/lsap.hip, and the
200-step run of tools/dp2_bisect.sh is clean."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_sharing_the_gpu_run_the_bench_flow():
    env = dict(os.environ, COMBO_SINGLE_DEVICE="1", COMBO_DIST_BACKEND="gloo", COMBO_MIOPEN_BENCHMARK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29633", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--clips", "2",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert "HSA_STATUS_ERROR" not in r.stderr, r.stderr[-2000:]
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-1000:]  # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0 and out["config"]["global_batch_clips"] == 4
    # the line proves what ran: every rank's device identity, the process group's size, the measured all-reduce time
    # (two ranks on one GPU cannot use RCCL - "Duplicate GPU detected" - so this functional run travels over gloo and the line
    # says so: `world_size` + `dist_backend`; `rccl_world_size` appears only when RCCL carried the collectives)
    assert out["dist_backend"] == "gloo" and out["world_size"] == 2 and "rccl_world_size" not in out
    assert [r_["rank"] for r_ in out["ranks"]] == [0, 1]
    assert all(r_["pci"] and r_["name"] and r_["pid"] > 0 for r_ in out["ranks"]) and out["ranks"][0]["pid"] != out["ranks"][1]["pid"]
    assert out["distinct_devices"] == 1 and out["shared_device_run"] is True  # two ranks on ONE GPU: allowed only by COMBO_SINGLE_DEVICE=1
    ar = out["all_reduce"]
    assert ar["all_reduce_ms_per_step"] > 0 and ar["collectives_per_step"] >= 1 and ar["bytes_per_step"] > 300e6, ar


def test_two_ranks_on_one_device_are_refused_without_the_waiver():
    """An N-rank run whose ranks share a physical device must not produce an "N-GPU" line: without COMBO_SINGLE_DEVICE=1 every
    rank raises after the device census (all_gather_object of UUID / PCI address)."""
    env = dict(os.environ, COMBO_DIST_BACKEND="gloo", COMBO_MIOPEN_BENCHMARK="0", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT,
               )
    env.pop("COMBO_SINGLE_DEVICE", None)
    # LOCAL_RANK 0 and 1 both map to physical device 0: set_device(local_rank) would fail on a 1-GPU box, so both ranks are
    # given local rank 0 by a wrapper environment - torch.distributed.run sets LOCAL_RANK itself, hence the census is what refuses
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29635", os.path.join(ROOT, "tests", "dp_census_child.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode != 0 and "distinct devices" in r.stderr, r.stderr[-2000:]


def _json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, stdout[-1000:]  # rank 0 only
    return json.loads(lines[0])


def test_bench_gpus_2_spawns_two_ranks_by_itself():
    """`python bench.py --gpus 2` - the form the driver uses at N = 1, no launcher - must start two ranks itself
    (bench.spawn_ranks: a child process group created before the parent touches the GPU) and relay rank 0's line."""
    env = dict(os.environ, COMBO_SINGLE_DEVICE="1", COMBO_DIST_BACKEND="gloo", COMBO_MIOPEN_BENCHMARK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--clips", "2",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    out = _json_line(r.stdout)
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2" and out["config"]["global_batch_clips"] == 4
    assert out["value"] > 0


def test_bench_gpus_2_over_rccl_when_two_devices_exist():
    """The same form on the `nccl` (= RCCL) backend, one rank per device: runs the first time a box has >= 2 GPUs."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL over xGMI)")
    env = dict(os.environ, COMBO_MIOPEN_BENCHMARK="0", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "COMBO_SINGLE_DEVICE", "COMBO_DIST_BACKEND"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    out = _json_line(r.stdout)
    assert out["n_gpus"] == 2 and out["config"]["global_batch_clips"] == 16 and out["value"] > 0


def test_rccl_all_reduces_the_flat_gradient_buffer_on_one_gpu():
    """RCCL itself, executed: a ONE-rank `nccl` process group on the one GPU (COMBO_FORCE_PG=1), so that `bench.py` initialises
    RCCL (`dist.init_process_group("nccl", device_id=...)`), captures its graphs next to the live communicator and runs the
    348 MB flat-buffer all-reduce (+ the overlapped head-region one on the side stream, + the `num_masks` one) through
    `ncclAllReduce` every step - the code path of `train_net.py:284-291` / d2 `launch` at N = 8, at world size 1.  A fresh child
    process, as the other tests of this file."""
    env = dict(os.environ, COMBO_FORCE_PG="1", COMBO_MIOPEN_BENCHMARK="0", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT,
               MASTER_ADDR="127.0.0.1", MASTER_PORT="29641", NCCL_DEBUG="VERSION")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "COMBO_SINGLE_DEVICE", "COMBO_DIST_BACKEND"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2", "--clips", "2",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    out = _json_line(r.stdout)
    assert out["n_gpus"] == 1 and out["value"] > 0
    assert out["config"].get("collective") == "nccl all-reduce (forced one-rank process group)", out["config"]  # "nccl" IS RCCL on ROCm
    assert out["config"]["launch"].startswith("2 hipGraphs"), out["config"]["launch"]  # the overlapped two-region flow ran
