"""CPU: `python bench.py --gpus N` (N > 1, no launcher around it) must hand the run to N child ranks through
torch.distributed.run BEFORE the parent touches the GPU, with the command line the driver itself uses, and must refuse loudly
when the box has fewer devices than ranks (instead of silently running one rank and printing `n_gpus: 1`)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_spawn_builds_the_drivers_command_line(monkeypatch):
    import bench
    seen = {}

    class FakeProc:
        stdout = iter(['{"metric": "x"}\n'])

        def wait(self):
            return 0

    def fake_popen(cmd, **kw):
        seen["cmd"], seen["env"] = cmd, kw["env"]
        return FakeProc()
    monkeypatch.setattr(subprocess, "Popen", fake_popen)
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 8)
    monkeypatch.delenv("MASTER_PORT", raising=False)
    rc = bench.spawn_ranks(4, ["--gpus", "4", "--steps", "3", "--warmup", "1"])
    assert rc == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-7:] == [os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_refuses_more_ranks_than_devices():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("box has >= 2 devices")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "COMBO_SINGLE_DEVICE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 2 and "--gpus 2 but only" in r.stderr
    assert '"metric"' not in r.stdout


def test_census_refuses_ranks_that_share_a_device():
    """bench.census_verdict: the table every rank gathers (rank, device UUID / PCI address) must hold N distinct devices"""
    import pytest
    import bench
    a = {"rank": 0, "uuid": "GPU-aa", "pci": "0000:05:00"}
    b = {"rank": 1, "uuid": "GPU-bb", "pci": "0000:15:00"}
    ok = bench.census_verdict([a, b], shared=False)
    assert ok["distinct_devices"] == 2 and not ok["shared_device_run"]
    with pytest.raises(RuntimeError, match="distinct devices"):
        bench.census_verdict([a, dict(a, rank=1)], shared=False)
    waived = bench.census_verdict([a, dict(a, rank=1)], shared=True)
    assert waived["distinct_devices"] == 1 and waived["shared_device_run"]
    # no UUID exposed by this torch: the PCI address alone decides
    c, d = {"rank": 0, "uuid": None, "pci": "0000:05:00"}, {"rank": 1, "uuid": None, "pci": "0000:25:00"}
    assert bench.census_verdict([c, d], shared=False)["distinct_devices"] == 2
    # a torch build that exposes neither: the run is NOT refused (advisor, round 5) - the line says the identity is unverifiable
    e = {"rank": 0, "uuid": None, "pci": "unverifiable:host:HIP_VISIBLE_DEVICES=:0", "verifiable": False}
    f = {"rank": 1, "uuid": None, "pci": "unverifiable:host:HIP_VISIBLE_DEVICES=:1", "verifiable": False}
    v = bench.census_verdict([e, f], shared=False)
    assert v["distinct_devices"] == 2 and v["device_identity"] == "unverifiable"
    with pytest.raises(RuntimeError, match="distinct"):
        bench.census_verdict([e, dict(e, rank=1)], shared=False)
