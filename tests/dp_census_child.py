"""child of tests/test_dp_gpu.py::test_two_ranks_on_one_device_are_refused_without_the_waiver: two gloo ranks, both on cuda:0,
run bench.device_census - it must raise on every rank"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import bench  # noqa: E402

dist.init_process_group("gloo")
torch.cuda.set_device(0)
try:
    bench.device_census(dist.get_rank(), int(os.environ.get("LOCAL_RANK", "0")), torch.device("cuda", 0))
finally:
    dist.destroy_process_group()
