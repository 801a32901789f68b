"""The optional all-gather of per-view (depth, confidence) maps on DEVICE tensors over RCCL (backend "nccl"), one process
per GPU -- SURVEY.md 8e, BASELINE config 5's exchange step.  Needs two GPUs in the box: skipped on the one-GPU test
boxes (no 1 -> 8 scaling curve has been measured by the build; DESIGN.md section 6 says so)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _view(i, h=37, w=53):
    g = torch.Generator().manual_seed(4000 + i)
    return torch.rand((2, h, w), generator=g) + i


def _worker(rank, world, port, n_views, out_dir):
    sys.path.insert(0, ROOT)
    from deep3d_aerial_amd import sharding

    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    sharding.init_from_env("nccl")
    full = sharding.run_sharded(lambda i: _view(i).cuda(), n_views, gather=True)
    assert full.is_cuda
    torch.save(full.cpu(), os.path.join(out_dir, "rank%d.pt" % rank))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("n_views", [5, 8])
def test_all_gather_maps_on_devices(tmp_path, n_views):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL needs one device per rank)")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, n_views, str(tmp_path)), nprocs=2, join=True)
    want = torch.stack([_view(i) for i in range(n_views)])
    for rank in range(2):
        assert torch.equal(torch.load(os.path.join(str(tmp_path), "rank%d.pt" % rank)), want)


def test_bench_two_ranks(tmp_path):
    """bench.py's N > 1 path as the driver launches it (torch.distributed.run, one process per rank): process-group set-up,
    the two barriers, the MAX over ranks and rank 0's JSON line.  With one GPU in the box the two ranks share it (gloo for
    the control-plane collectives, noted in the line); with two or more it is the real one-process-per-GPU RCCL run."""
    import json
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if torch.cuda.device_count() < 2:
        env["D3D_BENCH_SHARE_GPU"] = "1"    # bench.py refuses to share a card unless asked to
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                       # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["roofline"]["bound"] == "valu+lds" and d["roofline"]["priced_against"] == "hbm" and "cpu_baseline" not in d and "secondary" not in d
    # BASELINE config 5's exchange: the all-gather of the maps ahead of fusion, timed outside the step loop
    ex = d["exchange"]
    assert ex["allgather_ms"] > 0 and ex["content_ok"] and ex["shape_per_rank"][1:] == [2, 3712, 2752]
    assert ex["backend"] == ("gloo" if torch.cuda.device_count() < 2 else "nccl")
    assert d["config"].get("ranks_share_one_gpu", False) == (torch.cuda.device_count() < 2)
    # every rank reports its own kernel time, wall time and device, so a first multi-GPU run can be read rank by rank
    pr = d["per_rank"]
    assert [r["rank"] for r in pr] == [0, 1]
    assert all(r["kernel_ms"] > 0 and r["elapsed_s"] > 0 and isinstance(r["device"], str) and r["device"] for r in pr)
    if torch.cuda.device_count() >= 2:
        assert pr[0]["device_index"] != pr[1]["device_index"]
    else:
        assert pr[0]["device_index"] == pr[1]["device_index"] == 0


@pytest.mark.gpu
def test_bench_refuses_to_share_a_card_silently():
    """With fewer devices than ranks bench.py exits with a message unless D3D_BENCH_SHARE_GPU=1 (a scaling number measured
    with two ranks on one card would be meaningless)."""
    import subprocess

    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a one-GPU box")
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    env.pop("D3D_BENCH_SHARE_GPU", None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert res.returncode != 0 and "D3D_BENCH_SHARE_GPU" in (res.stderr + res.stdout)
