"""Multi-device plumbing on ONE MI355X: the RCCL gather of the C ABI at world size 1 and the in-process
`devices=` sharding with both shards on device 0 (the driver's 8-GPU run exercises N > 1; world-size-2
sharding / launcher logic is covered on CPU in tests/test_distributed_cpu.py)."""

from __future__ import annotations

import numpy as np
import pytest

from asyncflow_amd.workloads import lb_two_servers

pytestmark = pytest.mark.gpu


def test_engine_gather_through_the_c_abi_at_world_size_one():
    import torch

    from asyncflow_amd.distributed import EngineComm, gather_engine_summaries
    from asyncflow_amd.engine import Engine
    from asyncflow_amd.runner import SimulationRunner

    res = SimulationRunner(simulation_input=lb_two_servers(horizon=20), replicas=24).run()
    summ = res.summary(rps=True, hist_bins=64, hist_max=0.128, series=True)
    eng = Engine(res.plan, 0)
    comm = EngineComm(0, 1, 0)
    try:
        assert comm.count() == (1, 0)          # ncclCommCount / ncclCommUserRank through af_comm_count (ABI 6)
        tensors = {k: summ[k] for k in ("stats", "rps", "hist", "series_mean", "series_max")}
        got = gather_engine_summaries(eng, comm, tensors, n_max=32)          # padded shard: 24 -> 32 rows
        assert eng.stats().gather_ms > 0.0
        for k, t in tensors.items():
            assert got[k].shape[0] == 32 and torch.equal(got[k][:24], t), k
            assert not got[k][24:].any()
    finally:
        comm.close()
        eng.close()


def test_devices_option_shards_in_process_and_keeps_the_sweep_order():
    from asyncflow_amd.runner import SimulationRunner

    payload = lb_two_servers(horizon=20)
    seeds = np.arange(30, dtype=np.uint64) + 400
    users = np.linspace(50, 600, 30)[np.random.default_rng(0).permutation(30)]
    sweep = {"rqs_input.avg_active_users.mean": users}
    one = SimulationRunner(simulation_input=payload, seeds=seeds, sweep=sweep).run()
    two = SimulationRunner(simulation_input=payload, seeds=seeds, sweep=sweep, devices=[0, 0]).run()
    assert len(two) == 30 and len(two.shards) == 2 and abs(len(two.shards[0]) - len(two.shards[1])) <= 1
    loads = [users[ix].sum() for ix in two.index]
    assert abs(loads[0] - loads[1]) / sum(loads) < 0.05            # dealt by expected load
    assert np.array_equal(two.counts[:, :6], one.counts[:, :6]) and np.array_equal(two.seeds, seeds)
    for i in (0, 7, 29):
        assert np.array_equal(two[i].rqs_clock, one[i].rqs_clock) and np.array_equal(two[i]._samples, one[i]._samples)  # noqa: SLF001
    a, b = one.summary(rps=True), two.summary(rps=True)
    assert np.array_equal(a["stats"].cpu().numpy(), b["stats"].cpu().numpy(), equal_nan=True)
    assert np.array_equal(a["rps"].cpu().numpy(), b["rps"].cpu().numpy())
    assert two.aggregate()["mean"]["p95"] == one.aggregate()["mean"]["p95"]


# ------------------------------------------------------------------------------------------------ N ranks, N devices
def _rccl_rank(rank: int, world: int, tmp: str) -> None:
    """One rank of the N-GPU smoke test (spawned): af_comm_init_rank on ITS device from an id shared through a file,
    af_engine_gather of fabricated summary rows, the same rows through torch.distributed for comparison."""
    import os
    import time
    from pathlib import Path

    import torch
    import torch.distributed as dist

    from asyncflow_amd.distributed import EngineComm, gather_engine_summaries
    from asyncflow_amd.engine import Engine
    from asyncflow_amd.plan import lower

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    id_file = Path(tmp) / "rccl_id"

    def share(ident):
        if rank == 0:
            id_file.with_suffix(".tmp").write_bytes(ident)
            id_file.with_suffix(".tmp").rename(id_file)
            return ident
        for _ in range(600):
            if id_file.exists():
                return id_file.read_bytes()
            time.sleep(0.1)
        raise TimeoutError("rank 0 never published the RCCL id")

    n = 40 + rank                                            # ragged shards, padded to n_max by the caller
    n_max = 40 + world - 1
    g = torch.Generator().manual_seed(1000 + rank)
    tensors = {"stats": (torch.rand((n, 8), generator=g, dtype=torch.float64) + rank).to(dev),
               "rps": torch.rand((n, 30), generator=g, dtype=torch.float32).to(dev),
               "hist": torch.randint(0, 1000, (n, 64), generator=g, dtype=torch.int32).to(dev)}
    eng = Engine(lower(lb_two_servers(horizon=30)), rank)
    comm = EngineComm(rank, world, rank, share=share)
    got = gather_engine_summaries(eng, comm, tensors, n_max)
    dist.init_process_group(backend="nccl", init_method=f"file://{tmp}/pg", rank=rank, world_size=world, device_id=dev)
    for k, t in tensors.items():
        padded = torch.cat([t, torch.zeros((n_max - n, *t.shape[1:]), dtype=t.dtype, device=dev)], dim=0).contiguous()
        ref = torch.empty((world * n_max, *t.shape[1:]), dtype=t.dtype, device=dev)
        dist.all_gather_into_tensor(ref, padded)
        assert torch.equal(got[k], ref), f"rank {rank}: af_engine_gather and torch.distributed disagree on {k}"
        assert torch.equal(got[k][rank * n_max: rank * n_max + n], t)
    dist.destroy_process_group()
    comm.close()
    eng.close()


def test_engine_gather_over_every_visible_gpu(tmp_path):
    """af_comm_init_rank + af_engine_gather with one rank per GPU (the path bench.py --gpus N takes), checked against
    torch.distributed's all-gather.  Needs >= 2 GPUs: the 1-GPU boxes of this pool skip it with that reason."""
    import torch

    world = torch.cuda.device_count()
    if world < 2:
        pytest.skip(f"{world} GPU visible: the N-rank RCCL path needs at least 2 (world size 1 is covered above)")
    import torch.multiprocessing as mp

    mp.spawn(_rccl_rank, args=(world, str(tmp_path)), nprocs=world, join=True)


def test_bench_rank_rows_ride_in_the_one_gather(tmp_path):
    """bench.py's N > 1 path on ONE GPU (AF_BENCH_FORCE_DIST: a process group of size 1): the RCCL gather through the C ABI
    is taken (no fallback), and the rank's scalars come back from the extra row of the gathered stats array."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, AF_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, str(root / "bench.py"), "--scenarios", "256", "--horizon", "30", "--steps", "1", "--warmup", "0",
                          "--no-cpu-baseline", "--no-diagnostics"], env=env, capture_output=True, text=True, timeout=600, check=False)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().splitlines()[-1].startswith('{"metric"'), "the JSON line must be the LAST line of stdout"
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["gather_fallback"] is False and "af_engine_gather" in line["gather_path"] and line["gather_ms"] > 0.0
    assert line["config"]["scenarios_total"] == 256 and line["parity_spot_check"]["ok"] is True


def test_the_bench_line_of_a_launched_rank_carries_what_rccl_says(tmp_path):
    """`bench.py` as ONE launched rank (RANK / WORLD_SIZE in the environment, backend nccl): the path the driver's
    `torch.distributed.run --nproc-per-node N` takes.  The JSON line must carry the rank count RCCL itself reports for the
    communicator the gather ran on, and the per-rank kernel times."""
    import json
    import os
    import socket
    import subprocess
    import sys
    from pathlib import Path

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               AF_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    root = Path(__file__).resolve().parent.parent
    res = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--scenarios", "512",
                          "--horizon", "60", "--no-cpu-baseline", "--no-diagnostics"], env=env, capture_output=True, text=True,
                         timeout=900, check=False)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert res.returncode == 0 and lines, res.stderr[-2000:]
    line = json.loads(lines[-1])
    assert line.get("error") is None, line
    assert line["n_gpus"] == 1 and line["world_size_launcher"] == 1 and line["rccl_ranks"] == 1 and not line["gather_fallback"]
    pr = line["per_rank"]
    assert pr["rccl_comm_count"] == [1] and pr["rccl_comm_user_rank"] == [0] and pr["scenarios"] == [512]
    assert pr["flow_kernel_ms"][0] > 0.0 and abs(pr["flow_kernel_ms"][0] - line["flow_kernel_ms"]) < 1e-6
    assert line["parity_spot_check"]["ok"] is True
