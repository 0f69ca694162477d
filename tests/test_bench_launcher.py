"""bench.py's multi-rank launch path on CPU: `python bench.py --gpus 2` without a torchrun environment must re-execute
itself under torch.distributed.run (one rank per GPU), and rank 0 must print exactly one JSON line.  The step is the
--stub one (a trivial CPU loop over gloo): what is tested is the launcher, the rendezvous on 127.0.0.1, the
barrier-bracketed timing and the max-over-ranks -- everything of the N > 1 path that does not need a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_spawns_its_own_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--stub"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["value"] > 0


def test_four_ranks_report_four_figures_and_disjoint_shards():
    """the rank-count-dependent bookkeeping of the line beyond world size 2 -- N per-rank figures, the
    max-over-ranks `value` no larger than the sum of them, contiguous disjoint shards of the job's N * batch items -- so that the
    first real SCALE_*.json (rccl_ranks == N, N figures, scaling "weak") can be checked at a glance"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "5", "--warmup", "1",
                        "--batch", "7", "--stub"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 4 and out["rccl_ranks"] == 4 and out["scaling"] == "weak"
    hp = out["hot_path"]
    assert len(hp["per_rank_shapes_per_s"]) == 4 and all(v > 0 for v in hp["per_rank_shapes_per_s"])
    assert out["value"] <= sum(hp["per_rank_shapes_per_s"]) * (1 + 1e-9)          # whole job at the slowest rank's pace
    assert abs(out["value"] - 4 * 7 * 5 / (out["ms_per_step"] * 5 / 1e3)) < 1e-6 * out["value"]
    assert hp["shards"] == [[0, 7], [7, 14], [14, 21], [21, 28]]


def _train_worker(rank, world, port, ret):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch
    torch.set_num_threads(4)
    torch.cuda.synchronize = lambda *a, **k: None                       # the section's device fences: no device here
    torch.cuda.empty_cache = lambda *a, **k: None
    import bench
    from genre_shapehd_amd import dist_utils
    dist = dist_utils.init_from_env("gloo")
    res = bench.train_bench(torch.device("cpu"), dist, dist_utils, world, rank, 1, which=("shapehd",))
    ret[rank] = res
    dist.barrier()
    dist.destroy_process_group()


def test_train_section_runs_on_every_rank_under_ddp():
    """bench.py's `train` section at N = 2 (gloo, CPU): every rank builds the full-width ShapeHD step under
    DistributedDataParallel, the timing is barrier-bracketed and the maximum over ranks -- both ranks report the SAME
    whole-job figure -- and a config that is not selected leaves no key"""
    import torch.multiprocessing as mp
    world, port = 2, 32600 + (os.getpid() % 2000)
    ret = mp.Manager().dict()
    mp.spawn(_train_worker, args=(world, port, ret), nprocs=world, join=True)
    a, b = ret[0], ret[1]
    assert set(a) == {"what", "steps", "shapehd_b8"} and "error" not in a["shapehd_b8"], a
    assert a["shapehd_b8"]["batch_per_gpu"] == 8 and a["shapehd_b8"]["samples_per_s"] > 0
    assert a["shapehd_b8"]["ms_per_step"] == b["shapehd_b8"]["ms_per_step"]             # max over ranks: one number
    assert abs(a["shapehd_b8"]["samples_per_s"] * a["shapehd_b8"]["ms_per_step"] / 1e3 - world * 8) < 1e-6


def _failing_worker(rank, world, port, ret):
    os.environ["GENRE_BENCH_INJECT_FAILURE"] = "1:shapehd_b8"           # rank 1 cannot build the config
    _train_worker(rank, world, port, ret)


def test_a_failure_on_one_rank_drops_the_config_on_every_rank():
    """the train section runs on every rank before rank 0 prints the bench line; a rank that fails must not leave
    the others in a barrier / all-reduce.  Rank 1's build raises: both ranks agree (MIN all-reduce), both report the config
    as failed, and both reach the end of the section."""
    import torch.multiprocessing as mp
    world, port = 2, 34600 + (os.getpid() % 2000)
    ret = mp.Manager().dict()
    mp.spawn(_failing_worker, args=(world, port, ret), nprocs=world, join=True)
    assert "error" in ret[0]["shapehd_b8"] and "error" in ret[1]["shapehd_b8"], dict(ret)
    assert "injected" in ret[1]["shapehd_b8"]["error"] and "another rank" in ret[0]["shapehd_b8"]["error"]
