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
