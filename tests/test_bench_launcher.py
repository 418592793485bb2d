"""bench.py's own launcher (`--gpus N` without torch.distributed.run), on CPU:

  * N child processes are started, rendezvous over gloo on 127.0.0.1 and exchange fixed-size records
    (kzg_snark_amd.sharding.all_gather_bytes) -- `--rehearse-launch`, no GPU involved;
  * asking for more GPUs than are visible fails loudly instead of printing a smaller job under the
    requested label;
  * a launcher whose WORLD_SIZE disagrees with --gpus is refused."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(args, env=None, timeout=300):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH, *args], capture_output=True, text=True, timeout=timeout, env=e)


def test_launcher_spawns_one_process_per_rank_and_they_exchange():
    p = run(["--gpus", "2", "--rehearse-launch"])
    assert p.returncode == 0, p.stderr
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out == {"launcher": "ok", "world": 2, "n_gpus": 2}


def test_more_gpus_than_visible_is_refused():
    import torch
    have = torch.cuda.device_count()
    p = run(["--gpus", str(have + 2)])
    assert p.returncode != 0
    assert "visible" in p.stderr and not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_world_size_must_match_gpus():
    p = run(["--gpus", "4", "--rehearse-launch"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "disagree" in p.stderr
