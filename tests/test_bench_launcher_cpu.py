"""bench.py --gpus N must really run N ranks (`--gpus` was parsed and never read, so a driver run with
--gpus 8 would have reported a 1-GPU number).  CPU / gloo plumbing check of both ways of starting it:
  * plain `python bench.py --gpus 2`  -> bench.py re-executes itself under torch.distributed.run with 2 ranks;
  * as the driver starts it, under torch.distributed.run;
and the refusal to run when --gpus disagrees with WORLD_SIZE."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_plain_invocation_starts_n_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and line["backend"] == "gloo"


def test_driver_style_launch():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run"]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _json_line(r.stdout)["n_gpus"] == 2


def test_gpus_must_match_world_size():
    env = _env()
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--dry-run"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout)


def test_single_rank_dry_run():
    r = subprocess.run([sys.executable, BENCH, "--dry-run", "--steps", "2", "--warmup", "0"], env=_env(), capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _json_line(r.stdout)["n_gpus"] == 1
