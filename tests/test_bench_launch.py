"""bench.py launch contract on a CPU-only machine: `python bench.py --gpus N` must start its own N ranks, and the same
file must still run as a rank under `python -m torch.distributed.run`.  --dry-run = rendezvous (gloo) + the barrier bracket, no GPU."""
import json
import os
import socket
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _json_lines(out):
    return [json.loads(l) for l in out.splitlines() if l.startswith("{")]


def test_plain_python_bench_gpus_2_launches_its_own_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "7", "--warmup", "1", "--dry-run"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout          # ONE line, from rank 0
    rec = lines[0]
    assert rec["n_gpus"] == 2 and rec["steps"] == 7 and rec["warmup"] == 1 and rec["dry_run"] is True
    assert rec["config"]["ranks_seen_by_process_group"] == 2
    assert rec["config"]["workload"] == "c4"  # N > 1 runs BASELINE configs[3]


def test_bench_as_a_rank_under_torch_distributed_run():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "0", "--dry-run"],
                       capture_output=True, text=True, timeout=300, cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["config"]["ranks_seen_by_process_group"] == 2


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True, timeout=120, env=env, cwd=REPO)
    assert r.returncode != 0 and "nproc-per-node" in (r.stderr + r.stdout)


def test_dry_run_at_world_size_8_shards_and_wires():
    """the first real 8-GPU run is the driver's: the launch path, the shard arithmetic and the per-wire link arithmetic of `--gpus 8` on a
    CPU-only machine"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5", "--dry-run"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    rec = lines[0]
    assert rec["n_gpus"] == 8 and rec["config"]["ranks_seen_by_process_group"] == 8 and rec["config"]["workload"] == "c4"
    assert rec["wire"] == "f32"                            # the headline wire is the bit-exact one (the reference's float observations); bf16 / q8 are labelled lossy variants
    shards = sorted(rec["config"]["shards"], key=lambda d: d["rank"])
    assert [d["rank"] for d in shards] == list(range(8))
    assert [d["envs"] for d in shards] == [[512 * k, 512 * (k + 1)] for k in range(8)]      # BASELINE configs[3]: 4096 envs, contiguous shards
    assert all(d["env_id_offset"] == d["envs"][0] and d["drones"] == 512 * 32 for d in shards)
    pw = rec["config"]["exchange_per_wire"]
    assert [pw[w]["row_bytes"] for w in ("f32", "bf16", "q8")] == [216, 108, 72] and pw["f32"]["lossy"] is False and pw["q8"]["lossy"] is True
    assert abs(pw["f32"]["predicted_link_us_per_step_at_77_GB_per_s"] - 1e6 * 16384 * 216 / 77e9) < 1e-6
