#!/usr/bin/env python
"""Scheduler / code-generation sweep of the config-specialised team kernels: the same sources under different machine-scheduler settings.
Why: round 5's own-branch counter experiment showed that a different schedule of the same ~3400 instructions of wave 0 moves
single phases of a step by hundreds of clocks; the instruction ORDER is the only thing these flags change (results are bit-identical).
  python tools/sched_sweep.py build            (build container: compile every variant's objects for c2 / c3 / c4 into spec_cache/)
  python tools/sched_sweep.py run [reps]       (GPU box: us per step of every variant, interleaved)  -> gpurun_out/<tag>_sched_sweep.txt
QS_SPEC_TEAM_FLAGS replaces the team objects' default scheduler flag, QS_SPEC_EXTRA_FLAGS adds to the command line (both are part of the
cache key)."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

MAXILP = "-mllvm -amdgpu-sched-strategy=max-ilp"
NOPOST = MAXILP + " -mllvm -enable-post-misched=0"
SWEEPS = {
    "1": [   # name, QS_SPEC_TEAM_FLAGS (None = the library's default), QS_SPEC_EXTRA_FLAGS
        ("library default", None, ""),   # (sweeps 1 and 2 ran when that was max-ilp for every team object)
        ("max-occupancy (compiler default)", "", ""),
        ("max-memory-clause", "-mllvm -amdgpu-sched-strategy=max-memory-clause", ""),
        ("max-ilp, no post-RA scheduler", NOPOST, ""),
        ("max-ilp, bottom-up only", MAXILP + " -mllvm -misched-bottomup", ""),
        ("max-ilp, top-down only", MAXILP + " -mllvm -misched-topdown", ""),
        ("max-ilp + own-branch counter", MAXILP, "-DQS_AB_OWN_CTR"),
        ("max-occupancy + own-branch counter", "", "-DQS_AB_OWN_CTR"),
        ("max-ilp, no post-RA + own-branch counter", NOPOST, "-DQS_AB_OWN_CTR"),
        ("no machine scheduler (source order)", "-mllvm -enable-misched=0", ""),
    ],
    "2": [   # around the winner of sweep 1 (max-ilp without the post-RA scheduler)
        ("library default", None, ""),   # (sweeps 1 and 2 ran when that was max-ilp for every team object)
        ("max-ilp, no post-RA", NOPOST, ""),
        ("no post-RA, no clustered low-occupancy stage", NOPOST + " -mllvm -amdgpu-disable-clustered-low-occupancy-reschedule", ""),
        ("no post-RA, no unclustered high-RP stage", NOPOST + " -mllvm -amdgpu-disable-unclustered-high-rp-reschedule", ""),
        ("no post-RA, no memop clustering", NOPOST + " -mllvm -misched-cluster=0", ""),
        ("no post-RA, AMDGPU RP trackers", NOPOST + " -mllvm -amdgpu-use-amdgpu-trackers", ""),
        ("no post-RA, pre-RA top-down", NOPOST + " -mllvm -misched-prera-direction=topdown", ""),
        ("no post-RA, pre-RA bottom-up", NOPOST + " -mllvm -misched-prera-direction=bottomup", ""),
        ("no post-RA, no post-RA sinking", NOPOST + " -mllvm -disable-postra-machine-sink", ""),
        ("max-ilp, AMDGPU RP trackers", MAXILP + " -mllvm -amdgpu-use-amdgpu-trackers", ""),
        ("max-ilp, no memop clustering", MAXILP + " -mllvm -misched-cluster=0", ""),
        ("max-ilp, no unclustered high-RP stage", MAXILP + " -mllvm -amdgpu-disable-unclustered-high-rp-reschedule", ""),
    ],
    "3": [   # source-level switches and optimisation level on the 8-wave default (c2 / c3 are the 8-wave shapes)
        ("library default", None, ""),
        ("state arrays loaded in first-use order", None, "-DQS_LD_ORDER=1"),
        ("filters first, then the old order", None, "-DQS_LD_ORDER=2"),
        ("own-branch counter", None, "-DQS_AB_OWN_CTR"),
        ("own-branch counter + first-use order", None, "-DQS_AB_OWN_CTR -DQS_LD_ORDER=1"),
        ("row skipping", None, "-DQS_SKIP_ROWS=1"),
        ("-O2", None, "-O2"),
        ("occupancy bias 0", NOPOST + " -mllvm -amdgpu-schedule-metric-bias=0", ""),
    ],
    "4": [   # the single-wave throughput objects (SWEEP_TEAM=0 SWEEP_ENVS=131072): they are built without scheduler flags
        ("library default", None, ""),
        ("no post-RA scheduler", None, "-mllvm -enable-post-misched=0"),
        ("max-ilp, no post-RA scheduler", None, NOPOST),
        ("no memop clustering", None, "-mllvm -misched-cluster=0"),
        ("occupancy bias 0", None, "-mllvm -amdgpu-schedule-metric-bias=0"),
        ("AMDGPU RP trackers", None, "-mllvm -amdgpu-use-amdgpu-trackers"),
    ],
    "5": [   # combinations of what moved sweep 4
        ("library default", None, ""),
        ("AMDGPU RP trackers", None, "-mllvm -amdgpu-use-amdgpu-trackers"),
        ("RP trackers, no memop clustering", None, "-mllvm -amdgpu-use-amdgpu-trackers -mllvm -misched-cluster=0"),
        ("RP trackers, occupancy bias 0", None, "-mllvm -amdgpu-use-amdgpu-trackers -mllvm -amdgpu-schedule-metric-bias=0"),
        ("occupancy bias 0, no memop clustering", None, "-mllvm -amdgpu-schedule-metric-bias=0 -mllvm -misched-cluster=0"),
        ("RP trackers, bias 0, no clustering", None, "-mllvm -amdgpu-use-amdgpu-trackers -mllvm -amdgpu-schedule-metric-bias=0 -mllvm -misched-cluster=0"),
    ],
}
VARIANTS = SWEEPS[os.environ.get("SWEEP", "1")]
TEAM = os.environ.get("SWEEP_TEAM")      # "0": the single-wave objects (what batches that fill the chip run)
ENVS = os.environ.get("SWEEP_ENVS")      # envs per GPU of the measured shape (default: the workload's own)
WORKLOADS = tuple(os.environ.get("SWEEP_WORKLOADS", "c2,c3,c4").split(","))


def env_of(team_flags, extra):
    env = dict(os.environ)
    env.pop("QS_SPEC_TEAM_FLAGS", None)
    env.pop("QS_SPEC_EXTRA_FLAGS", None)
    if team_flags is not None:
        env["QS_SPEC_TEAM_FLAGS"] = team_flags
    if extra:
        env["QS_SPEC_EXTRA_FLAGS"] = extra
    return env


def build():
    code = ("import sys; sys.path.insert(0, %r)\nimport __graft_entry__ as g, bench\nfrom concurrent.futures import ProcessPoolExecutor\n"
            "jobs = [(dict(bench.WORKLOADS[w]['kw'], num_envs=bench.WORKLOADS[w]['num_envs'], write_rew_info=False), 'f32') + %r for w in %r]\n"
            "with ProcessPoolExecutor(max_workers=4) as ex: print(len([p for p in ex.map(g._spec_one, jobs) if p]))\n" % (REPO, (int(TEAM),) if TEAM else (), WORKLOADS))
    for name, tf, xf in VARIANTS:
        out = subprocess.run([sys.executable, "-c", code], env=env_of(tf, xf), capture_output=True, text=True)
        print(f"{name}: {out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:]} objects")


def run(reps, tag):
    path = os.path.join(REPO, "gpurun_out", f"{tag}_sched_sweep.txt")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "a") as f:
        for rep in range(reps):
            for name, tf, xf in VARIANTS:
                row = []
                for wl in WORKLOADS:
                    envs = ENVS.split(",")[WORKLOADS.index(wl)] if ENVS and "," in ENVS else ENVS
                    shape = ["--envs-per-gpu", envs, "--steps", "300", "--warmup", "50", "--prewarm", "200", "--rollout-steps", "0"] if ENVS else ["--steps", "3000"]
                    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--workload", wl, "--cpu-seconds", "0", "--no-f64",
                                          "--no-closed-loop", "--no-variants", "--no-c5-train"] + shape, env=env_of(tf, xf), capture_output=True, text=True, timeout=600)
                    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
                    d = json.loads(lines[-1]) if lines else None
                    row.append(f"{wl} {1e3 * d['ms_per_step']:.3f}" + ("" if d["roofline"].get("specialized") else " (GENERIC)") if d else f"{wl} failed")
                line = f"rep {rep} | {name:45s} | " + " | ".join(row)
                print(line, flush=True)
                f.write(line + "\n")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    else:
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 2, os.environ.get("SWEEP_TAG", "sweep"))
