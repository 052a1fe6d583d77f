#!/usr/bin/env python
"""BASELINE.json configs[4] ("C5"): the reference's published training recipe (train_local.sh:1-17) with the HIP stepper as the
environment, through Sample Factory's APPO - when Sample Factory is installed.  Prints one JSON line: either the frames per
second SF reports for `--train_for_env_steps` steps, or the exact import error (this image has no sample_factory and no network).
  python tools/train_c5.py [--steps 200000] [--num_envs 1024]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200000)
    ap.add_argument("--num_envs", type=int, default=1024)
    args = ap.parse_args()
    try:
        import sample_factory  # noqa: F401
        from sample_factory.train import run_rl
    except Exception as exc:   # noqa: BLE001 - the exact error is the result
        print(json.dumps({"c5": "not run", "error": f"{type(exc).__name__}: {exc}"}))
        return 0
    from quad_swarm_rl_amd import sf_env
    sf_env.register_swarm_components()
    argv = ["--env=quadrotor_multi", f"--train_for_env_steps={args.steps}", "--algo=APPO", "--use_rnn=False", "--num_workers=1",
            "--num_envs_per_worker=1", "--learning_rate=0.0001", "--ppo_clip_value=5.0", "--recurrence=1", "--nonlinearity=tanh",
            "--actor_critic_share_weights=False", "--policy_initialization=xavier_uniform", "--adaptive_stddev=False", "--with_vtrace=False",
            "--max_policy_lag=100000000", "--rnn_size=256", "--gae_lambda=1.00", "--max_grad_norm=5.0", "--exploration_loss_coeff=0.0",
            "--rollout=128", "--batch_size=1024", "--with_pbt=False", "--normalize_input=False", "--normalize_returns=False", "--reward_clip=10",
            "--quads_use_numba=True", "--anneal_collision_steps=300000000", "--replay_buffer_sample_prob=0.75", "--quads_mode=mix",
            "--quads_episode_duration=15.0", "--quads_obs_repr=xyz_vxyz_R_omega", "--quads_neighbor_hidden_size=256",
            "--quads_neighbor_obs_type=pos_vel", "--quads_collision_hitbox_radius=2.0", "--quads_collision_falloff_radius=4.0",
            "--quads_collision_reward=5.0", "--quads_collision_smooth_max_penalty=10.0", "--quads_neighbor_encoder_type=attention",
            "--quads_neighbor_visible_num=6", "--quads_use_obstacles=False", "--quads_use_downwash=True", f"--quads_num_envs={args.num_envs}",
            "--experiment=c5_hip_stepper", "--serial_mode=True", "--async_rl=False", "--batched_sampling=True"]
    cfg = sf_env.parse_swarm_cfg(argv=argv)
    t0 = time.time()
    status = run_rl(cfg)
    dt = time.time() - t0
    print(json.dumps({"c5": "ran", "status": int(status), "env_steps": args.steps, "seconds": dt, "fps": args.steps / dt}))
    return 0


if __name__ == "__main__":
    sys.exit(main())
