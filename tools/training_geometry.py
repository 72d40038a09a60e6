"""Where do the agents go when the policy TRAINS, and what does that do to the update time?  (round 4)
The bench's `value` is measured on a random-init policy whose agents sit on the rim of the 84x84 observation grid; its
`interior_policy` leg injects agents into the interior.  This run trains (PAAC, conv policy, lr 1e-4) and logs, per 50 updates: the
wall-clock ms per update, the share of agents whose observation bins lie inside 8..75 on both axes, the mean bins, and the mean
reward of the finished training episodes.   usage: python tools/training_geometry.py [envs=8192] [updates=800]"""
import json, sys, time
import numpy as np
sys.path.insert(0, 'golds-rl-gym_amd')
from goldsrl import _ffi, rollout

E = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
U = int(sys.argv[2]) if len(sys.argv) > 2 else 800
T = 20
eng = _ffi.Engine(_ffi.ENV_SWARM, E, seed=1692)
eng.reset()
eng.episodes_enable(capacity=T * E)
r = rollout.ConvPolicyRollout(eng, T, train=True, lr=1e-4)
rows, t_blk, rew = [], time.time(), []
for u in range(U + 1):
    if u % 50 == 0:
        eng.wait()
        pos = eng.read("positions").astype(int)
        ab = eng.read("agent_bins").astype(int)
        inside = ((pos >= 8) & (pos <= 75)).all(axis=2)
        row = {"update": u, "env_steps": u * E * T, "ms_per_update": (time.time() - t_blk) * 1e3 / 50 if u else None,
               "agents_inside_bins_8_75": float(inside.mean()), "agents_outside_the_box": float((ab[:, :, 0] == 255).mean()),
               "mean_bin_x": float(pos[:, :, 0].mean()), "mean_bin_y": float(pos[:, :, 1].mean()),
               "train_episode_reward_mean": float(np.mean(rew)) if rew else None}
        rows.append(row)
        print(json.dumps(row), flush=True)
        rew, t_blk = [], time.time()
    if u == U:
        break
    r.run()
    eps = eng.episodes_read()
    if len(eps):
        rew.append(float(eps["total_reward"].mean()))
json.dump({"envs": E, "T": T, "updates": U, "lr": 1e-4, "rows": rows}, open("gpurun_out/training_geometry.json", "w"), indent=1)
