"""End-to-end training sanity run: PAAC on Swarm-v0 with the conv policy; logs loss / critic loss / mean reward per update and an
eval episode on Swarm-eval-v0 (seed 192) every `eval_every` updates."""
import json, sys, time
import numpy as np
sys.path.insert(0, 'golds-rl-gym_amd')
from goldsrl import _ffi, envs, rollout
from goldsrl.agents.paac import policy_monitor as PM
from goldsrl.agents.state_processors import SwarmStateProcessor

E, T, U = int(sys.argv[1]), 20, int(sys.argv[2])
lr = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-4
eval_every = int(sys.argv[4]) if len(sys.argv) > 4 else 0
eng = _ffi.Engine(_ffi.ENV_SWARM, E, seed=1692)
eng.reset()
eng.episodes_enable(capacity=T * E)       # R6: episode reward sums kept on the device (paac.py:331-349)
r = rollout.ConvPolicyRollout(eng, T, train=True, lr=lr)


class _Global(object):      # the estimator-object surface the monitor needs
    def get_flat_params(self):
        return r.net.get_params()


conf = dict(name='eval', num_actions=2, clip_norm=40.0, clip_norm_type='global', device='/gpu:0', entropy_regularisation_strength=0.02,
            scale=1000.0, height=84, width=84, channels=3)
mon = PM.SwarmPolicyMonitor(envs.make("Swarm-eval-v0"), _Global(), SwarmStateProcessor(grid_size=84), None, network_conf=conf) if eval_every else None
if mon is not None:
    mon.actions_path = "gpurun_out/swarm-eval.json"
rows, evals = [], []
t0 = time.time()
for u in range(U):
    if mon is not None and u % eval_every == 0:
        tot, n, _ = mon.eval_once()       # Swarm-eval-v0: the reference's seed-192 episode, action noise from its re-seeded stream
        evals.append({"update": u, "env_steps": u * E * T, "eval_total_reward": tot, "episode_length": n})
        print(json.dumps(evals[-1]), flush=True)
    r.run()
    eng.wait()
    s = dict(r.last_stats)
    s.update(update=u, elapsed_s=time.time() - t0)
    eps = eng.episodes_read()
    if len(eps):
        s["train_episodes_finished"] = int(len(eps))
        s["train_episode_reward_mean"] = float(eps["total_reward"].mean())       # the mean of the update's `rl/reward` points
    if u % 20 == 0:
        s["mean_reward"] = float(r.net.read_rollout("rewards", (T, E * 10)).mean())
    rows.append(s)
    if u % 100 == 0 or u == U - 1:
        print(json.dumps(s), flush=True)
json.dump({"envs": E, "T": T, "updates": U, "lr": lr, "rows": [x for i, x in enumerate(rows) if i % 10 == 0 or i == len(rows) - 1], "evals": evals,
           "wall_s": time.time() - t0}, open("gpurun_out/learning_curve.json", "w"))
assert all(np.isfinite(list(x.values())).all() for x in rows)
