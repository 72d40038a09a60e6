"""PAAC on Solow-v0 with FlatPolicyVNetwork (BASELINE config 2): eval episodes through SolowPolicyMonitor while training."""
import json, sys, time
import numpy as np
sys.path.insert(0, 'golds-rl-gym_amd')
from goldsrl import _ffi
from goldsrl.envs.fed_env import SolowEnv
from goldsrl.agents.paac import policy_monitor as PM
from goldsrl.agents.paac.policy_v_network import FlatPolicyVNetwork
from goldsrl.agents.state_processors import SolowStateProcessor

E, T, U = 4096, 20, int(sys.argv[1]) if len(sys.argv) > 1 else 2000
lr = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-4
conf = dict(name='local_learning', num_actions=1, clip_norm=40.0, clip_norm_type='global', device='/gpu:0', scale=100.0,
            static_size=2, temporal_size=2, entropy_regularisation_strength=0.02, static_hidden_size=32, rnn_hidden_size=32)
eng = _ffi.Engine(_ffi.ENV_SOLOW, E, seed=1692, max_episode_steps=1024)
eng.reset()
gnet = FlatPolicyVNetwork(conf).bind(eng, max_samples=E * T)
mon = PM.SolowPolicyMonitor(SolowEnv(p=1, q=1, T=256, seed=1692, max_episode_steps=256), gnet, SolowStateProcessor(), None, network_conf=conf)
evals, t0 = [], time.time()
for u in range(U):
    if u % 200 == 0:
        np.random.seed(7)
        tot, n, rew = mon.eval_once()
        evals.append({"update": u, "env_steps": u * E * T, "eval_total_reward": tot, "episode_length": n, "elapsed_s": time.time() - t0})
        print(json.dumps(evals[-1]), flush=True)
    gnet.net.rollout(T)
    st = gnet.net.train_rollout(lr)
assert np.isfinite(list(st.values())).all()
json.dump({"envs": E, "T": T, "updates": U, "lr": lr, "evals": evals, "last_stats": st}, open("gpurun_out/learning_curve_solow.json", "w"))
