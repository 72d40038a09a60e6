"""End-to-end training sanity run: PAAC on Swarm-v0 with the conv policy; logs loss / critic loss / mean reward per update."""
import json, sys, time
import numpy as np
sys.path.insert(0, 'golds-rl-gym_amd')
from goldsrl import _ffi, rollout
E, T, U = int(sys.argv[1]), 20, int(sys.argv[2])
eng = _ffi.Engine(_ffi.ENV_SWARM, E, seed=1692)
eng.reset()
r = rollout.ConvPolicyRollout(eng, T, train=True, lr=float(sys.argv[3]) if len(sys.argv) > 3 else 1e-4)
rows = []
t0 = time.time()
for u in range(U):
    r.run()
    eng.wait()
    rew = r.net.read_rollout("rewards", (T, E * 10))
    s = dict(r.last_stats)
    s.update(update=u, mean_reward=float(rew.mean()), elapsed_s=time.time() - t0)
    rows.append(s)
    if u % 10 == 0 or u == U - 1:
        print(json.dumps(s), flush=True)
json.dump({"envs": E, "T": T, "updates": U, "rows": rows}, open("gpurun_out/learning_curve.json", "w"))
assert all(np.isfinite(list(x.values())).all() for x in rows)
