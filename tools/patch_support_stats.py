"""How much of an agent's 5x5 dense1 patch can be non-zero?  From the agents' positions on the 84x84 grid (the engine's `positions`)
the conv1 -> conv2 -> conv3 footprints follow; prints the mean share of the 25 patch pixels inside the footprint."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "golds-rl-gym_amd"))
from goldsrl import _ffi  # noqa: E402
from goldsrl import rollout as R  # noqa: E402


def rows(ph):
    o1 = [o for o in range(20) if 4 * o <= ph <= 4 * o + 7]
    o2 = sorted({o for a in o1 for o in range(9) if 2 * o <= a <= 2 * o + 3})
    r = sorted({x for o in o2 for x in range(o - 2, o + 1) if 0 <= x <= 6})
    oy = min(max(min(o2) - 2, 0), 2)
    return len(o2), (r[0] - oy, r[-1] - oy + 1)


tab = [rows(p) for p in range(84)]
E = 4096
eng = _ffi.Engine(_ffi.ENV_SWARM, E, seed=1692)
eng.reset()
roll = R.ConvPolicyRollout(eng, 20, train=False)
for it in range(26):
    roll.run(); eng.wait()
    pos = eng.read("positions").reshape(-1, 2).astype(int)
    nr = np.array([tab[p][0] for p in pos[:, 0]]); nc = np.array([tab[p][0] for p in pos[:, 1]])
    sr = np.array([tab[p][1][1] - tab[p][1][0] for p in pos[:, 0]]); sc = np.array([tab[p][1][1] - tab[p][1][0] for p in pos[:, 1]])
    if it % 5 == 0: print("after %d rollouts: slots/agent %.2f, support %.2f x %.2f pixels = %.1f %% of the 25; positions in the outer 8 pixels: %.1f %%"
          % (it + 1, (nr * nc).mean(), sr.mean(), sc.mean(), 100 * (sr * sc).mean() / 25,
             100 * ((pos < 8) | (pos > 75)).any(axis=1).mean()))

# how many agents of an env stand on a pixel another agent of the same env already occupies (identical input image)?
pos3 = eng.read("positions").reshape(E, 10, 2).astype(int)
code = pos3[:, :, 0] * 84 + pos3[:, :, 1]
dup = sum(10 - len(set(row.tolist())) for row in code)
print("agents that duplicate another agent's pixel in their env: %.1f %%" % (100.0 * dup / (E * 10)))

# conv3 slot rows: a touched conv2 pixel u reaches conv3 outputs u - t, t in 0..2, inside 0..6 only: which share of the 9 taps is live?
def o2list(ph):
    o1 = [o for o in range(20) if 4 * o <= ph <= 4 * o + 7]
    return sorted({o for a in o1 for o in range(9) if 2 * o <= a <= 2 * o + 3})
live = tot = 0
pp = eng.read("positions").reshape(-1, 2).astype(int)
tapsy = {p: [sum(1 for t in range(3) if 0 <= u - t <= 6) for u in o2list(p)] for p in range(84)}
for ph, pw in pp[:20000]:
    for ty in tapsy[ph]:
        for tx in tapsy[pw]:
            live += ty * tx; tot += 9
print("live (slot, tap) pairs: %.1f %% of 9 per slot" % (100.0 * live / tot))

# the shared trunk: how many of the 20x20 conv1 outputs of an env see any locust / agent bin at all (the rest are relu(bias))?
lbn = eng.read("locust_bins").reshape(E, 80, 2).astype(int)
abn = eng.read("agent_bins").reshape(E, 10, 2).astype(int)
t1 = t2 = t3 = inbox = ubins = 0
o3all, o2all = [], []
for e in range(0, E, 8):
    pts = np.concatenate([lbn[e], abn[e]])
    pts = pts[pts[:, 0] != 255]
    touched = np.zeros((20, 20), bool)
    for h, w in pts:
        for oy in (h // 4, h // 4 - 1):
            for ox in (w // 4, w // 4 - 1):
                if 0 <= oy < 20 and 0 <= ox < 20 and h - 4 * oy < 8 and w - 4 * ox < 8:
                    touched[oy, ox] = True
    t1 += touched.sum()
    o2 = np.zeros((9, 9), bool)
    for oy, ox in zip(*np.nonzero(touched)):
        for q in range(9):
            for r in range(9):
                if 0 <= oy - 2 * q < 4 and 0 <= ox - 2 * r < 4:
                    o2[q, r] = True
    t2 += o2.sum()
    o3 = np.zeros((7, 7), bool)
    for q, r in zip(*np.nonzero(o2)):
        o3[max(q - 2, 0):min(q, 6) + 1, max(r - 2, 0):min(r, 6) + 1] = True
    t3 += o3.sum()
    inbox += len(pts)
    ubins += len({(h, w) for h, w in pts})
    o3all.append(o3.copy()); o2all.append(o2.copy())
ne = len(range(0, E, 8))
print("points inside the box: %.1f of 90 per env, distinct bins %.1f; conv3 outputs that see an affected conv2 output: %.1f of 49"
      % (inbox / ne, ubins / ne, t3 / ne))
print("shared trunk: conv1 outputs touched by any bin: %.1f of 400 per env; conv2 outputs that see a touched input: %.1f of 81" % (t1 / ne, t2 / ne))

# do the envs' affected sets coincide?  union over groups of 128 envs (what a 128-row GEMM tile would have to cover)
o3a, o2a = np.array(o3all), np.array(o2all)
for g in (16, 128, len(o3a)):
    u3 = [o3a[i:i + g].any(axis=0).sum() for i in range(0, len(o3a) - g + 1, g)]
    u2 = [o2a[i:i + g].any(axis=0).sum() for i in range(0, len(o2a) - g + 1, g)]
    print("union of the affected sets over %d envs: conv3 outputs %.1f of 49, conv2 outputs %.1f of 81" % (g, np.mean(u3), np.mean(u2)))
print("how often each conv3 output is affected (7x7, per cent of envs):")
print(np.round(100 * o3a.mean(axis=0)).astype(int))
