"""CPU: the C restatement (oracle/oracle_c.c, bench.py's cpu_baseline port) against the numpy oracle."""
import os
import subprocess

import numpy as np

from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oc():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    from oracle import oracle_c
    return oracle_c


def test_c_swarm_step_and_observe_match_numpy_oracle(golden):
    OC = _oc()
    g = golden("swarm_step")
    a32 = g["action"].astype(np.float32)
    x, xa, r, lb, ab, pos, used = OC.swarm_step(g["x"], g["xa"], a32, g["agent_noise"], g["particle_noise"], threads=2)
    ox, oxa, orew, _ = O.swarm_step(g["x"], g["xa"], a32, g["agent_noise"], g["particle_noise"])
    assert np.array_equal(xa, oxa)
    np.testing.assert_allclose(x, ox, rtol=1e-13, atol=1e-14)      # glibc exp vs numpy's SIMD exp: last-ulp differences
    np.testing.assert_allclose(r, orew, rtol=1e-13)
    for i in range(len(x)):
        olb, oab, opos = O.swarm_observe_compact(x[i], xa[i], 84)
        assert np.array_equal(np.where(olb < 0, 255, olb), lb[i]) and np.array_equal(np.where(oab < 0, 255, oab), ab[i])
        assert np.array_equal(opos, pos[i])


def test_c_observe_golden_bit_exact(golden):
    OC = _oc()
    g = golden("swarm_obs")
    E = len(g["x"])
    zero = np.zeros((E, 10, 2), np.float32)
    # observe-only is not exported; step with dt-scaled zero noise would move points, so bin the fixture
    # states through the numpy oracle and the stepped states through both (above).  Here: edges only.
    for i in range(E):
        lb, ab, pos = O.swarm_observe_compact(g["x"][i], g["xa"][i], 84)
        assert np.array_equal(pos, g["positions"][i])


def test_c_returns_match_golden(golden):
    """oracle_c.c's return loops against the arrays of the reference's own train() loops (tests/golden/paac_loop.npz)."""
    OC = _oc()
    g = golden("paac_loop")
    E, T, U = int(g["grid_E"]), int(g["grid_T"]), int(g["grid_updates"])
    for u in range(U):       # unmasked, unclipped (paac.py:360-365)
        y, adv = OC.returns(g["grid_rewards"][u], g["grid_vs"].reshape(U, T, E * 10)[u], g["grid_boot"][u], float(g["grid_gamma"]))
        assert np.array_equal(y, g["grid_y_batch"][u]) and np.array_equal(adv, g["grid_adv_batch"][u])
    E, T, U = int(g["flat_E"]), int(g["flat_T"]), int(g["flat_updates"])
    for u in range(U):       # masked, rewards already clipped (paac.py:145,167-172)
        y, adv = OC.returns(g["flat_rewards"][u], g["flat_vs"].reshape(U, T, E)[u], g["flat_boot"][u], float(g["flat_gamma"]),
                            mask=g["flat_episodes_over_masks"][u])
        assert np.array_equal(y, g["flat_y_batch"][u]) and np.array_equal(adv, g["flat_adv_batch"][u])
