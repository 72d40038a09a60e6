"""The Ticker oracle (oracle/ticker.py) against golden vectors captured from the unmodified reference
(tests/golden/ticker.npz, generator tests/golden/gen_golden_ticker.py).  CPU only."""
import os

import numpy as np
import pytest

from oracle import ticker as TK

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ticker.npz")


@pytest.fixture(scope="module")
def g():
    return np.load(GOLD)


def test_sampler_matrix_bit_exact(g):
    m = TK.open_close_to_sequence(g["tbl_open"], g["tbl_close"], g["tbl_volume"])
    assert m.shape == g["matrix"].shape == (1400, 4)
    assert np.array_equal(m, g["matrix"])
    # the property the reference asserts (sampler.py:26): row 1 is the first close
    assert m[1, 0] == g["tbl_close"][0]
    # inverse asset: log returns are the negatives
    np.testing.assert_allclose(np.diff(np.log(m[:, 1])), -np.diff(np.log(m[:, 0])), atol=1e-13)


@pytest.mark.parametrize("case", ["e0", "e1", "e2", "e3", "crash"])
def test_env_trajectory_bit_exact(g, case):
    matrix = g["crash_matrix"] if case == "crash" else g["matrix"]
    start = 0 if case == "crash" else int(g[case + "_start"])
    st, obs0 = TK.ticker_reset(matrix, [start])
    assert np.array_equal(obs0[0], g[case + "_obs0"])
    disc, cont = g[case + "_disc"], g[case + "_cont"]
    for t in range(len(g[case + "_obs"])):
        obs, rew, done = TK.ticker_step(matrix, st, disc[t][None], cont[t][None])
        assert np.array_equal(obs[0], g[case + "_obs"][t]), t
        assert st["cash"][0] == g[case + "_cash"][t] and st["assets"][0] == g[case + "_assets"][t]
        assert np.array_equal(st["qty"][0], g[case + "_qty"][t])
        assert rew[0] == g[case + "_reward"][t], t
        assert bool(done[0]) == bool(g[case + "_done"][t])
    if case == "crash":
        assert g["crash_done"][-1] and not g["crash_done"][:-1].any()


def test_batched_equals_single(g):
    matrix = g["matrix"]
    starts = [int(g["e%d_start" % e]) for e in range(4)]
    st, _ = TK.ticker_reset(matrix, starts)
    for t in range(48):
        disc = np.stack([g["e%d_disc" % e][t] for e in range(4)])
        cont = np.stack([g["e%d_cont" % e][t] for e in range(4)])
        obs, rew, done = TK.ticker_step(matrix, st, disc, cont)
        for e in range(4):
            assert np.array_equal(obs[e], g["e%d_obs" % e][t]) and rew[e] == g["e%d_reward" % e][t]


def test_state_processor_and_action_transform(g):
    assert np.array_equal(TK.ticker_process_state(g["proc_in"]), g["proc_out"])
    assert np.array_equal(TK.ticker_process_temporal_states(list(g["proc_out"][:5])), g["proc_temporal"])
    _, cont = TK.ticker_transform_raw_action([1, 2, 0], g["tra_in"])
    np.testing.assert_allclose(cont, g["tra_out"], rtol=1e-15)


def test_reference_env_properties(g):
    """tests/env_tests.py:55-80 (deplete_test, buysell_test) on the table-driven env."""
    matrix = g["matrix"]
    st, _ = TK.ticker_reset(matrix, [5])
    for _ in range(100):
        obs, rew, done = TK.ticker_step(matrix, st, [[TK.BUY, TK.BUY]], [[0.1, 0.1]])
    assert not done[0] and obs[0, 0] <= TK.MIN_CASH and (obs[0, 1:3] > 0).all()
    st, _ = TK.ticker_reset(matrix, [5])
    TK.ticker_step(matrix, st, [[TK.BUY, TK.BUY]], [[0.1, 0.1]])
    obs, rew, done = TK.ticker_step(matrix, st, [[TK.SELL, TK.SELL]], [[1.0, 1.0]])
    np.testing.assert_array_almost_equal(obs[0, 1:3], 0)
