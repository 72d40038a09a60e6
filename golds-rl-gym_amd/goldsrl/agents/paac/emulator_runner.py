"""EmulatorRunner mirror (reference fed_gym/agents/paac/emulator_runner.py).  There are no worker
processes: the per-env loop body (step / auto-reset / process_state / history / reward / done) runs in
the device kernels; these classes keep the slot indices and the two statics the learner calls."""
import numpy as np

from ... import _ffi

_engines = {}


def _engine(kind):
    if kind not in _engines:
        _engines[kind] = _ffi.Engine(kind, 1, max_episode_steps=0)
    return _engines[kind]


class EmulatorRunner(object):
    STATE_IDX = 0
    HISTORY_IDX = 1
    REWARD_IDX = 2
    DONE_IDX = 3
    ACTIONS_IDX = 4
    ENV_KIND = None

    @staticmethod
    def transform_actions_for_env(actions):
        return actions


class SolowRunner(EmulatorRunner):
    ENV_KIND = _ffi.ENV_SOLOW

    @staticmethod
    def transform_actions_for_env(actions):
        """sigmoid (emulator_runner.py:77-79)"""
        a = np.asarray(actions)
        return _engine(_ffi.ENV_SOLOW).transform_actions(a.reshape(-1, 1)).reshape(a.shape)


class SwarmRunner(EmulatorRunner):
    STATE_IDX = 0
    HISTORY_IDX = 1
    AGENT_POSITIONS_IDX = 2
    REWARD_IDX = 3
    DONE_IDX = 4
    ACTIONS_IDX = 5
    MAX_MOVE_NORM = 1
    ENV_KIND = _ffi.ENV_SWARM

    @staticmethod
    def get_local_states(state, agent_positions):
        """(G,G,2) grid + positions -> list of 10 (G,G,3) arrays with the one-hot third channel
        (emulator_runner.py:98-111).  Host-side compat helper; the rollout never builds these."""
        new_states = []
        for agent in range(len(agent_positions)):
            grid_position = np.zeros_like(state[:, :, 0])
            grid_position[agent_positions[agent][0], agent_positions[agent][1]] = 1.
            new_states.append(np.concatenate([state, grid_position[:, :, None]], axis=-1))
        return new_states

    @staticmethod
    def transform_actions_for_env(actions):
        """rows with norm >= 1 are normalised (emulator_runner.py:113-118)"""
        a = np.asarray(actions)
        return _engine(_ffi.ENV_SWARM).transform_actions(a.reshape(-1, 2)).reshape(a.shape)
