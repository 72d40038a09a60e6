"""StateProcessor mirror (reference fed_gym/agents/state_processors.py).  The arithmetic runs in the
device kernels (swarm observe stage / flat-env obs); these classes only keep the reference's call shape."""
import numpy as np

from .. import _ffi


class StateProcessor(object):
    def __init__(self, scales):
        self.scales = scales

    def process_temporal_states(self, history):
        raise NotImplementedError


class SwarmStateProcessor(StateProcessor):
    """process_state([x, xa]) -> (grid, grid, 2) float64 density, and .positions (10,2) uint8
    (state_processors.py:15-42).  Also exposes the compact form the kernels use."""

    def __init__(self, scales=1., grid_size=20, device_id=0):
        super().__init__(scales)
        self.grid_size = grid_size
        self.positions = None
        self.WIDTH = 3.
        self.HEIGHT = 3.
        self._eng = _ffi.Engine(_ffi.ENV_SWARM, 1, device_id=device_id, grid_size=grid_size, max_episode_steps=0)

    def process_state_compact(self, state):
        self._eng.set_state("SWARM_X", np.asarray(state[0], np.float64)[None])
        self._eng.set_state("SWARM_XA", np.asarray(state[1], np.float64)[None])
        self._eng.observe()
        self.positions = self._eng.read("positions")[0]
        return self._eng.read("locust_bins")[0], self._eng.read("agent_bins")[0], self.positions

    def process_state(self, state):
        lb, ab, _ = self.process_state_compact(state)
        G = self.grid_size
        counts = np.zeros((G, G, 2), np.int64)
        for bx, by in lb[lb[:, 0] != 255]:
            counts[bx, by, 0] += 1
        for bx, by in ab[ab[:, 0] != 255]:
            counts[bx, by, 1] += 1
        # x_grid / len(state[0]), xa_grid / len(state[1])
        return np.stack([counts[:, :, 0] / float(len(state[0])), counts[:, :, 1] / float(len(state[1]))], axis=-1)


class SolowStateProcessor(StateProcessor):
    def __init__(self):
        super(SolowStateProcessor, self).__init__(np.array([100., 1.]))

    def process_state(self, state):
        return np.asarray(state) / self.scales

    def process_temporal_states(self, history):
        if len(history) == 1:
            return np.array(history[0]).reshape((1, -1))
        return np.array(history)


class TickerTraderStateProcessor(StateProcessor):
    """state_processors.py:45-66.  The device env already emits this vector as its `obs` output (csrc/ticker.hip,
    float32); this host form serves single states handed over by the gym-style TickerEnv."""

    def __init__(self, n_assets):
        super(TickerTraderStateProcessor, self).__init__(None)
        self.n_assets = n_assets

    def process_state(self, raw_state):
        raw = np.asarray(raw_state, dtype=np.float64)
        n = self.n_assets
        cash, held, prices, volumes = raw[0], raw[1:1 + n], raw[1 + n:1 + 2 * n], raw[1 + 2 * n:]
        return np.concatenate([[np.log(cash + 1e-4)], np.log(held + 1), np.log(prices), volumes])

    def process_temporal_states(self, history):
        return np.vstack(history)[:, 1 + self.n_assets:]
