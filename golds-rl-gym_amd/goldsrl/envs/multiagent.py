"""SwarmEnv facade (reference fed_gym/envs/multiagent.py:7-115) over a 1-env device engine.

Same surface: reset() -> [x (80,2), xa (10,2)], step(v_action (10,2)) -> ([x, xa], reward, done, {}),
seed(); class constants as in the reference.  Differences that cannot be hidden: randomness comes from
the build's counter-based generator, not numpy's global MT19937, and the returned arrays
are copies (the reference hands out its live state list -- quirk Q9)."""
import numpy as np

from .. import _ffi


class SwarmEnv(object):
    N_LOCUSTS = 80
    N_AGENTS = 10
    GRID_SIZE = 40
    NOISE = 0.0001
    GRAVITY = -1
    WIND_SPEED = 1
    F = 0.5
    L = 10
    dt = 0.05
    N_BURN_IN = 10

    def __init__(self, seed=None, max_episode_steps=None, device_id=0, generator_seed=None):
        self.n_seed = seed
        self.states = None
        self.t = 0
        flags = _ffi.F_RESEED_EACH_RESET if seed else 0    # `if self.n_seed: np.random.seed(...)` on every reset
        gseed = generator_seed if generator_seed is not None else (seed if seed else 1692)
        self._eng = _ffi.Engine(_ffi.ENV_SWARM, 1, device_id=device_id, seed=int(gseed), flags=flags,
                                max_episode_steps=int(max_episode_steps or 0))

    # gym.Env public API (gym 0.9.x forwards to the underscore hooks)
    def reset(self):
        return self._reset()

    def step(self, action):
        return self._step(action)

    def seed(self, seed=None):
        return []

    def _reset(self):
        self._eng.reset()
        self.t = self.N_BURN_IN
        self.states = [self._eng.get_state("SWARM_X")[0], self._eng.get_state("SWARM_XA")[0]]
        return self.states

    def _step(self, v_action, add_wind=True):
        if not add_wind:
            raise NotImplementedError("add_wind=False is never used by the reference's callers")
        a = np.asarray(v_action, dtype=np.float32).reshape(1, self.N_AGENTS, 2)
        self._eng.step(a)
        reward = float(self._eng.read("reward_f64")[0])
        done = bool(self._eng.read("done")[0])
        self.states = [self._eng.get_state("SWARM_X")[0], self._eng.get_state("SWARM_XA")[0]]
        return self.states, reward, done, {}
