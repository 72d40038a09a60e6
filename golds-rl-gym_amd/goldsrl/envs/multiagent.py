"""SwarmEnv facade (reference fed_gym/envs/multiagent.py:7-115) over a 1-env device engine.

Same surface: reset() -> [x (80,2), xa (10,2)], step(v_action (10,2)) -> ([x, xa], reward, done, {}),
seed(); class constants as in the reference.

Randomness.  A SEEDED env (`SwarmEnv(seed=n)`, i.e. `Swarm-eval-v0` with n = 192) reproduces the reference's episode: the
reference calls np.random.seed(n) at every reset and then draws x, xa, the burn-in actions and the two noise tables from
numpy's legacy MT19937 stream in a fixed order (multiagent.py:46-56); the facade draws the same numbers from its own
np.random.RandomState(n) (same algorithm, without touching the global generator) and hands them to the device, which runs the
10 burn-in steps (grl_swarm_reset_injected).  `np_random` then continues that stream, so a monitor that samples its action
noise from it gets the numbers the reference's np.random.normal would return next.  An UNSEEDED env (`Swarm-v0`) draws from
the build's counter-based device generator.  Returned arrays are copies (the reference hands out its live state
list -- quirk Q9)."""
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
        self.np_random = None       # the reference's generator stream of a seeded env, continued after the reset draws

    # gym.Env public API (gym 0.9.x forwards to the underscore hooks)
    def reset(self):
        return self._reset()

    def step(self, action):
        return self._step(action)

    def seed(self, seed=None):
        return []

    def _reset(self):
        if self.n_seed:
            rs = np.random.RandomState(self.n_seed)                       # np.random.seed(self.n_seed) (multiagent.py:47-48)
            x = rs.rand(self.N_LOCUSTS, 2)                                # draw order of multiagent.py:51-56
            xa = rs.rand(self.N_AGENTS, 2)
            random_actions = rs.normal(size=(self.N_BURN_IN, self.N_AGENTS, 2))
            agent_noise = rs.normal(size=(128 + self.N_BURN_IN, self.N_AGENTS, 2))
            particle_noise = rs.normal(size=(128 + self.N_BURN_IN, self.N_LOCUSTS, 2))
            # rows 0..9 drive the burn-in, row 10 every later step (self.t stays 10: quirk Q1); rows 11.. are never read
            self._eng.swarm_reset_injected(x[None], xa[None], random_actions[None], agent_noise[None, :self.N_BURN_IN + 1],
                                           particle_noise[None, :self.N_BURN_IN + 1])
            self.np_random = rs
        else:
            self._eng.reset()
        self.t = self.N_BURN_IN
        self.states = [self._eng.get_state("SWARM_X")[0], self._eng.get_state("SWARM_XA")[0]]
        return self.states

    def _step(self, v_action, add_wind=True):
        v_action = np.asarray(v_action)
        if not add_wind:                        # multiagent.py:35-36 skipped: the agents move by the action alone
            self._eng.swarm_step_opts(v_action.reshape(1, self.N_AGENTS, 2), add_wind=False)
        elif v_action.dtype == np.float32:      # what the worker reads from the float32 shared array (quirk Q7)
            self._eng.step(v_action.reshape(1, self.N_AGENTS, 2))
        else:                                   # a direct caller's float64 actions are used as they are (multiagent.py:30-44)
            self._eng.swarm_step_f64(v_action.astype(np.float64).reshape(1, self.N_AGENTS, 2))
        reward = float(self._eng.read("reward_f64")[0])
        done = bool(self._eng.read("done")[0])
        self.states = [self._eng.get_state("SWARM_X")[0], self._eng.get_state("SWARM_XA")[0]]
        return self.states, reward, done, {}
