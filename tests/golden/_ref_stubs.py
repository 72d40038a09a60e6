"""In-memory stand-ins for the two third-party modules the reference imports but this
image lacks (gym==0.9.4, tensorflow==1.4.1 -- /root/reference/requirements.txt:6,33).

Used ONLY by gen_golden.py, in the build container, to import the *unmodified* reference
from /root/reference and capture golden vectors.  Nothing here ships arithmetic of the
reference; the single piece of third-party behaviour that had to be restated is gym
0.9.4's TimeLimit wrapper (elapsed counter, done when elapsed >= max_episode_steps) and
the gym.make() registry plumbing -- see SURVEY.md section 8c ("parity unpinned" at that boundary).
"""
import importlib
import sys
import types

import numpy as np


def _make_gym():
    gym = types.ModuleType("gym")

    class Env(object):
        # gym 0.9.x: public step/reset/seed forward to the underscore hooks
        def step(self, action):
            return self._step(action)

        def reset(self):
            return self._reset()

        def seed(self, seed=None):
            return self._seed(seed)

        def _seed(self, seed=None):
            return []

    class TimeLimit(Env):
        """gym 0.9.4 wrappers/time_limit.py restated: counts wrapped steps only."""

        def __init__(self, env, max_episode_steps=None):
            self.env = env
            self._max_episode_steps = max_episode_steps
            self._elapsed_steps = 0

        @property
        def unwrapped(self):
            return self.env

        def _past_limit(self):
            return (self._max_episode_steps is not None
                    and self._max_episode_steps <= self._elapsed_steps)

        def _step(self, action):
            observation, reward, done, info = self.env.step(action)
            self._elapsed_steps += 1
            if self._past_limit():
                done = True
            return observation, reward, done, info

        def _reset(self):
            self._elapsed_steps = 0
            return self.env.reset()

        def _seed(self, seed=None):
            return self.env.seed(seed)

    class Box(object):
        def __init__(self, low, high, shape=None):
            if shape is None:
                self.low = np.asarray(low)
                self.high = np.asarray(high)
            else:
                shape = (shape,) if np.isscalar(shape) else tuple(shape)
                self.low = low + np.zeros(shape)
                self.high = high + np.zeros(shape)
            self.shape = self.low.shape

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and (x >= self.low).all() and (x <= self.high).all()

    class Tuple(object):
        def __init__(self, spaces):
            self.spaces = spaces

    class Discrete(object):
        def __init__(self, n):
            self.n = n

    spaces = types.ModuleType("gym.spaces")
    spaces.Box, spaces.Tuple, spaces.Discrete = Box, Tuple, Discrete

    registry = {}

    def register(id, entry_point=None, max_episode_steps=None, kwargs=None, **_):
        registry[id] = (entry_point, max_episode_steps, kwargs or {})

    def make(id):
        entry_point, max_steps, kwargs = registry[id]
        mod_name, cls_name = entry_point.split(":")
        cls = getattr(importlib.import_module(mod_name), cls_name)
        env = cls(**kwargs)
        if max_steps is not None:
            env = TimeLimit(env, max_episode_steps=max_steps)
        return env

    envs = types.ModuleType("gym.envs")
    registration = types.ModuleType("gym.envs.registration")
    registration.register = register
    envs.registration = registration
    envs.make = make
    envs.registry = registry

    wrappers = types.ModuleType("gym.wrappers")

    class Monitor(Env):
        def __init__(self, env, *a, **k):
            self.env = env

        def _step(self, a):
            return self.env.step(a)

        def _reset(self):
            return self.env.reset()

    wrappers.Monitor = Monitor
    wrappers.TimeLimit = TimeLimit

    gym.Env = Env
    gym.spaces = spaces
    gym.envs = envs
    gym.wrappers = wrappers
    gym.make = make
    return {"gym": gym, "gym.spaces": spaces, "gym.envs": envs,
            "gym.envs.registration": registration, "gym.wrappers": wrappers}


class _Anything(types.ModuleType):
    """Permissive attribute bag: tensorflow is imported by reference modules at module
    scope but none of the functions captured into fixtures touch it."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        child = _Anything(self.__name__ + "." + name)
        setattr(self, name, child)
        return child

    def __call__(self, *a, **k):
        return _Anything(self.__name__ + "()")

    def __mro_entries__(self, bases):
        return (object,)


def install(reference_root="/root/reference"):
    for name, mod in _make_gym().items():
        sys.modules.setdefault(name, mod)
    for name in ("tensorflow", "tensorflow.contrib", "tensorflow.contrib.keras"):
        sys.modules.setdefault(name, _Anything(name))
    if reference_root not in sys.path:
        sys.path.insert(0, reference_root)
