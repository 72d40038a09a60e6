"""gym-style registry for the device-backed envs.  Ids, entry points, episode caps and kwargs are
those of the reference's registrations (fed_gym/__init__.py:3-33, fed_gym/envs/fed_env.py:10-27);
`make` applies gym 0.9.4's TimeLimit semantics through the engine's `max_episode_steps`."""
from .fed_env import SolowEnv, SolowSSEnv, TickerEnv, TradeAR1Env, register_solow_env, registry  # noqa: F401
from .multiagent import SwarmEnv  # noqa: F401

registry.update({
    "TradeAR1-v0": (TradeAR1Env, 1024, {}),
    "Solow-v0": (SolowEnv, 1024, {}),
    "SolowSS-v0": (SolowSSEnv, 1024, {}),
    "Swarm-v0": (SwarmEnv, 128, {}),
    "Swarm-eval-v0": (SwarmEnv, 128, dict(seed=192)),
})


def make(env_id):
    """gym.envs.make(id): env wrapped in TimeLimit(max_episode_steps) when the registration has a cap."""
    if env_id not in registry:
        raise KeyError("No registered env with id: %s" % env_id)      # gym.error.UnregisteredEnv
    cls, max_steps, kwargs = registry[env_id]
    return cls(max_episode_steps=max_steps, **kwargs)
