"""reference fed_gym/agents/paac/environment_creator.py:4-20."""
from ... import envs


class SolowEnvironmentCreator(object):
    def __init__(self, p, q):
        self.num_actions = 1
        self.p, self.q = p, q
        envs.register_solow_env(p, q)
        self.create_environment = lambda: envs.make("Solow-%s-%s-finite-v0" % (p, q))


class SwarmEnvironmentCreator(object):
    def __init__(self):
        self.num_actions = 2
        self.create_environment = lambda: envs.make("Swarm-v0")
