"""Environment creators: the two objects the reference's training scripts hand to the learner
(fed_gym/agents/paac/environment_creator.py:4-20) -- `num_actions` plus a zero-argument `create_environment`."""
from ... import envs


class _Creator(object):
    ENV_ID = None
    NUM_ACTIONS = 0

    def __init__(self):
        self.num_actions = self.NUM_ACTIONS
        env_id = self.ENV_ID
        self.create_environment = lambda: envs.make(env_id)


class SwarmEnvironmentCreator(_Creator):
    ENV_ID = "Swarm-v0"
    NUM_ACTIONS = 2


class SolowEnvironmentCreator(_Creator):
    NUM_ACTIONS = 1

    def __init__(self, p, q):
        self.p, self.q = p, q
        envs.register_solow_env(p, q)                     # the reference registers Solow-p-q-finite-v0 at import time
        self.ENV_ID = "Solow-{}-{}-finite-v0".format(p, q)
        super(SolowEnvironmentCreator, self).__init__()
