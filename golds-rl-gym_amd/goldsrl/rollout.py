"""Device-resident PAAC rollout + update for Swarm with the conv policy: the body of
GridPAACLearner.train()'s while-loop (reference fed_gym/agents/paac/paac.py:302-387) as two C calls."""
from . import _ffi_net


class ConvPolicyRollout(object):
    def __init__(self, eng, T, train=True, lr=1e-4, reward_layout=0, seed=3, chunk=40960, **net_kw):
        self.eng, self.T, self.train, self.lr, self.reward_layout = eng, T, train, lr, reward_layout
        chunk = min(chunk, eng.E * 10)
        self.net = _ffi_net.ConvNet(eng, max_chunk_samples=chunk, **net_kw)
        self.net.set_params(_ffi_net.glorot_uniform_flat(seed))
        self.last_stats = None

    def run(self):
        self.net.rollout(self.T, self.reward_layout)
        if self.train:
            self.last_stats = self.net.train_rollout(self.lr)
