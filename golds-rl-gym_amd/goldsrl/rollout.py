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
        self.host_allreduce = None      # callable(flat float32 gradient) -> summed gradient, world size: gradient exchange on the host

    def run(self):
        self.net.rollout(self.T, self.reward_layout)
        if not self.train:
            return
        if self.host_allreduce is None:
            self.last_stats = self.net.train_rollout(self.lr)        # RCCL all-reduce inside when a communicator is attached
        else:       # no device communicator: local gradient -> host exchange -> clip + Adam on the mean of the ranks' means
            self.net.train_rollout_grads()
            summed, world = self.host_allreduce(self.net.get_grads())
            self.net.set_grads(summed)
            self.last_stats = self.net.apply_grads(self.lr, 1.0 / world)
