"""Device-resident PAAC rollout + update: the body of GridPAACLearner.train()'s while-loop (reference
fed_gym/agents/paac/paac.py:302-387; Swarm, conv policy) and of PAACLearner.train()'s (paac.py:119-196; Solow / TradeAR1,
FlatPolicyVNetwork) as two C calls each, plus the gradient exchange when the env batch is sharded over ranks."""
from . import _ffi_net


class _GradientExchange(object):
    """What both rollouts do with the gradient: RCCL all-reduce inside train_rollout when a communicator is attached to the net,
    otherwise (host_allreduce set) local gradient -> host sum over ranks -> clip + Adam on the mean of the ranks' means."""
    host_allreduce = None      # callable(flat float32 gradient) -> (summed gradient, world size)
    ranks = None               # set with host_allreduce: the ranks agree on the outcome of the local pass before they exchange

    def _update(self):
        if self.host_allreduce is None:
            return self.net.train_rollout(self.lr)
        err = None
        try:
            self.net.train_rollout_grads()
        except Exception as e:       # noqa: BLE001 -- e.g. GRL_E_RANGE on this rank only
            err = e
        if self.ranks is not None and self.ranks.min(0 if err else 1) == 0:
            # every rank leaves the update together: nobody is left waiting in the exchange, no replica is updated
            raise err if err else RuntimeError("the gradient pass failed on another rank; no replica was updated")
        if err:
            raise err
        # a conv net whose ROLLOUT left the fp16 range gave this update up (goldsrl_net.h, Arithmetic): then no replica applies one
        skipped = 1 if (hasattr(self.net, "range_info") and self.net.range_info()["update_skipped"]) else 0
        if self.ranks is not None:
            skipped = int(self.ranks.max(skipped))
        if skipped:
            nan = float("nan")
            return {"loss": nan, "policy_loss": nan, "critic_loss_mean": nan, "global_norm": nan}
        summed, world = self.host_allreduce(self.net.get_grads())
        self.net.set_grads(summed)
        return self.net.apply_grads(self.lr, 1.0 / world)


class ConvPolicyRollout(_GradientExchange):
    def __init__(self, eng, T, train=True, lr=1e-4, reward_layout=0, seed=3, chunk=81920, **net_kw):
        self.eng, self.T, self.train, self.lr, self.reward_layout = eng, T, train, lr, reward_layout
        import os
        if os.environ.get("GRL_NET_CHUNK"):      # tuning knob: samples per chunk (a multiple of 10, <= 131 072)
            chunk = int(os.environ["GRL_NET_CHUNK"])
        chunk = min(chunk, eng.E * 10)
        self.net = _ffi_net.ConvNet(eng, max_chunk_samples=chunk, **net_kw)
        self.net.set_params(_ffi_net.glorot_uniform_flat(seed))
        self.last_stats = None

    def run(self):
        self.net.rollout(self.T, self.reward_layout)
        if self.train:
            self.last_stats = self._update()


class FlatPolicyRollout(_GradientExchange):
    """Solow (BASELINE config 2) or TradeAR1 (config 5) handle + FlatPolicyVNetwork with the sizes the env dictates."""

    def __init__(self, eng, T, train=True, lr=1e-4, seed=3, **net_kw):
        from . import _ffi, _ffi_flat
        self.eng, self.T, self.train, self.lr = eng, T, train, lr
        if eng.kind == _ffi.ENV_TRADE:
            S = 1 + 2 * eng.cfg.n_assets      # INPUT_SIZE = TEMPORAL_SIZE = 1+2n, NUM_ACTIONS = n (scripts/train_trade.py:38-40)
            sizes = dict(static_size=S, temporal_size=S, num_actions=eng.cfg.n_assets)
        else:
            sizes = dict(static_size=2, temporal_size=2, num_actions=1)
        net_kw.setdefault("rnn_length", eng.cfg.rnn_length)
        net_kw.setdefault("max_samples", T * eng.E)
        sizes.update(net_kw)
        self.net = _ffi_flat.FlatNet(eng, **sizes)
        self.net.set_params(_ffi_flat.default_init_flat(seed, static_size=sizes["static_size"], temporal_size=sizes["temporal_size"],
                                                        num_actions=sizes["num_actions"]))
        self.last_stats = None
        # A rollout that is trained on keeps its activations for the gradient step instead of recomputing them there.  Measured
        # (tools/flat_keep_cost.py, ms rollout + gradient step, recompute -> keep): TradeAR1-16, rnn 20, 8 192 envs 1.28 + 2.31 -> 1.44 + 1.43;
        # Solow, rnn 5, 4 096 envs 0.51 + 0.72 -> 0.59 + 0.50.  GRL_FLAT_KEEP=0 / 1 forces it (the A/B switch and the equality test).
        import os
        self.keep_activations = os.environ.get("GRL_FLAT_KEEP", "1") != "0"

    def run(self):
        # a rollout that is trained on keeps its activations for the gradient step (GRL_FLAT_KEEP=0: recompute them, the A/B switch)
        self.net.set_keep_activations(self.train and self.keep_activations)
        self.net.rollout(self.T)
        if self.train:
            self.last_stats = self._update()
