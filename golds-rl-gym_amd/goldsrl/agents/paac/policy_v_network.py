"""Estimator objects with the reference's conf keys (fed_gym/agents/paac/policy_v_network.py,
networks.py:100-167; conf built in scripts/train_paac_conv.py:67-83)."""
from ... import _ffi_field, _ffi_flat, _ffi_net


class ConvSingleAgentPolicyNetwork(object):
    """policy_v_network.py:5-80.  `bind(engine)` attaches the device net to a Swarm batch engine;
    predict() then evaluates mu / sigma / vs for the engine's current observation."""

    def __init__(self, conf):
        self.conf = conf
        self.name = conf.get('name', 'local_learning')
        self.num_actions = conf['num_actions']
        self.clip_norm = conf['clip_norm']
        self.clip_norm_type = conf['clip_norm_type']
        self.device = conf['device']
        self.entropy_beta = conf['entropy_regularisation_strength']
        self.scale = conf['scale']
        self.height, self.width, self.channels = conf['height'], conf['width'], conf['channels']
        self.fc_hidden = 256
        if (self.height, self.width, self.channels) != (84, 84, 3):
            raise ValueError("the device net is built for 84x84x3 inputs (train_paac_conv.py defaults; the conv geometry is compiled in)")
        if not 1 <= self.num_actions <= 4:
            raise ValueError("num_actions must be in 1..4 (the mu / sigma heads are Dense(num_actions), policy_v_network.py:40-43)")
        if self.clip_norm_type not in ('global', 'ignore'):
            # 'local' is broken in the reference too (actor_learner.py:60-61 iterates (grad, var) tuples)
            raise Exception('Norm type not recognized')
        self.net = None

    def bind(self, engine, gamma=0.99, seed=3, chunk=81920):
        clip = self.clip_norm if self.clip_norm_type == 'global' else 0.0
        self.net = _ffi_net.ConvNet(engine, max_chunk_samples=min(chunk, engine.E * 10), scale=self.scale,
                                    entropy_beta=self.entropy_beta, clip_norm=clip, gamma=gamma, num_actions=self.num_actions)
        self.net.set_params(_ffi_net.glorot_uniform_flat(seed, self.num_actions))
        return self

    def predict(self, states=None, session=None):
        """predict(states, session) in the reference; here the states are the bound engine's current ones."""
        return self.net.predict()

    def get_flat_params(self):
        return self.net.get_params()

    def set_flat_params(self, flat):
        self.net.set_params(flat)


class FlatPolicyVNetwork(object):
    """policy_v_network.py:194-264 (GRU(32) over the history window + static MLP, three heads)."""

    def __init__(self, conf):
        self.conf = conf
        self.name = conf.get('name', 'local_learning')
        self.num_actions = conf['num_actions']
        self.clip_norm = conf['clip_norm']
        self.clip_norm_type = conf['clip_norm_type']
        self.device = conf['device']
        self.scale = conf['scale']
        self.static_size = conf['static_size']
        self.temporal_size = conf['temporal_size']
        self.entropy_regularisation_strength = conf['entropy_regularisation_strength']    # unused by the loss (:230-235)
        if conf['static_hidden_size'] != 32 or conf['rnn_hidden_size'] != 32:
            raise ValueError("the device net is built for the reference's default hidden sizes (32)")
        if self.clip_norm_type not in ('global', 'ignore'):
            raise Exception('Norm type not recognized')
        self.net = None

    def bind(self, engine, rnn_length=5, gamma=0.99, seed=3, max_samples=None):
        clip = self.clip_norm if self.clip_norm_type == 'global' else 0.0
        self.net = _ffi_flat.FlatNet(engine, static_size=self.static_size, temporal_size=self.temporal_size, rnn_length=rnn_length,
                                     num_actions=self.num_actions, scale=self.scale, clip_norm=clip, gamma=gamma,
                                     max_samples=max_samples or engine.E * 64,
                                     gae_lambda=float(self.conf.get('gae_lambda', 1.0)))     # <1: the A3C worker's GAE targets
        self.net.set_params(_ffi_flat.default_init_flat(seed, static_size=self.static_size, temporal_size=self.temporal_size,
                                                        num_actions=self.num_actions))
        return self

    def predict(self, states, histories, session=None):
        out = self.net.predict(states, histories)
        return {'mu': out['mu'], 'sigma': out['sigma']}          # the reference's predict returns only these two (:253-264)

    def get_flat_params(self):
        return self.net.get_params()

    def set_flat_params(self, flat):
        self.net.set_params(flat)


class ConvPolicyVFieldNetwork(object):
    """policy_v_network.py:83-191: the field-output estimator (3x3 'same' convs + max-pools, Dense(H*W*A) mu / sigma fields
    gathered at the agents' grid positions).  No script of the reference builds it; tests/estimators_tests.py:152-215 checks its
    output shapes at 32x32x3, 5 filters, 2 conv layers, 3 actions.  `bind(engine)` puts it on the engine's device."""

    def __init__(self, conf):
        self.conf = conf
        self.name = conf.get('name', 'local_learning')
        self.num_actions = conf['num_actions']
        self.clip_norm = conf['clip_norm']
        self.clip_norm_type = conf['clip_norm_type']
        self.device = conf['device']
        self.entropy_beta = conf['entropy_regularisation_strength']
        self.scale = conf['scale']
        self.height, self.width, self.channels = conf['height'], conf['width'], conf['channels']
        self.filters, self.conv_layers = conf['filters'], conf['conv_layers']
        self.fc_hidden = 32
        self.rnn_layers = 2
        self.use_rnn = False          # :88 -- the history placeholder is never consumed
        if self.clip_norm_type not in ('global', 'ignore'):
            raise Exception('Norm type not recognized')
        self.net = None

    def _geometry(self):
        return dict(height=self.height, width=self.width, channels=self.channels, filters=self.filters, conv_layers=self.conv_layers,
                    num_actions=self.num_actions)

    def bind(self, engine, seed=3, max_samples=256):
        clip = self.clip_norm if self.clip_norm_type == 'global' else 0.0
        self.net = _ffi_field.FieldNet(engine, max_samples=max_samples, scale=self.scale, entropy_beta=self.entropy_beta, clip_norm=clip,
                                       **self._geometry())
        self.net.set_params(_ffi_field.glorot_uniform_flat(seed, **self._geometry()))
        return self

    def predict(self, states, histories, positions, session=None):
        """predict(states, histories, positions, session) (:175-191): histories are accepted and ignored, as in the reference."""
        return self.net.predict(states, positions)

    def train(self, states, positions, actions, advantages, critic_target, lr, apply_update=True):
        """one session.run of a train op on the network.loss feed (states, agent_positions, actions, advantages, critic_target)"""
        return self.net.train(states, positions, actions, advantages, critic_target, lr, apply_update)

    def get_flat_params(self):
        return self.net.get_params()

    def set_flat_params(self, flat):
        self.net.set_params(flat)
