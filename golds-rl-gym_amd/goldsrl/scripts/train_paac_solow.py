"""`python -m goldsrl.scripts.train_paac_solow` -- the reference's scripts/train_paac_solow.py CLI
(same flags and defaults, :96-129) driving the device learner."""
import copy
import logging
import sys

from goldsrl.agents.paac import environment_creator
from goldsrl.agents.paac.emulator_runner import SolowRunner
from goldsrl.agents.paac.paac import PAACLearner
from goldsrl.agents.paac.policy_v_network import FlatPolicyVNetwork
from goldsrl.agents.state_processors import SolowStateProcessor
from goldsrl.scripts.train_paac_conv import get_arg_parser as _conv_parser

logging.basicConfig(stream=sys.stdout, level=logging.INFO)


def get_network_and_environment_creator(args, random_seed=3):
    env_creator = environment_creator.SolowEnvironmentCreator(1, 1)       # train_paac_solow.py:60
    args.num_actions = env_creator.num_actions
    args.random_seed = random_seed
    network_conf = {
        'num_actions': args.num_actions, 'entropy_regularisation_strength': args.entropy_regularisation_strength,
        'device': args.device, 'scale': args.scale, 'clip_norm': args.clip_norm, 'clip_norm_type': args.clip_norm_type,
        'static_size': args.static_size, 'temporal_size': args.temporal_size, 'static_hidden_size': args.static_hidden_size,
        'rnn_hidden_size': args.temporal_hidden_size,
    }

    def network_creator(name='local_learning'):
        conf = copy.copy(network_conf)
        conf['name'] = name
        return FlatPolicyVNetwork(conf)

    return network_creator, env_creator


def get_arg_parser():
    p = _conv_parser()
    p.set_defaults(scale=100.)              # train_paac_solow.py:122 (the conv script uses 1000)
    return p


def main(args):
    network_creator, env_creator = get_network_and_environment_creator(args)
    learner = PAACLearner(network_creator, env_creator, args, SolowRunner, SolowStateProcessor())
    logging.info('Starting training')
    learner.train()
    logging.info('Finished training')


if __name__ == '__main__':
    main(get_arg_parser().parse_args())
