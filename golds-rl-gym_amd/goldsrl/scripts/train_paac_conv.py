"""`python -m goldsrl.scripts.train_paac_conv` -- the reference's scripts/train_paac_conv.py CLI
(same flags and defaults, :95-121) driving the device learner."""
import argparse
import copy
import logging
import sys

from goldsrl.agents.paac import environment_creator
from goldsrl.agents.paac.emulator_runner import SwarmRunner
from goldsrl.agents.paac.paac import GridPAACLearner
from goldsrl.agents.paac.policy_v_network import ConvSingleAgentPolicyNetwork
from goldsrl.agents.state_processors import SwarmStateProcessor

logging.basicConfig(stream=sys.stdout, level=logging.INFO)


def bool_arg(text):
    """'true' / 'false' (any case) -> bool, as the reference's flag type does."""
    table = {'true': True, 'false': False}
    if text.lower() not in table:
        raise argparse.ArgumentTypeError("Expected True or False, but got {}".format(text))
    return table[text.lower()]


def get_network_and_environment_creator(args, random_seed=3):
    env_creator = environment_creator.SwarmEnvironmentCreator()
    args.num_actions = env_creator.num_actions
    args.random_seed = random_seed
    network_conf = {
        'num_actions': args.num_actions, 'entropy_regularisation_strength': args.entropy_regularisation_strength,
        'device': args.device, 'height': args.height, 'width': args.height, 'channels': 3, 'filters': args.filters,
        'conv_layers': 2, 'scale': args.scale, 'clip_norm': args.clip_norm, 'clip_norm_type': args.clip_norm_type,
        'static_size': args.static_size, 'temporal_size': args.temporal_size, 'static_hidden_size': args.static_hidden_size,
        'rnn_hidden_size': args.temporal_hidden_size,
    }

    def network_creator(name='local_learning'):
        conf = copy.copy(network_conf)
        conf['name'] = name
        return ConvSingleAgentPolicyNetwork(conf)

    return network_creator, env_creator


# (flag strings, dest, type, default): the reference script's options and defaults (scripts/train_paac_conv.py:95-121)
_REFERENCE_FLAGS = (
    (('-d', '--device'), 'device', str, '/gpu:0'),
    (('--e',), 'e', float, 0.1),
    (('--alpha',), 'alpha', float, 0.99),
    (('-lr', '--initial_lr'), 'initial_lr', float, 1e-4),
    (('-lra', '--lr_annealing_steps'), 'lr_annealing_steps', int, 80000000),
    (('--entropy',), 'entropy_regularisation_strength', float, 0.02),
    (('--clip_norm',), 'clip_norm', float, 40.0),
    (('--clip_norm_type',), 'clip_norm_type', str, 'global'),
    (('--gamma',), 'gamma', float, 0.99),
    (('--max_global_steps',), 'max_global_steps', int, 80000000),
    (('--max_local_steps',), 'max_local_steps', int, 5),
    (('--single_life_episodes',), 'single_life_episodes', bool_arg, False),
    (('-ec', '--emulator_counts'), 'emulator_counts', int, 32),
    (('-ew', '--emulator_workers'), 'emulator_workers', int, 8),
    (('-df', '--debugging_folder'), 'debugging_folder', str, 'logs/'),
    (('-rs', '--random_start'), 'random_start', bool_arg, True),
    (('--scale',), 'scale', float, 1000.0),
    (('--height',), 'height', int, 84),
    (('--filters',), 'filters', int, 32),
    (('--rnn-length',), 'rnn_length', int, 5),
    (('--static-size',), 'static_size', int, 2),
    (('--temporal-size',), 'temporal_size', int, 2),
    (('--static-hidden-size',), 'static_hidden_size', int, 32),
    (('--temporal-hidden-size',), 'temporal_hidden_size', int, 32),
)


def get_arg_parser():
    p = argparse.ArgumentParser(description="PAAC on Swarm-v0 with the conv policy (device engine)")
    for flags, dest, kind, default in _REFERENCE_FLAGS:
        p.add_argument(*flags, dest=dest, type=kind, default=default)
    p.add_argument('--reward-layout', default='broadcast', choices=['broadcast', 'reference'], dest="reward_layout",
                   help="'reference' reproduces paac.py:331-338's (T, E*10) indexing (quirk Q4)")
    p.add_argument('--eval-every', default=30.0, type=float, dest="eval_every",
                   help="seconds between eval episodes on Swarm-eval-v0 (paac.py:277-282); 0 disables the monitor")
    p.add_argument('--checkpoint-every', default=0, type=int, dest="checkpoint_every", help="updates between flat-weights checkpoints")
    p.add_argument('--checkpoint-path', default='checkpoint.npz', type=str, dest="checkpoint_path")
    p.add_argument('--resume', default=None, type=str, help="flat-weights checkpoint to start from")
    p.add_argument('--summaries', default=True, type=bool_arg,
                   help="write global_norm / loss scalars per update to <debugging_folder> (JSON lines + TensorBoard event file)")
    return p


def main(args):
    network_creator, env_creator = get_network_and_environment_creator(args)
    learner = GridPAACLearner(network_creator, env_creator, args, SwarmRunner,
                              state_processor=None if args.emulator_counts > 4096 else SwarmStateProcessor(grid_size=args.height))
    logging.info('Starting training')
    learner.train()
    logging.info('Finished training')


if __name__ == '__main__':
    main(get_arg_parser().parse_args())
