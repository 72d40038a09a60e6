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


def bool_arg(string):
    value = string.lower()
    if value == 'true':
        return True
    elif value == 'false':
        return False
    raise argparse.ArgumentTypeError("Expected True or False, but got {}".format(string))


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


def get_arg_parser():
    p = argparse.ArgumentParser()
    p.add_argument('-d', '--device', default='/gpu:0', type=str, dest="device")
    p.add_argument('--e', default=0.1, type=float, dest="e")
    p.add_argument('--alpha', default=0.99, type=float, dest="alpha")
    p.add_argument('-lr', '--initial_lr', default=0.0001, type=float, dest="initial_lr")
    p.add_argument('-lra', '--lr_annealing_steps', default=80000000, type=int, dest="lr_annealing_steps")
    p.add_argument('--entropy', default=0.02, type=float, dest="entropy_regularisation_strength")
    p.add_argument('--clip_norm', default=40.0, type=float, dest="clip_norm")
    p.add_argument('--clip_norm_type', default="global", dest="clip_norm_type")
    p.add_argument('--gamma', default=0.99, type=float, dest="gamma")
    p.add_argument('--max_global_steps', default=80000000, type=int, dest="max_global_steps")
    p.add_argument('--max_local_steps', default=5, type=int, dest="max_local_steps")
    p.add_argument('--single_life_episodes', default=False, type=bool_arg, dest="single_life_episodes")
    p.add_argument('-ec', '--emulator_counts', default=32, type=int, dest="emulator_counts")
    p.add_argument('-ew', '--emulator_workers', default=8, type=int, dest="emulator_workers")
    p.add_argument('-df', '--debugging_folder', default='logs/', type=str, dest="debugging_folder")
    p.add_argument('-rs', '--random_start', default=True, type=bool_arg, dest="random_start")
    p.add_argument('--scale', default=1000., type=float)
    p.add_argument('--height', default=84, type=int)
    p.add_argument('--filters', default=32, type=int)
    p.add_argument('--rnn-length', default=5, type=int)
    p.add_argument('--static-size', default=2, type=int)
    p.add_argument('--temporal-size', default=2, type=int)
    p.add_argument('--static-hidden-size', default=32, type=int)
    p.add_argument('--temporal-hidden-size', default=32, type=int)
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
