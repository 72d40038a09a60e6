"""ActorLearner base (reference fed_gym/agents/paac/actor_learner.py:10-124): argument plumbing, reward
clip, linear lr anneal.  TF session / saver / summary plumbing has no counterpart."""


class ActorLearner(object):
    def __init__(self, network_creator, environment_creator, args, emulator_class):
        self.global_step = 0
        self.emulator_class = emulator_class
        self.max_local_steps = args.max_local_steps
        self.num_actions = args.num_actions
        self.initial_lr = args.initial_lr
        self.lr_annealing_steps = args.lr_annealing_steps
        self.emulator_counts = args.emulator_counts
        self.device = args.device
        self.debugging_folder = args.debugging_folder
        self.max_global_steps = args.max_global_steps
        self.gamma = args.gamma
        self.eval_every = getattr(args, 'eval_every', 0.0)              # seconds between eval episodes (reference: 30 s, in a thread)
        self.checkpoint_every = getattr(args, 'checkpoint_every', 0)    # updates between flat-weights checkpoints (0: never)
        self.checkpoint_path = getattr(args, 'checkpoint_path', 'checkpoint.npz')
        self.resume = getattr(args, 'resume', None)
        if getattr(args, 'max_episode_steps', None):       # TimeLimit of the registered id (fed_gym/__init__.py); tests shorten it
            self.max_episode_steps = int(args.max_episode_steps)
        if getattr(args, 'seed', None) is not None:
            self.seed = int(args.seed)
        self.summary_writer = None      # created by train() when args.summaries is set (actor_learner.py:79-83)
        self.summaries = bool(getattr(args, 'summaries', False))
        self.network_creator = network_creator
        self.environment_creator = environment_creator
        self.network = network_creator()
        self.runners = None
        self.total_rewards = []     # reward per step of every finished training episode (paac.py:60,150,342)
        self.episode_log = []       # (global_step, env, length, total_reward) per finished episode = the `rl/reward` points
        self.ranks = None           # goldsrl.distributed.Ranks, formed by train()
        self.gradient_exchange = "none"

    def rescale_reward(self, reward, lb=-2, ub=2):
        """Clip immediate reward (actor_learner.py:91-97)."""
        if reward > ub:
            reward = ub
        elif reward < lb:
            reward = lb
        return reward

    def get_lr(self):
        """actor_learner.py:115-119"""
        if self.global_step <= self.lr_annealing_steps:
            return self.initial_lr - (self.global_step * self.initial_lr / self.lr_annealing_steps)
        return 0.0

    def _open_summaries(self):
        """One writer per rank, each in its own directory: rank 0 in debugging_folder (the reference's single writer,
        actor_learner.py:29), rank r > 0 in debugging_folder/rank<r> for the `rl/reward` points of ITS env shard."""
        if self.summaries and self.summary_writer is None:
            import os
            from .policy_monitor import ScalarWriter
            rank = self.ranks.rank if self.ranks is not None else 0
            folder = self.debugging_folder if rank == 0 else os.path.join(self.debugging_folder, "rank%d" % rank)
            self.summary_writer = ScalarWriter(folder)
        return self.summary_writer

    def _log_update(self, stats):
        """'global_norm' (actor_learner.py:83) and the loss terms of one update, keyed by global_step.  The values are
        replicated over ranks (the gradient is all-reduced): rank 0 writes them once."""
        w = self.summary_writer
        if w is None or not stats or (self.ranks is not None and self.ranks.rank != 0):
            return
        w.add_scalar("global_norm", stats["global_norm"], self.global_step)
        w.add_scalar("loss/total", stats["loss"], self.global_step)
        w.add_scalar("loss/policy", stats["policy_loss"], self.global_step)
        w.add_scalar("loss/critic_mean", stats["critic_loss_mean"], self.global_step)
        w.flush()

    def save_checkpoint(self, path):
        """Flat-weights checkpoint of the bound estimator + global_step (the reference's Saver path, actor_learner.py:70-89,
        is dead code in its scripts)."""
        self.network.net.save_checkpoint(path, global_step=self.global_step)

    def load_checkpoint(self, path):
        extra = self.network.net.load_checkpoint(path)
        self.global_step = int(extra.get("global_step", self.global_step))

    def cleanup(self):
        pass
