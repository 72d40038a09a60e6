"""Host-side sharding plan for N GPUs: environments are independent (no cross-env term anywhere in
the reference), so -- exactly like the reference's `np.split(emulators, workers)`
(fed_gym/agents/paac/runners.py:18-19) -- each rank owns a contiguous block of env ids.  Generator
streams are keyed by GLOBAL env id, so a sharded run reproduces the single-GPU run bit for bit.
Pure Python; exercised on CPU by tests/test_sharding_gloo.py and tests/test_spawn_ranks.py (world size 2 and 3)."""


def shard_range(total_envs, rank, world_size):
    """[lo, hi) of the env ids rank owns; np.split semantics: total must divide evenly."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size %r/%r" % (rank, world_size))
    if total_envs % world_size != 0:
        # np.split raises ValueError('array split does not result in an equal division')
        raise ValueError("array split does not result in an equal division: %d envs over %d ranks" % (total_envs, world_size))
    per = total_envs // world_size
    return rank * per, (rank + 1) * per


def env_id_offset(rank, envs_per_rank):
    """Global id of local env 0 under weak scaling (fixed envs per GPU)."""
    return rank * envs_per_rank


def global_mean_from_shards(local_sum, local_count, all_reduce_sum):
    """Mean over the whole batch from per-rank partial sums: the reference's losses are means over
    the full T*B batch (policy_v_network.py:54,62), so gradients are summed over ranks and divided
    by the global count.  `all_reduce_sum` maps a float to its sum over ranks."""
    return all_reduce_sum(local_sum) / all_reduce_sum(float(local_count))
