"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT PATH.

A batched numpy (float64) restatement of the reference's algorithm for the PAAC hot path
(SURVEY.md section 8a).  Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline`
leg may import this module, and only as the checker.  The product (golds-rl-gym_amd/) never
imports it and has no CPU fallback.

Pinning: every function below is checked against golden vectors captured from the
UNMODIFIED reference run in the build container (tests/golden/*.npz, generator
tests/golden/gen_golden.py) by tests/test_oracle_golden.py.  Two boundaries are
"parity unpinned" because the reference delegates them to third-party packages absent from
/root/reference and from this image:
  * gym==0.9.4 TimeLimit (requirements.txt:6)  -> restated in `time_limit_done`;
  * tensorflow==1.4.1 layers/GRUCell/Adam (requirements.txt:33) -> restated in oracle/nets.py.

All paths cited are relative to /root/reference/.
"""
import numpy as np

# ----------------------------------------------------------------------------- Swarm
# constants: fed_gym/envs/multiagent.py:8-21
N_LOCUSTS, N_AGENTS = 80, 10
NOISE, GRAVITY, WIND_SPEED, F_ATT, L_ATT, DT, N_BURN_IN = 1e-4, -1.0, 1.0, 0.5, 10.0, 0.05, 10
BOX_WIDTH, BOX_HEIGHT = 3.0, 3.0     # agents/state_processors.py:22-23


def _cutoff(x, v):
    """xv_cutoff (envs/multiagent.py:77-86), batched, in place: rows with y<=0 are put on the
    ground, lose their horizontal velocity and any downward velocity."""
    g = x[..., 1] <= 0
    x[..., 1][g] = 0
    v[..., 0][g] = 0
    down = g & (v[..., 1] <= 0)
    v[..., 1][down] = 0


def swarm_x_update(x, v, noise_scaled):
    """x_update (envs/multiagent.py:70-75): cutoff; x += dt*v + noise; cutoff.  In place."""
    _cutoff(x, v)
    x += DT * v + noise_scaled
    _cutoff(x, v)
    return x


def _s(r):
    """s(r,F,L) (envs/multiagent.py:65-68)."""
    return F_ATT * np.exp(-r / L_ATT) - np.exp(-r)


def swarm_v_calculate(x, xa):
    """v_calculate (envs/multiagent.py:88-115) for a batch: x (E,80,2), xa (E,10,2).
    Returns v (E,80,2) and reward (E,) = -mean_j |v_j|^2.  Summation order follows numpy's:
    the per-target sums are 1-D contiguous `.sum()` calls in the reference (pairwise,
    8 accumulators), reproduced here by reducing over a contiguous last axis."""
    # D[e, j, i] = distance from source i to target locust j  (reference builds D[i, j])
    dx_ll = x[:, None, :, 0] - x[:, :, None, 0]          # x_i - x_j  -> (E, j, i)
    dy_ll = x[:, None, :, 1] - x[:, :, None, 1]
    d_ll = np.sqrt(np.square(dx_ll) + np.square(dy_ll))
    dx_al = xa[:, None, :, 0] - x[:, :, None, 0]         # (E, j, a)
    dy_al = xa[:, None, :, 1] - x[:, :, None, 1]
    d_al = np.sqrt(np.square(dx_al) + np.square(dy_al))
    s_ll, s_al = _s(d_ll), _s(d_al)
    v0 = np.ascontiguousarray(s_ll * dx_ll / (d_ll + 0.000001)).sum(axis=-1)
    v1 = np.ascontiguousarray(s_ll * dy_ll / (d_ll + 0.000001)).sum(axis=-1)
    a0 = np.ascontiguousarray(s_al * dx_al / (d_al + 0.000001)).sum(axis=-1)
    a1 = np.ascontiguousarray(s_al * dy_al / (d_al + 0.000001)).sum(axis=-1)
    v = np.empty_like(x)
    v[..., 0] = (WIND_SPEED + v0) + a0
    v[..., 1] = (GRAVITY + v1) + a1
    energy = np.ascontiguousarray((v ** 2).sum(axis=-1)).sum(axis=-1) / N_LOCUSTS
    return v, -energy


def swarm_step(x, xa, action, agent_noise_row, particle_noise_row, add_wind=True):
    """SwarmEnv._step (envs/multiagent.py:30-44), batched.  Inputs are NOT modified
    (the reference mutates in place -- quirk Q9); returns (x', xa', reward, done).
    The action keeps its dtype, as `v_action.copy()` does: a float32 row (what the worker reads from the learner's shared
    array, quirk Q7) gets its wind added and its `dt * v` product taken in float32 -- pinned by tests/golden/swarm_runner.npz."""
    x = np.array(x, dtype=np.float64)
    xa = np.array(xa, dtype=np.float64)
    v_action = np.array(action)
    if v_action.dtype != np.float32:
        v_action = v_action.astype(np.float64)
    if add_wind:
        v_action[..., 0] += WIND_SPEED
    xa = swarm_x_update(xa, v_action, NOISE * np.asarray(agent_noise_row, dtype=np.float64))
    v, reward = swarm_v_calculate(x, xa)
    x = swarm_x_update(x, v, NOISE * np.asarray(particle_noise_row, dtype=np.float64))
    return x, xa, reward, reward >= 0


def swarm_burn_in(x0, xa0, random_actions, agent_noise, particle_noise):
    """The 10 burn-in steps of SwarmEnv._reset (envs/multiagent.py:58-61) with noise rows
    0..9; afterwards t == 10 and stays there (quirk Q1: row 10 is reused by every step).
    Shapes: x0 (E,80,2) xa0 (E,10,2) random_actions (E,10,10,2) *_noise (E,>=10,N,2)."""
    x, xa = np.array(x0, dtype=np.float64), np.array(xa0, dtype=np.float64)
    for t in range(N_BURN_IN):
        x, xa, _, _ = swarm_step(x, xa, random_actions[:, t], agent_noise[:, t], particle_noise[:, t])
    return x, xa


def swarm_edges(mean_x, grid):
    """Bin edges of np.histogram2d(..., bins=grid, range=box) (agents/state_processors.py:25-33):
    np.linspace(lo, hi, grid+1) == arange(grid+1)*step + lo with the last edge forced to hi."""
    lo, hi = mean_x - BOX_WIDTH / 2.0, mean_x + BOX_WIDTH / 2.0
    step = (hi - lo) / grid
    ex = np.arange(0, grid + 1) * step + lo
    ex[-1] = hi
    ylo, yhi = 0.0, 2 * BOX_HEIGHT
    ystep = (yhi - ylo) / grid
    ey = np.arange(0, grid + 1) * ystep + ylo
    ey[-1] = yhi
    return ex, ey


def _hist_bin(v, edges, grid):
    """histogramdd's bin rule: searchsorted(right)-1, value == last edge goes to the last
    bin, everything else outside is dropped (returned as -1)."""
    k = np.searchsorted(edges, v, side="right") - 1
    k = np.where(v == edges[-1], grid - 1, k)
    return np.where((k < 0) | (k >= grid), -1, k)


def swarm_observe_compact(x, xa, grid=84):
    """process_state (agents/state_processors.py:29-42) in compact form for ONE env:
    locust bins (80,2) int (-1 = outside the box), agent bins (10,2) int (-1 = outside) and
    `positions` (10,2) uint8 = digitize() clamped to grid-1 (1-based, quirk Q2)."""
    pts = np.vstack([x, xa])
    mean_x = np.mean(pts, axis=0)[0]
    ex, ey = swarm_edges(mean_x, grid)
    lb = np.stack([_hist_bin(x[:, 0], ex, grid), _hist_bin(x[:, 1], ey, grid)], axis=1)
    ab = np.stack([_hist_bin(xa[:, 0], ex, grid), _hist_bin(xa[:, 1], ey, grid)], axis=1)
    lb[(lb < 0).any(axis=1)] = -1
    ab[(ab < 0).any(axis=1)] = -1
    px = np.searchsorted(ex, xa[:, 0], side="right")
    py = np.searchsorted(ey, xa[:, 1], side="right")
    pos = np.stack([np.minimum(px, grid - 1), np.minimum(py, grid - 1)], axis=1).astype(np.uint8)
    return lb, ab, pos


def swarm_grid_from_compact(lb, ab, grid=84):
    """Dense (grid,grid,2) float64 density: counts/80 and counts/10."""
    g = np.zeros((grid, grid, 2))
    for bx, by in lb:
        if bx >= 0:
            g[bx, by, 0] += 1.0
    for bx, by in ab:
        if bx >= 0:
            g[bx, by, 1] += 1.0
    g[:, :, 0] /= N_LOCUSTS
    g[:, :, 1] /= N_AGENTS
    return g


def swarm_local_states(g, positions):
    """SwarmRunner.get_local_states (agents/paac/emulator_runner.py:98-111)."""
    out = np.zeros((len(positions),) + g.shape[:2] + (3,))
    out[..., :2] = g[None]
    for a, (px, py) in enumerate(positions):
        out[a, px, py, 2] = 1.0
    return out


def swarm_transform_actions(actions):
    """SwarmRunner.transform_actions_for_env (agents/paac/emulator_runner.py:113-118):
    rows with ||a||_2 >= 1 are divided by their norm.  Keeps the input dtype."""
    a = np.array(actions)
    d = np.sqrt(np.sum(a * a, axis=-1))
    m = d >= 1
    a[m] = a[m] / d[m][:, None]
    return a


def swarm_history_window(local_states, n_hist, rnn_length):
    """HISTORY slot of SwarmRunner._run (agents/paac/emulator_runner.py:129-145) AS IT BEHAVES, one env: local_states
    (10,G,G,3) is the env's current STATE row, n_hist = len(self.histories[i]) after this step's append (1 after a reset or the
    first step).  Every list entry is a view of the shared STATE row (quirk Q11), so the window is min(n, rnn) copies of the
    CURRENT state; while n < rnn the worker pads through keras' pad_sequences with its DEFAULT dtype int32, which truncates
    the densities toward zero (k/80 -> 0 unless all locusts share the bin; the one-hot survives) -- restated from Keras 2.0.8,
    third party, parity unpinned at that call.  Returns (10, rnn, G, G, 3) float64, agent-major as np.swapaxes leaves it."""
    out = np.zeros((local_states.shape[0], rnn_length) + local_states.shape[1:])
    n = min(int(n_hist), rnn_length)
    cur = local_states if n_hist >= rnn_length else np.trunc(local_states)
    out[:, :n] = cur[:, None]
    return out


def time_limit_done(elapsed_after_step, max_episode_steps):
    """gym==0.9.4 TimeLimit (third party, not in /root/reference; restated, parity unpinned):
    the wrapper counts wrapped steps and forces done once elapsed >= max_episode_steps."""
    return elapsed_after_step >= max_episode_steps


# ----------------------------------------------------------------------------- Solow
SOLOW_ALPHA = 0.33


def solow_rhos(p, q):
    """SolowEnv.__init__ (envs/fed_env.py:179-189)."""
    if p > 0:
        rho_z = 0.5 ** np.arange(1, p + 1)
        rho_z /= rho_z.sum() / 0.95
    else:
        rho_z = np.array([0.95])
    rho_e = 0.5 ** np.arange(1, q + 1) if q > 0 else np.array([0.5])
    return rho_z, rho_e


def solow_k_ss(savings, delta=0.02):
    """_k_ss (envs/fed_env.py:198-199)."""
    return (savings / delta) ** (1 / (1 - SOLOW_ALPHA))


def solow_step(k, z, e, e_t, s, rho_z, rho_e, delta=0.02):
    """SolowEnv._step (envs/fed_env.py:201-236), batched over envs.
    k (E,), z (E,p) oldest..newest, e (E,q), e_t (E,) shock popped from the END of the tape,
    s (E,) savings rate.  Returns k', z', e', obs (E,2), reward (E,)."""
    s = np.maximum(1e-3, s)
    y = np.exp(z[:, -1]) * k ** SOLOW_ALPHA
    k_next = (1 - delta) * k + s * y
    z_next = (rho_z * z).sum(axis=1) + (rho_e * e).sum(axis=1) + e_t
    z_new = np.concatenate([z[:, 1:], z_next[:, None]], axis=1)
    e_new = np.concatenate([e[:, 1:], e_t[:, None]], axis=1)
    reward = np.log((1 - s) * y + 1e-4)
    return k_next, z_new, e_new, np.stack([k_next, z_next], axis=1), reward


def solow_process_state(obs):
    """SolowStateProcessor.process_state (agents/state_processors.py:11-12,69-71)."""
    return obs / np.array([100.0, 1.0])


def sigmoid(x):
    """a3c.worker.sigmoid (agents/a3c/worker.py:17-34), branch-stable logistic."""
    x = np.asarray(x, dtype=np.float64)
    z = np.exp(-np.abs(x))
    return np.where(x >= 0, 1 / (1 + z), z / (1 + z))


def history_window(cur_state, n_since_reset, rnn_length):
    """EmulatorRunner._run history (agents/paac/emulator_runner.py:50-63) AS IT BEHAVES: every
    list entry is a *view* of the env's row in the shared STATE variable, so the window is
    min(n, rnn_length) copies of the CURRENT processed state, zero-padded at the end
    (quirk Q11, pinned by tests/golden/solow_runner.npz).  n counts states since reset (>=1)."""
    E, S = cur_state.shape
    h = np.zeros((E, rnn_length, S))
    n = np.minimum(n_since_reset, rnn_length)
    for r in range(rnn_length):
        h[:, r, :] = np.where((r < n)[:, None], cur_state, 0.0)
    return h


# ----------------------------------------------------------------------------- TradeAR1
def trade_std_e(std_p=0.05, rho_p=0.9):
    """TradeAR1Env.__init__ (envs/fed_env.py:277-278)."""
    return np.sqrt((std_p ** 2) * (1 - rho_p ** 2))


def trade_step(cash, assets, quantity, prices, action, normals, std_e, rho_p=0.9):
    """TradeAR1Env._step (envs/fed_env.py:300-321), batched.  cash, assets (E,);
    quantity, prices, action, normals (E,n).  Returns new (cash, assets, quantity, prices),
    obs (E,1+2n), reward (E,), done (E,)."""
    n = action.shape[1]
    buy = action > 0
    q_add = np.where(buy, (action / n) * cash[:, None] / prices, action * quantity)
    quantity = quantity + q_add
    cash = cash + -(q_add * prices).sum(axis=1)
    new_assets = cash + np.sum(quantity * prices, axis=1)
    done = new_assets < 1.0
    new_prices = (prices ** rho_p) * np.exp(std_e * normals)
    obs = np.concatenate([cash[:, None], quantity, new_prices], axis=1)
    reward = np.log(new_assets + 1e-4) - np.log(assets + 1e-4)
    return cash, new_assets, quantity, new_prices, obs, reward, done


def trade_process_state(raw):
    """TradeWorker.process_state (agents/a3c/worker.py:420-431) AS IT BEHAVES: n_assets is taken
    as len(raw)-1 (=2n), so the 'quantity' slice swallows the prices too and EVERY entry after
    cash gets log(.+1); the log(prices) branch sees an empty slice (quirk Q10, pinned by
    tests/golden/trade.npz)."""
    raw = np.asarray(raw, dtype=np.float64)
    return np.concatenate([np.log(raw[..., :1] + 1e-4), np.log(raw[..., 1:] + 1)], axis=-1)


# ----------------------------------------------------------------------------- rollout math
def rescale_reward(r, lb=-2.0, ub=2.0):
    """ActorLearner.rescale_reward (agents/paac/actor_learner.py:91-97)."""
    return np.clip(r, lb, ub)


def nstep_returns(rewards, values, boot, gamma, masks=None):
    """PAAC n-step return / advantage (agents/paac/paac.py:159-172 masked; :360-365 unmasked).
    rewards, values (T,B); boot (B,); masks (T,B) = 1-done or None.  float64 accumulation,
    EXCEPT that the bootstrap value keeps its dtype: the network hands back float32, so the
    first `gamma * est` product is rounded to float32 (numpy scalar*f32-array rule) before it
    meets the float64 rewards -- pinned by tests/golden/returns.npz."""
    T = rewards.shape[0]
    y = np.zeros(rewards.shape)
    adv = np.zeros(rewards.shape)
    est = np.copy(boot)
    for t in reversed(range(T)):
        est = rewards[t] + gamma * est * masks[t] if masks is not None else rewards[t] + gamma * est
        y[t] = est
        adv[t] = est - values[t]
    return y, adv


def episode_bookkeeping(rewards, dones, total=None, steps=None, global_step=0):
    """The learner's per-env bookkeeping inside the rollout loop (agents/paac/paac.py:142-157 flat, :331-349 grid; the grid
    form reads column 0 of the broadcast reward/done rows, i.e. the env's scalar).  rewards (T,E) float32 as the shared
    array holds them, dones (T,E).  Returns (records, total, steps, global_step): records in append order, each
    (global_step at the add_summary call, env, episode length, total_episode_reward) -- `rl/reward` is the total,
    total_rewards.append(total / length).  total_episode_rewards starts as int 0, so under the reference's numpy 1.13.3
    `0 + np.float32` promotes to float64 and the sum stays float64 (numpy 2 would keep float32): float64 here."""
    T, E = rewards.shape
    total = np.zeros(E, np.float64) if total is None else np.array(total, np.float64)
    steps = np.zeros(E, np.int64) if steps is None else np.array(steps, np.int64)
    records = []
    for t in range(T):
        for e in range(E):
            total[e] += np.float64(np.float32(rewards[t, e]))
            steps[e] += 1
            global_step += 1
            if dones[t, e]:
                records.append((global_step, e, int(steps[e]), float(total[e])))
                total[e] = 0.0
                steps[e] = 0
    return records, total, steps, global_step


def gae(rewards, values, boot, gamma, lam):
    """A3C GAE (agents/a3c/worker.py:232-239, 284-294): delta_t = r_t + g V_{t+1} - V_t,
    adv = lfilter([1],[1,-g*lam]) over reversed time, target = adv + V_t.  (T,B) inputs."""
    T = rewards.shape[0]
    v_all = np.concatenate([values, boot[None]], axis=0).astype(np.float64)
    adv = np.zeros(rewards.shape)
    run = np.zeros(rewards.shape[1:])
    for t in reversed(range(T)):
        delta = rewards[t] + gamma * v_all[t + 1] - v_all[t]
        run = delta + (gamma * lam) * run
        adv[t] = run
    return adv, adv + v_all[:-1]


def get_lr(global_step, initial_lr, anneal_steps):
    """ActorLearner.get_lr (agents/paac/actor_learner.py:115-119)."""
    if global_step <= anneal_steps:
        return initial_lr - (global_step * initial_lr / anneal_steps)
    return 0.0


# ----------------------------------------------------------------------------- device RNG
# NOT from the reference (which uses numpy's global MT19937): this is the CPU
# restatement of the BUILD's own counter-based generator, so on-device resets can be checked.
_PH_M0, _PH_M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_PH_W0, _PH_W1 = 0x9E3779B9, 0xBB67AE85


def philox4x32(ctr, key):
    """Philox-4x32-10.  ctr: (...,4) uint32, key: (...,2) uint32 -> (...,4) uint32."""
    c = [np.asarray(ctr[..., i], dtype=np.uint64) for i in range(4)]
    k0 = np.asarray(key[..., 0], dtype=np.uint64)
    k1 = np.asarray(key[..., 1], dtype=np.uint64)
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = _PH_M0 * c[0]
        p1 = _PH_M1 * c[2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & mask
        hi1, lo1 = p1 >> np.uint64(32), p1 & mask
        c = [(hi1 ^ c[1] ^ k0) & mask, lo1, (hi0 ^ c[3] ^ k1) & mask, lo0]
        k0 = (k0 + np.uint64(_PH_W0)) & mask
        k1 = (k1 + np.uint64(_PH_W1)) & mask
    return np.stack(c, axis=-1).astype(np.uint32)


def rng_block(seed, env_id, episode, stream, counter):
    """One Philox block: key = seed lo/hi, ctr = (counter, global env id, episode, stream).
    Mirrors golds-rl-gym_amd/csrc/rng.h:rng_block.  Broadcasts over arrays."""
    env_id = np.asarray(env_id, dtype=np.uint64)
    counter = np.asarray(counter, dtype=np.uint64)
    episode = np.asarray(episode, dtype=np.uint64)
    shape = np.broadcast(env_id, counter, episode).shape
    ctr = np.zeros(shape + (4,), dtype=np.uint32)
    ctr[..., 0] = (counter & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    ctr[..., 1] = (env_id & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    ctr[..., 2] = (episode & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    ctr[..., 3] = np.uint32(stream)
    key = np.zeros(shape + (2,), dtype=np.uint32)
    key[..., 0] = np.uint32(seed & 0xFFFFFFFF)
    key[..., 1] = np.uint32((seed >> 32) & 0xFFFFFFFF)
    return philox4x32(ctr, key)


def u01_pair(block):
    """Two float64 uniforms in [0,1) from one block: 53 high bits of each 64-bit half."""
    b = block.astype(np.uint64)
    w0 = (b[..., 1] << np.uint64(32)) | b[..., 0]
    w1 = (b[..., 3] << np.uint64(32)) | b[..., 2]
    scale = 1.0 / 9007199254740992.0
    return (w0 >> np.uint64(11)).astype(np.float64) * scale, (w1 >> np.uint64(11)).astype(np.float64) * scale


def normal_pair(block):
    """Box-Muller on (1-u0, u1): two float64 N(0,1) per block."""
    u0, u1 = u01_pair(block)
    r = np.sqrt(-2.0 * np.log(1.0 - u0))
    th = 2.0 * np.pi * u1
    return r * np.cos(th), r * np.sin(th)
