#!/usr/bin/env python3
"""bench.py -- env-steps/s of the PAAC hot path on N MI355X GPUs of one node.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 the driver's elastic launcher starts it once per
GPU with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set.  Prints ONE JSON line on rank 0.  The host side is the standard library
and ctypes only: the ranks meet over goldsrl.distributed's own TCP store, the gradient crosses xGMI through RCCL inside the library.

A "step" is one full PAAC update of GridPAACLearner (reference paac.py:302-387) on this rank's shard
of Swarm-v0 envs (BASELINE configs[2]: 32 768 envs per GPU, 84x84x3 observation, T=20):
  T x [ConvSingleAgentPolicyNetwork forward on E*10 agent images, a = mu + sigma*N(0,1), norm clip,
       SwarmEnv.step, TimeLimit, auto-reset, process_state]  + bootstrap forward + n-step returns
  + loss/backward over the T*E*10 samples + [RCCL all-reduce of the gradient] + clip + Adam.
Everything is resident in HBM when the timed region starts; nothing crosses PCIe inside it.
`value` = env-steps/s of that full update (policy forward AND training included).  The same JSON
line also carries, measured in the same run: `rollout_only` (no training) and `env_only` (random
Gaussian policy: the env/observation/returns kernels alone).

Envs are independent, so N GPUs shard the env batch (global env ids key the generators; results do
not depend on N); the only collective is one all-reduce of the 2.2M-float gradient per update.
scaling = weak (32 768 envs per GPU = BASELINE configs[3] at N=8); for N>1 the line also carries a
`strong_scaling` point (32 768 envs in total, the shape BASELINE's target is quoted on).

N>1 without a launcher: `python bench.py --gpus N` starts N fresh rank processes itself (before anything
touches the GPU) and relays rank 0's line; under a launcher WORLD_SIZE must equal --gpus.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# The host driver of this pool supports dmabuf IPC only: without this RCCL's intra-node transport fails with
# `hipIpcGetMemHandle: invalid argument`.  The GPU boxes export it already; a launcher that scrubs the environment must not lose it.
# Set before anything loads the HIP runtime.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "golds-rl-gym_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

SWARM_BYTES_PER_ENV_STEP = 4613      # algorithmic HBM bytes of one Swarm env-step
SWARM_VALU_INSTR_PER_PAIR = 112        # counted in the ISA of swarm_kernel<MODE_STEP, exact> (hipcc 7.2, -O3 -ffp-contract=off)
VALU_LANE_INSTR_PEAK = 256 * 4 * 16 * 2.4e9
SWARM_PAIRS_PER_ENV_STEP = 7200      # pair interactions (6400 locust-locust + 800 agent-locust)
HBM_PEAK_GBS = 8000.0                # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3         # MI355X_MICROARCH.md: dense fp32 matrix peak (v_mfma_f32_32x32x2_f32)
MFMA_F16_PEAK_TFLOPS = 2516.6        # MI355X_MICROARCH.md: dense fp16 / bf16 matrix peak (v_mfma_f32_16x16x32_f16, 16 cycles each)
# the GEMMs split every fp32 operand into two fp16 terms (22-23 significand bits) and issue THREE fp16 MFMAs per fp32 product
# block (net_gemm.h), so the matrix pipe bounds the fp32-equivalent rate at 2516.6 / 3.  (Rounds 1-2 used an exact three-way
# bf16 split with SIX products: peak 419.4, same kernels otherwise -- the fraction is not comparable across that change.)
F16_PRODUCTS_PER_FP32 = 3
MFMA_X3_PEAK_TFLOPS = MFMA_F16_PEAK_TFLOPS / F16_PRODUCTS_PER_FP32
GEMM_TRAFFIC_FILE = "r05_gemm_traffic.json"   # PMC pass of the GEMM kernels (tools/run_prof.sh); refreshed per round
GEMM_TRAFFIC_FILE_INTERIOR = "r05_gemm_traffic_interior.json"   # the same pass on the interior_policy state distribution
SHARD_ENVS = 4096                             # 32 768 / 8: one GPU's share of BASELINE's target shape (strong scaling)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--envs", type=int, default=32768, help="Swarm envs per GPU")
    ap.add_argument("--T", type=int, default=20, help="max_local_steps")
    ap.add_argument("--policy", default="conv", choices=["conv", "random"],
                    help="conv: the full PAAC update (default); random: env-only rollout")
    ap.add_argument("--no-train", action="store_true", help="conv policy rollout without the gradient step")
    ap.add_argument("--fast-math", action="store_true", help="GRL_F_SWARM_FAST_MATH (not the parity default)")
    ap.add_argument("--single-stream", action="store_true",
                    help="GRL_NET_F_SINGLE_STREAM: every chunk on one stream (for rocprofv3 passes: clean per-kernel durations)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the rollout_only / env_only side measurements")
    ap.add_argument("--cpu-sample-envs", type=int, default=2048)
    ap.add_argument("--no-strong", action="store_true", help="N>1: skip the strong-scaling point (32 768 envs in total)")
    ap.add_argument("--interior", action="store_true",
                    help="profiling aid: run the main measurement on the interior_policy state distribution (agents re-injected into the "
                         "interior of the box before every update, outside the clock)")
    ap.add_argument("--no-legs", action="store_true", help="skip the dense_form / interior_policy legs (two more 32 768-env jobs)")
    ap.add_argument("--no-shard", action="store_true", help="skip the strong-scaling shard leg (the full update at 32 768 / 8 envs)")
    ap.add_argument("--no-flat-configs", action="store_true", help="skip the Solow-4096 / TradeAR1-16 side blocks (configs 2 and 5)")
    ap.add_argument("--exchange", default="rccl", choices=["rccl", "host"],
                    help="gradient exchange for N>1: RCCL all-reduce inside the library (falls back together if any rank cannot "
                         "form the communicator), or the host-store path outright")
    return ap.parse_args()


class RandomPolicyRollout(object):
    """T-step rollout with a N(0,1) Gaussian policy drawn on the device (SURVEY 8d 'env-only'):
    actions -> norm clip -> step/auto-reset/observe -> reward capture; then n-step returns."""

    def __init__(self, eng, T, gamma=0.99, scale=1000.0):
        self.eng, self.T, self.E = eng, T, eng.E
        E = self.E
        self.act = eng.dev_alloc(T * E * 20 * 4)
        self.rew = eng.dev_alloc(T * E * 4)
        self.val = eng.dev_alloc(T * E * 4)
        self.boot = eng.dev_alloc(E * 4)
        self.y = eng.dev_alloc(T * E * 4)
        self.adv = eng.dev_alloc(T * E * 4)
        self.gamma, self.scale = gamma, scale
        self.reward_ptr = eng.out_ptrs().reward
        self.counter = 0

    def run(self):
        eng, E, T = self.eng, self.E, self.T
        for t in range(T):
            a = self.act + t * E * 20 * 4
            eng.dev_randn(a, E * 20, 2, self.counter)
            self.counter += E * 10
            eng.transform_actions_device(a, E * 10)
            eng.step_device(a)
            eng.dev_copy(self.rew + t * E * 4, self.reward_ptr, E * 4)
        # GridPAACLearner form: unmasked, unclipped, adv/scale (paac.py:360-372)
        eng.returns_device(self.rew, self.val, None, self.boot, T, E, self.gamma, 1.0, self.scale, 0.0, 0.0, self.y, self.adv)


def _cpu_cores():
    ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:   # the GPU box grants a CPU share through the cgroup, not through affinity
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            ncores = max(1, min(ncores, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return min(ncores, 32)


def cpu_baseline(sample_envs, T, full_update_envs=48, mask=None):
    """CPU baseline from the oracle ("port"), bounded sample of the SAME workload: a full PAAC update
    (dense 84x84x3 images as the reference builds them, conv policy forward per step, C env step + observation,
    n-step returns, loss + backward over the T*E*10 samples) on `full_update_envs` envs; numpy/BLAS threads +
    OpenMP on the granted cores.  The env-only rate of the C port on more envs is reported beside it."""
    from oracle import nets as NN
    from oracle import oracle as O
    from oracle import oracle_c as OC
    # The baseline uses the cores the JOB was granted, not the NUMA node the rank's enqueue thread was pinned to for the GPU part
    # (advisor r4: pinned first, `allcores` silently became the node's cores): the mask from before the pin is put back for the
    # duration (threads the OpenMP runtime creates in here inherit it), the pin restored afterwards.
    pinned = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None
    if mask and pinned is not None and set(mask) != set(pinned):
        os.sched_setaffinity(0, mask)
    try:
        return _cpu_baseline(sample_envs, T, full_update_envs, NN, O, OC)
    finally:
        if mask and pinned is not None and set(mask) != set(pinned):
            os.sched_setaffinity(0, pinned)


def _cpu_baseline(sample_envs, T, full_update_envs, NN, O, OC):
    ncores = _cpu_cores()
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=ncores)
    except Exception:   # noqa: BLE001
        limiter = None
    rng = np.random.RandomState(0)
    # ---- env-only (C port, OpenMP)
    E = sample_envs
    x, xa = rng.rand(E, 80, 2), rng.rand(E, 10, 2)
    an, pn = rng.normal(size=(E, 10, 2)), rng.normal(size=(E, 80, 2))
    act = rng.normal(size=(E, 10, 2)).astype(np.float32)
    OC.swarm_step(x[:64], xa[:64], act[:64], an[:64], pn[:64], threads=1)
    env_only = {}
    for label, threads in (("1core", 1), ("allcores", ncores)):
        xx, xxa = x.copy(), xa.copy()
        t0 = time.perf_counter()
        for t in range(T):
            xx, xxa, _, _, _, _, used = OC.swarm_step(xx, xxa, act, an, pn, threads=threads)
        env_only[label] = E * T / (time.perf_counter() - t0)
    # ---- full PAAC update on a few envs
    Ef = full_update_envs
    p = NN.conv_init(3)
    x, xa = rng.rand(Ef, 80, 2), rng.rand(Ef, 10, 2)
    an, pn = rng.normal(size=(Ef, 10, 2)), rng.normal(size=(Ef, 80, 2))

    def images(x, xa):
        out = []
        for e in range(Ef):
            lb, ab, pos = O.swarm_observe_compact(x[e], xa[e], 84)
            out.append(O.swarm_local_states(O.swarm_grid_from_compact(lb, ab, 84), pos))
        return np.concatenate(out).astype(np.float32).astype(np.float64)
    t0 = time.perf_counter()
    states, actions, values, rewards = [], [], [], []
    for t in range(T):
        s = images(x, xa)
        mu, sigma, vs = NN.conv_forward(p, s, 1000.0)
        a = mu + sigma * rng.normal(size=mu.shape)
        env_a = O.swarm_transform_actions(a).reshape(Ef, 10, 2).astype(np.float32)
        x, xa, r, _, _, _, _ = OC.swarm_step(x, xa, env_a, an, pn, threads=ncores)
        states.append(s); actions.append(a); values.append(vs); rewards.append(np.repeat(r, 10))
    _, _, boot = NN.conv_forward(p, images(x, xa), 1000.0)
    y, adv = O.nstep_returns(np.array(rewards), np.array(values), boot, 0.99)
    loss, _, _, g, _ = NN.conv_loss_and_grads(p, np.concatenate(states), np.concatenate(actions), (adv / 1000.0).reshape(-1),
                                              y.reshape(-1), 0.02, 1000.0)
    gf, _ = NN.clip_by_global_norm(NN.flatten_params(g), 40.0)
    NN.adam_step(NN.flatten_params(p), gf, np.zeros_like(gf), np.zeros_like(gf), 1, 1e-4)
    dt = time.perf_counter() - t0
    if limiter is not None:
        limiter.restore_original_limits() if hasattr(limiter, "restore_original_limits") else None
    ref_note = None      # the reference's own Python path, timed in the build container (it cannot travel): BASELINE.md section 3.1
    rpath = os.path.join(ROOT, "profiles", "r02_reference_cpu.json")
    if os.path.exists(rpath):
        with open(rpath) as f:
            rj = json.load(f)
        ref_note = {"where": rj["where"], "swarm_single_process_env_steps_per_s": rj["single_process"][0]["env_steps_per_s"],
                    "swarm_gridrunners_8_workers_env_steps_per_s": {str(r["envs"]): r["env_steps_per_s"] for r in rj["runners"] if r["env"] == "Swarm-v0"},
                    "note": "unmodified reference (env step + auto-reset + process_state + get_local_states, no policy), NOT measured in this run"}
    return {"value": Ef * T / dt, "unit": "env-steps/s", "cores": ncores, "kind": "port", "reference_python_build_container": ref_note,
            "sample": "oracle full PAAC update (oracle/nets.py float64 numpy conv policy fwd+bwd+clip+Adam, oracle/oracle_c.c env "
                      "step+observation, dense 84x84x3 images as the reference feeds them) on %d envs x %d steps: %.1f s" % (Ef, T, dt),
            "env_only_value": env_only["allcores"], "env_only_value_1core": env_only["1core"],
            "env_only_sample": "oracle/oracle_c.c SwarmEnv.step + process_state on %d envs x %d steps, OpenMP" % (E, T)}


def timed(run, wait, steps):
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    wait()
    return time.perf_counter() - t0


def timed_each(run, wait, steps, before=None):
    """K updates, the clock read after each one has drained (an update ends in a stream synchronisation anyway: its statistics
    come back to the host, paac.py:189-194).  `before` runs ahead of every update, off the clock.  Returns (total seconds, [seconds
    per update])."""
    per, total = [], 0.0
    for _ in range(steps):
        if before:
            before()
        t0 = time.perf_counter()
        run()
        wait()
        per.append(time.perf_counter() - t0)
        total += per[-1]
    return total, per


def spread(per_s):
    a = sorted(per_s)
    n = len(a)
    med = a[n // 2] if n % 2 else 0.5 * (a[n // 2 - 1] + a[n // 2])
    return {"median": med * 1e3, "min": a[0] * 1e3, "max": a[-1] * 1e3, "n": n}


def flat_config_block(kind, E, T, device_id, steps=10):
    """BASELINE configs[1] (Solow-v0, 4 096 envs) / the per-GPU share of configs[4] (TradeAR1 n=16, 65 536 envs / 8 GPUs = 8 192)
    with FlatPolicyVNetwork (GRU(32) + MLP): the T-step PAAC rollout as ONE persistent kernel (a workgroup keeps 64 / 32 / 16 envs for all
    T steps: forward, sample, env step, auto-reset, bookkeeping, returns -- csrc/net_flat_rollout.inc) + the gradient step, timed
    here so the numbers are driver-visible.  These shapes are LATENCY bound: a step moves E x 53 B (Solow) / E x 481 B
    (TradeAR1-16) of env state and ~45 kMAC per sample through 2 x rnn + 5 dependent stages of one workgroup per CU; a bandwidth
    fraction would be meaningless, so the block states the per-update time, the number of dependent launches and the bytes."""
    from goldsrl import _ffi
    from goldsrl import rollout as R
    if kind == "solow":
        eng = _ffi.Engine(_ffi.ENV_SOLOW, E, device_id=device_id, seed=1692)
        bytes_per_env_step, what = 53, "Solow-v0 (p=q=1), %d envs, T=%d, FlatPolicyVNetwork (GRU(32) over 5 rows + MLP)" % (E, T)
    else:
        eng = _ffi.Engine(_ffi.ENV_TRADE, E, device_id=device_id, seed=1692, n_assets=16, rnn_length=20)
        bytes_per_env_step, what = 481, "TradeAR1-v0 n=16, %d envs (65 536 / 8 GPUs), T=%d, GRU(32) policy over 20 rows of 33" % (E, T)
    eng.reset()
    roll = R.FlatPolicyRollout(eng, T)
    out = {"workload": what}
    for label, train in (("rollout_only", False), ("value", True)):
        roll.train = train
        roll.run(); eng.wait()
        dt = timed(roll.run, eng.wait, steps) / steps
        out[label] = E * T / dt
        out["ms_per_update" if train else "ms_per_rollout"] = dt * 1e3
    out.update({"unit": "env-steps/s", "steps": steps, "dtype": "f32",
                "bound": "launch/latency",
                # rollout: done-count memset + argument upload x2 + ONE kernel; update: forward (not when the rollout kept its
                # activations: windows of >= 10 rows), a memset, weight transposes, backward, slab reduce, sum of squares, finalize, Adam
                "dependent_launches_per_update": 4 + 8,
                "rollout_keeps_activations": bool(roll.keep_activations),
                "rollout_kernel": "flat_rollout_kernel<G> (persistent: one workgroup of 16 waves per G envs for all T steps; G = 16 up to 4 096 envs, 32 up to 8 192, else 64)",
                "env_bytes_per_update": E * T * bytes_per_env_step,
                "note": "value = rollout + loss/backward/clip/Adam; %d KB of env traffic per step against ~%d us per step: not "
                        "bandwidth bound at this size" % (E * bytes_per_env_step // 1024, int(out["ms_per_rollout"] * 1e3 / (T + 1)))})
    roll.net.close(); eng.close()
    return out


def inject_interior_agents(eng, rng):
    """Put every agent into the interior of its env's observation box (the place a trained policy takes them: near the swarm) and
    refresh the observation.  The box is x in [mean_x - 1.5, mean_x + 1.5], y in [0, 6] over 84 bins (state_processors.py:17-27);
    agents go to x = mean_x(locusts) + U(-1.2, 0), y = U(1.5, 4.3): bins ~8..42 in x (the wind carries them ~28 bins to the right
    over a 20-step rollout, multiagent.py:35-36) and 21..60 in y.  Returns the share of agents whose bins are all inside 8..75."""
    x = eng.get_state("SWARM_X")
    E = x.shape[0]
    mean_x = x[:, :, 0].mean(axis=1)
    xa = np.empty((E, 10, 2))
    xa[:, :, 0] = mean_x[:, None] + rng.uniform(-1.2, 0.0, size=(E, 10))
    xa[:, :, 1] = rng.uniform(1.5, 4.3, size=(E, 10))
    eng.set_state("SWARM_XA", xa)
    eng.observe()
    eng.wait()
    pos = eng.read("positions").astype(int)
    return float(((pos >= 8) & (pos <= 75)).all(axis=2).mean())


def family_table(gemm_tags, gfam=None):
    """Per GEMM family of one single-stream profiling pass: launches, ms, achieved TFLOP/s of executed fp32 work and its fraction of the
    three-product matrix-pipe roof; with gfam (HBM-side bytes per launch of a committed PMC pass of the SAME workload and chunk size)
    also the HBM fraction of the live launch duration and which roof is nearer."""
    out = {k: {"launches": v[0], "ms": v[1], "achieved": v[2] / (v[1] * 1e-3) / 1e12 if v[1] > 0 else 0.0,
               "frac": (v[2] / (v[1] * 1e-3) / 1e12 / MFMA_X3_PEAK_TFLOPS) if v[1] > 0 else 0.0} for k, v in gemm_tags.items()}
    for k, v in out.items():
        pf = (gfam or {}).get(k)
        if pf and v["launches"] > 0 and v["ms"] > 0:
            tbps = pf["hbm_bytes_per_launch"] / (v["ms"] * 1e-3 / v["launches"]) / 1e12
            v["hbm_bytes_per_launch"] = pf["hbm_bytes_per_launch"]
            v["hbm_TBps"] = tbps
            v["hbm_frac"] = tbps / (HBM_PEAK_GBS / 1e3)
            v["bound"] = "hbm" if v["hbm_frac"] > v["frac"] else "mfma"
    return out


def load_traffic(name):
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None, {}
    with open(path) as f:
        gj = json.load(f)
    return gj.get("hbm_bytes_per_launch"), gj.get("by_family", {})


def workload_leg(args, ranks, E, T, env=None, interior=False, steps=3, traffic_file=None):
    """The same full PAAC update as `value`, on one more form of the evaluation (env: GRL_* switches read when the net is created) or
    one more state distribution (interior: agents re-injected into the interior of the box before every update, outside the timed
    region): ms per update, env-steps/s, and -- from a single-stream pass with HIP events around every GEMM -- the executed share
    of SURVEY 8(d)'s contract FLOPs and of the dense1 5x5 patch."""
    from goldsrl import _ffi
    from goldsrl import rollout as R
    saved = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    device = int(os.environ["GRL_BENCH_FORCE_DEVICE"]) if "GRL_BENCH_FORCE_DEVICE" in os.environ else ranks.local_rank
    eng = _ffi.Engine(_ffi.ENV_SWARM, E, device_id=device, seed=1692)
    eng.reset()
    roll = R.ConvPolicyRollout(eng, T, train=True)
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    rng = np.random.RandomState(4)
    shares = []

    def prepare():
        if interior:
            shares.append(inject_interior_agents(eng, rng))
    prepare()
    roll.run(); eng.wait()
    per = []
    for _ in range(steps):
        prepare()
        t0 = time.perf_counter()
        roll.run(); eng.wait()
        per.append(time.perf_counter() - t0)
    prepare()
    roll.net.profile_enable(True)
    roll.run(); eng.wait()
    launches, ms, flops = roll.net.profile_read()
    tags = roll.net.profile_read_tags()
    roll.net.profile_enable(False)
    dt = sum(per) / len(per)
    contract = 18.10e6 * 10 * E * ((T + 1) + 3 * T)
    chunk_samples = int(roll.net.cfg.max_chunk_samples)
    patch = {}
    for fam in ("dense1_patch_fwd", "dense1_patch_dgrad", "dense1_patch_wgrad"):
        if fam in tags and tags[fam][0] > 0:
            patch[fam] = tags[fam][2] / (tags[fam][0] * 2.0 * chunk_samples * 1600 * 512)
    chunks_per_step = -(-E * 10 // chunk_samples)
    nl = 4 if not os.environ.get("GRL_NET_LANES") else int(os.environ["GRL_NET_LANES"])
    out = {"ms_per_update": dt * 1e3, "ms_per_update_spread": spread(per), "value": E * T / dt, "unit": "env-steps/s", "steps": steps,
           "executed_share": flops / contract, "executed_tflop_per_update": flops / 1e12,
           "patch_support_share": patch, "gemm_ms_single_stream": ms, "gemm_tflops": flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0,
           "gemm_frac": (flops / (ms * 1e-3) / 1e12 / MFMA_X3_PEAK_TFLOPS) if ms > 0 else 0.0,
           "chunk_samples": chunk_samples, "chunks_per_step": chunks_per_step,
           # stream lanes with work: a rollout gives every chunk of a step its own T-step pipeline, the gradient step deals its
           # T x chunks_per_step chunks round-robin
           "lanes_busy": {"rollout": min(nl, chunks_per_step), "gradient_step": min(nl, T * chunks_per_step), "lanes": nl}}
    gtraffic, gfam = load_traffic(traffic_file) if traffic_file else (None, {})
    out["roofline"] = {"bound": "mfma", "achieved": out["gemm_tflops"], "peak": MFMA_X3_PEAK_TFLOPS, "unit": "TFLOP/s",
                       "frac": out["gemm_frac"], "traffic": gtraffic,
                       "traffic_source": ("from_profile: profiles/%s (PMC passes of this workload at the same chunk size)" % traffic_file)
                       if gfam else None,
                       "by_family": family_table(tags, gfam),
                       "measured": "HIP event pair around every GEMM launch of one extra single-stream update of this leg"}
    if interior:
        out["agents_inside_bins_8_75"] = sum(shares) / len(shares)
    roll.net.close(); eng.close()
    return out


def measure_swarm(args, ranks, E, T, want_roofline, label):
    """One Swarm PAAC workload on this rank's shard of E envs: warmup, K timed updates (barrier + device sync on both sides,
    max over ranks), then (rank 0, single-stream extra update) the per-GEMM HIP-event pass for the roofline."""
    from goldsrl import _ffi, sharding
    from goldsrl import distributed as D
    rank, world = ranks.rank, ranks.world
    off = sharding.env_id_offset(rank, E)
    flags = _ffi.F_SWARM_FAST_MATH if args.fast_math else 0
    device = int(os.environ["GRL_BENCH_FORCE_DEVICE"]) if "GRL_BENCH_FORCE_DEVICE" in os.environ else ranks.local_rank
    eng = _ffi.Engine(_ffi.ENV_SWARM, E, device_id=device, seed=1692, env_id_offset=off, flags=flags)
    eng.reset()
    net, exchange = None, "none"
    if args.policy == "conv":
        from goldsrl import rollout as R
        roll = R.ConvPolicyRollout(eng, T, train=not args.no_train, reserved=4 if args.single_stream else 0)
        net = roll.net
        exchange = D.attach_gradient_exchange(roll, ranks, prefer=args.exchange)
    else:
        roll = RandomPolicyRollout(eng, T)

    def barrier():
        eng.wait()
        ranks.barrier()

    for _ in range(args.warmup):
        roll.run()
    barrier()
    ht0 = net.host_times() if net is not None else None
    before = None
    if getattr(args, "interior", False):
        irng = np.random.RandomState(4)
        before = lambda: inject_interior_agents(eng, irng)      # noqa: E731
    elapsed, per_update = timed_each(roll.run, eng.wait, args.steps, before)
    ht1 = net.host_times() if net is not None else None
    ranks.barrier()
    elapsed = ranks.max(elapsed)
    host = None
    if ht0 is not None and ht1["updates"] > ht0["updates"]:
        # what the host thread of THIS rank did per update: ms inside grl_net_rollout (asynchronous: all enqueue work) and inside
        # grl_net_train_rollout up to its final stream synchronisation, and ms waiting in that synchronisation
        k = float(ht1["updates"] - ht0["updates"])
        mine = {"rank": rank, "rollout_enqueue_ms": (ht1["rollout_enqueue_ms"] - ht0["rollout_enqueue_ms"]) / k,
                "train_enqueue_ms": (ht1["train_enqueue_ms"] - ht0["train_enqueue_ms"]) / k,
                "train_wait_ms": (ht1["train_wait_ms"] - ht0["train_wait_ms"]) / k}
        mine["host_enqueue_ms_per_update"] = mine["rollout_enqueue_ms"] + mine["train_enqueue_ms"]
        host = [json.loads(x.decode()) for x in ranks.allgather_bytes(json.dumps(mine).encode())]
    # env step kernel alone (roofline_env_step): in the conv rollout every chunk's step overlaps other chunks' kernels and has no
    # clean duration, so it is timed here on whole-batch launches of a short random-policy rollout (HIP events on the handle's stream)
    probe = RandomPolicyRollout(eng, 4) if args.policy == "conv" else roll
    eng.profile_enable(True)
    probe.run(); eng.wait()
    env_launches, env_kernel_ms = eng.profile_read()
    eng.profile_enable(False)
    comm = None
    if net is not None and world > 1:
        # what RCCL itself reports (ncclCommCount) and the all-reduce time per rank (HIP events around the collective), gathered
        # to rank 0; plus the cross-rank check that the replicas still hold identical parameters after the timed updates
        info = net.comm_info()
        mine = json.dumps({"rank": rank, "rccl_ranks": info["rccl_ranks"], "rccl_user_rank": info["rccl_user_rank"],
                           "allreduce_calls": info["allreduce_calls"],
                           "allreduce_ms": info["allreduce_ms_total"] / max(info["allreduce_calls"], 1)}).encode()
        per_rank = [json.loads(x.decode()) for x in ranks.allgather_bytes(mine)]
        comm = {"rccl_ranks": min(r["rccl_ranks"] for r in per_rank), "per_rank": per_rank,
                "allreduce_ms": [r["allreduce_ms"] for r in per_rank],
                "params_equal_across_ranks": bool(D.params_equal_across_ranks(net, ranks))}
    res = {"eng": eng, "roll": roll, "net": net, "exchange": exchange, "elapsed": elapsed, "E": E, "per_update": per_update,
           "comm": comm, "host": host,
           "env_launches": env_launches, "env_kernel_ms": env_kernel_ms, "gemm": None, "gemm_step_s": None, "gemm_tags": {}}
    # GEMM roofline: one more update of the same workload with a HIP event pair around every gemm_rowk / gemm_tn launch.
    # Per-launch events need the launches serialised, so this pass runs on one stream; the timed region above alternates
    # independent chunks between four streams, where kernels of different chunks overlap and have no clean duration.
    if net is not None and want_roofline:
        net.profile_enable(True)
        res["gemm_step_s"] = timed(roll.run, eng.wait, 1)      # every rank runs it: the update contains the collective
        res["gemm"] = net.profile_read()
        res["gemm_tags"] = net.profile_read_tags()
        net.profile_enable(False)
        ranks.barrier()
    return res


def main():
    args = parse()
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # No launcher: this parent has not imported goldsrl or touched HIP; it starts N fresh rank processes and only waits.
        # Rank 0 prints the JSON line on the inherited stdout.  Any failing rank stops the job with a non-zero exit code.
        import importlib.util
        spec = importlib.util.spec_from_file_location("_grl_distributed", os.path.join(ROOT, "golds-rl-gym_amd", "goldsrl", "distributed.py"))
        D = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(D)
        rc = D.spawn_local_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus)
        if rc != 0:
            sys.stderr.write("bench.py: a rank failed (exit code %d); no result line\n" % rc)
        sys.exit(rc)

    # One process per GPU: keep this rank's host thread (and every thread the HIP runtime / RCCL start later) on the cores of its
    # GPU's NUMA node.  Found in sysfs, applied BEFORE anything touches the GPU (goldsrl/affinity.py); reported in the line.
    from goldsrl import affinity
    mask0 = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None      # what the job may use (cpu_baseline runs on it)
    local = int(os.environ.get("GRL_BENCH_FORCE_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    pin = affinity.pin_to_gpu(local)

    from goldsrl import distributed as D
    ranks = D.Ranks()
    if ranks.world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d: refusing to print a line for a job of another size" % (args.gpus, ranks.world))
    ranks.init()
    rank, world = ranks.rank, ranks.world
    E, T = args.envs, args.T

    if not pin["pinned"] and "GRL_PIN_CPUS" not in pin.get("reason", ""):
        # the KFD topology is not readable here (containers hide its GPU nodes): ask the runtime for the PCI address instead; the
        # runtime's threads exist by then, so only this (the enqueuing) thread moves
        from goldsrl import _ffi
        pin2 = affinity.pin_to_pci(_ffi.device_pci_address(local))
        pin2["sysfs_attempt"] = pin.get("reason")
        pin = pin2
    m = measure_swarm(args, ranks, E, T, want_roofline=True, label="weak")
    m["pin"] = [json.loads(x.decode()) for x in ranks.allgather_bytes(json.dumps(pin).encode())]
    eng, roll, net, exchange, elapsed = m["eng"], m["roll"], m["net"], m["exchange"], m["elapsed"]
    env_launches, env_kernel_ms, gemm, gemm_step_s, gemm_tags = m["env_launches"], m["env_kernel_ms"], m["gemm"], m["gemm_step_s"], m["gemm_tags"]

    extras = {}
    if not args.no_extras and args.policy == "conv" and world == 1:
        if not args.no_train:      # same net, same rollout, without the gradient step
            roll.train = False
            roll.run(); eng.wait()
            dt = timed(roll.run, eng.wait, 2)
            extras["rollout_only"] = {"value": E * T * 2 / dt, "unit": "env-steps/s", "steps": 2,
                                      "what": "T x (conv forward + sample + env step + observe) + bootstrap + returns, no gradient step"}
            roll.train = True
        rr = RandomPolicyRollout(eng, T)
        rr.run(); eng.wait()
        dt = timed(rr.run, eng.wait, 5)
        extras["env_only"] = {"value": E * T * 5 / dt, "unit": "env-steps/s", "steps": 5,
                              "what": "random Gaussian policy: action draw + norm clip + env step + auto-reset + process_state + returns"}
    last_stats = getattr(roll, "last_stats", None)
    # release the 32 768-env job (rollout-resident activations) before the side measurements
    if net is not None:
        net.close()
    eng.close()

    if world > 1 and not args.no_strong and args.policy == "conv" and E % world == 0:
        # strong scaling: BASELINE's target shape, 32 768 envs IN TOTAL, E / world per rank
        sargs = argparse.Namespace(**vars(args))
        sargs.warmup, sargs.steps = 2, max(3, min(args.steps, 10))
        sm = measure_swarm(sargs, ranks, E // world, T, want_roofline=False, label="strong")
        extras["strong_scaling"] = {"value": E * T * sargs.steps / sm["elapsed"], "unit": "env-steps/s", "envs_total": E,
                                    "envs_per_gpu": E // world, "steps": sargs.steps, "ms_per_step": sm["elapsed"] / sargs.steps * 1e3,
                                    "ms_per_step_spread": spread(sm["per_update"]),
                                    "allreduce_ms": None if sm["comm"] is None else sm["comm"]["allreduce_ms"],
                                    "gradient_exchange": sm["exchange"], "scaling": "strong"}
        if sm["net"] is not None:
            sm["net"].close()
        sm["eng"].close()
    if world == 1 and not args.no_extras and not args.no_legs and args.policy == "conv" and not args.no_train and not args.single_stream:
        # how much of `value` is the workload: the same update (a) without the zero-skipping forms, (b) with agents in the interior
        extras["dense_form"] = workload_leg(args, ranks, E, T, env={"GRL_PATCH_SKIP": "off", "GRL_TRUNK_SKIP": "off"})
        extras["dense_form"]["what"] = ("the same update with GRL_PATCH_SKIP=off GRL_TRUNK_SKIP=off: plain 5x5 dense1 patches, all 9 conv3 taps of "
                                        "every slot, the env-level trunk over every pixel (no row lists, no background terms, no union mask)")
        extras["interior_policy"] = workload_leg(args, ranks, E, T, interior=True, traffic_file=GEMM_TRAFFIC_FILE_INTERIOR)
        extras["interior_policy"]["what"] = ("the same update (default zero-skipping forms) on a state distribution with the agents INSIDE the "
                                             "observation box -- positions injected before every update, outside the timed region: x = "
                                             "mean_x + U(-1.2, 0), y = U(1.5, 4.3) -- where an agent's conv3 footprint is ~75 % of its 5x5 "
                                             "patch instead of ~25 % at the rim; `value` above stays on SURVEY 8(d)'s prescribed workload "
                                             "(random-init policy from the env's own reset states)")
    if world == 1 and not args.no_extras and not args.no_shard and args.policy == "conv" and not args.no_train and not args.single_stream:
        # the per-GPU share of the shape BASELINE's target is quoted on (32 768 Swarm envs over 8 GPUs, >= 6x of the 1-GPU rate): the
        # same full update at 4 096 envs.  On one GPU this prices everything of that point except the 8.84 MB all-reduce.
        sh = workload_leg(args, ranks, SHARD_ENVS, T, steps=7)
        sh["ms_per_update"] = sh["ms_per_update_spread"]["median"]
        sh["value"] = SHARD_ENVS * T / (sh["ms_per_update"] * 1e-3)
        one = elapsed / args.steps * 1e3 * (32768.0 / E)
        sh["what"] = ("the same full PAAC update at %d envs = 32 768 / 8, median of %d: one rank's work at the strong-scaling point of "
                      "BASELINE's target" % (SHARD_ENVS, sh["steps"]))
        sh["projected_8gpu"] = {"measured": False,
                                "speedup_1_to_8": one / sh["ms_per_update"],
                                "env_steps_per_s": 32768 * T / (sh["ms_per_update"] * 1e-3),
                                "note": "PROJECTED, not measured: ms_per_step of this run (scaled to 32 768 envs) / this leg's median; "
                                        "it leaves out the all-reduce of the 2.2 M-float gradient (8.84 MB per rank and update over "
                                        "xGMI) and the ranks' mutual wait -- RCCL with world size > 1 has not run on any box yet"}
        extras["shard_%d" % SHARD_ENVS] = sh
    if world == 1 and not args.no_flat_configs and not args.no_extras and args.policy == "conv":
        device = int(os.environ.get("GRL_BENCH_FORCE_DEVICE", ranks.local_rank))
        extras["solow_4096"] = flat_config_block("solow", 4096, T, device)
        extras["trade16_gru_8192"] = flat_config_block("trade", 8192, T, device)

    if rank == 0:
        total_env_steps = world * E * T * args.steps
        value = total_env_steps / elapsed
        k_avg_ms = env_kernel_ms / max(env_launches, 1)
        env_achieved = E * SWARM_BYTES_PER_ENV_STEP / (k_avg_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "swarm_step_traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as f:
                tj = json.load(f)
            if tj.get("envs") == E and bool(tj.get("fast_math")) == bool(args.fast_math):
                traffic = tj.get("hbm_bytes_per_launch")
        env_roof = {"bound": "hbm", "kernel": "swarm_kernel<MODE_STEP>", "achieved": env_achieved, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": env_achieved / HBM_PEAK_GBS, "traffic": traffic, "avg_kernel_ms": k_avg_ms,
                    "launches": env_launches, "pair_interactions_per_s": E * SWARM_PAIRS_PER_ENV_STEP / (k_avg_ms * 1e-3),
                    "note": "fp64-VALU/transcendental bound (~90 flop/B), not HBM bound: see DESIGN.md",
                    # the bound that applies: the pair loop is 112 VALU instructions per pair in the gfx950 ISA (67 full-rate
                    # fp64 fma/mul/add, 3 divisions = 6 div_scale + 3 rcp + 3 div_fmas + 3 div_fixup, 2 exp, 1 rsq, selects);
                    # peak = 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz full-rate issue (rcp/rsq are quarter rate, so 100 % is not reachable)
                    "valu_f64_issue": {"instr_per_pair": SWARM_VALU_INSTR_PER_PAIR,
                                       "achieved": E * SWARM_PAIRS_PER_ENV_STEP * SWARM_VALU_INSTR_PER_PAIR / (k_avg_ms * 1e-3),
                                       "peak": VALU_LANE_INSTR_PEAK, "unit": "lane-instructions/s",
                                       "frac": E * SWARM_PAIRS_PER_ENV_STEP * SWARM_VALU_INSTR_PER_PAIR / (k_avg_ms * 1e-3) / VALU_LANE_INSTR_PEAK}}
        stages = "action draw + norm clip + SwarmEnv.step + TimeLimit + auto-reset + process_state + n-step returns"
        if args.policy == "conv":
            stages = ("conv policy forward (fp32) + Gaussian sample + norm clip + SwarmEnv.step + TimeLimit + auto-reset + "
                      "process_state + bootstrap + n-step returns" + ("" if args.no_train else
                      " + loss/backward over T*E*10 samples + gradient all-reduce + clip_by_global_norm + Adam"))
        out = {
            "metric": "env-steps/sec (whole node), 32k parallel Swarm-v0 envs, 20-step PAAC rollout",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "ms_per_step_spread": spread(m["per_update"]),
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.policy == "conv" else "f64", "data": "synthetic",
            "config": {"workload": "Swarm-v0 84x84, %d envs per GPU, T=%d PAAC update, conv policy of train_paac_conv.py "
                                   "(BASELINE configs[2]%s)" % (E, T, "" if world == 1 else "; %d envs over %d GPUs = configs[3] at N=8" % (E * world, world)),
                       "envs_per_gpu": E, "envs_total": E * world, "rollout_steps": T, "policy": args.policy,
                       "train": args.policy == "conv" and not args.no_train,
                       "swarm_math": "fast" if args.fast_math else "exact", "env_state_dtype": "f64", "stages": stages,
                       "streams": 1 if (args.single_stream or args.policy != "conv") else 4,
                       "gradient_exchange": exchange},
        }
        out["config"]["cpu_affinity"] = m["pin"][0] if world == 1 else m["pin"]
        if m["host"] is not None:
            # max over ranks: the slowest host thread is what a lock-step job waits for
            out["host_enqueue_ms_per_update"] = max(h["host_enqueue_ms_per_update"] for h in m["host"])
            out["host"] = {"per_rank": m["host"],
                           "note": "wall-clock ms of the rank's host thread inside grl_net_rollout (asynchronous) + inside "
                                   "grl_net_train_rollout before its final stream synchronisation, per update; train_wait_ms = inside "
                                   "that synchronisation.  Enqueue calls block when a stream's launch queue is full, so enqueue ~ "
                                   "ms_per_step means 'the host keeps the queues full', not 'the host is the bottleneck': the "
                                   "host-bound signature is train_wait_ms ~ 0"}
        if m["comm"] is not None:
            out["rccl_ranks"] = m["comm"]["rccl_ranks"]
            out["gradient_exchange"] = exchange
            out["allreduce_ms"] = m["comm"]["allreduce_ms"]
            out["comm"] = m["comm"]
        if gemm is not None:
            launches, ms, flops = gemm
            ach = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            gtraffic, gfam = load_traffic(GEMM_TRAFFIC_FILE)      # HBM bytes per GEMM launch from the committed PMC passes (same 81 920-sample chunks)
            out["roofline"] = {"bound": "mfma",
                               "kernel": "gemm_rowk / gemm_tn (fp32 implicit GEMMs: operands split into 2 fp16 terms, "
                                         "3 v_mfma_f32_16x16x32_f16 per 16x16 tile and K=32 step, fp32 accumulation)",
                               "achieved": ach, "peak": MFMA_X3_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_X3_PEAK_TFLOPS,
                               "peak_note": "achieved counts ALGORITHMIC fp32 FLOPs (2*M*N*K of the work actually executed: the dense1 patch "
                                            "and conv3 slot GEMMs skip operand tiles that are exact zeros -- patch pixels / taps outside an "
                                            "agent's conv3 footprint -- and the env-level conv2 / conv3 GEMMs run over the rows the env's "
                                            "bins reach, the constant background rows entering in closed form; all counted by their masks / "
                                            "live row counts, so the fraction fell from 0.24 when the GEMM time shrank with the FLOPs that "
                                            "ran fastest; `contract` prices the same update by SURVEY 8(d)'s per-agent figure); peak = "
                                            "dense fp16 MFMA peak %.1f / %d fp16 products per fp32 product (round 2's six-product bf16 "
                                            "form had peak %.1f: halving the MFMA count doubled the peak this fraction is taken of).  "
                                            "The executed fp16 MFMA rate is %d x achieved; the fp32 MFMA peak "
                                            "(v_mfma_f32_32x32x2_f32) would be %.1f" % (
                                                MFMA_F16_PEAK_TFLOPS, F16_PRODUCTS_PER_FP32, MFMA_F16_PEAK_TFLOPS / 6, F16_PRODUCTS_PER_FP32,
                                                MFMA_F32_PEAK_TFLOPS),
                               "executed_f16_tflops": ach * F16_PRODUCTS_PER_FP32,
                               "frac_of_round1_six_product_peak": ach / (MFMA_F16_PEAK_TFLOPS / 6.0),
                               "vs_fp32_mfma_peak": ach / MFMA_F32_PEAK_TFLOPS,
                               "traffic": gtraffic,
                               "traffic_source": "from_profile: profiles/%s (rocprofv3 PMC passes of the same 81 920-sample chunks; "
                                                 "not collected inside this run)" % GEMM_TRAFFIC_FILE,
                               "launches": launches, "gemm_ms_total": ms,
                               "gemm_share_of_step": (ms * 1e-3) / gemm_step_s if gemm_step_s else None,
                               "measured": "HIP event pair around every GEMM launch of ONE extra update run right after the timed "
                                           "region on a single stream (%.1f ms); the timed region itself deals chunks round-robin "
                                           "to four streams (GRL_NET_F_SINGLE_STREAM off)" % (gemm_step_s * 1e3 if gemm_step_s else 0.0),
                               "flops_per_launch_avg": flops / max(launches, 1),
                               # which roof a family is nearer: the live duration of its launches against the matrix pipe (frac) and
                               # against HBM (hbm_frac: HBM-side bytes per launch of the committed PMC passes of the same chunk size /
                               # the live average launch duration / 8 TB/s); the N <= 64 families (conv gathers, slots, class
                               # corrections) are byte bound, the square dense ones sit under both roofs
                               "by_family": family_table(gemm_tags, gfam)}
            for v in out["roofline"]["by_family"].values():
                if "hbm_frac" in v:
                    v["traffic_source"] = "from_profile: profiles/%s" % GEMM_TRAFFIC_FILE
            # SURVEY 8(d)'s algorithmic figure: the reference evaluates the net once per agent-sample, 18.10 MFLOP forward, backward = 2x
            # forward; an update = (T + 1) rollout forwards + T training forwards and backwards over E * 10 agent-samples
            contract_flops = 18.10e6 * 10 * E * ((T + 1) + (0 if args.no_train else 3 * T))
            out["roofline"]["contract"] = {
                "flops_per_agent_sample_forward": 18.10e6, "flops_per_update": contract_flops,
                "tflops_over_the_timed_update": contract_flops / (elapsed / args.steps) / 1e12,
                "executed_share": (flops / contract_flops) if contract_flops > 0 else None,
                "note": "what the per-agent evaluation of the reference would have to execute for the same update, over the WHOLE timed "
                        "step (helpers and env step included); the shared-trunk / slot / patch / background evaluation computes the same "
                        "function with executed_share of those FLOPs -- this is a statement about the algorithm, not about the matrix pipe"}
            # what the support masks leave of the dense1 patch GEMMs: executed FLOPs against the plain 5x5 patch (2 * samples * 1600 * 512
            # per launch of a chunk of min(E * 10, 81 920) samples)
            chunk_samples = min(E * 10, 81920)
            for fam in ("dense1_patch_fwd", "dense1_patch_dgrad", "dense1_patch_wgrad"):
                if fam in gemm_tags and gemm_tags[fam][0] > 0:
                    out["roofline"]["by_family"][fam]["executed_share_of_the_5x5_patch"] = \
                        gemm_tags[fam][2] / (gemm_tags[fam][0] * 2.0 * chunk_samples * 1600 * 512)
            out["roofline_env_step"] = env_roof
        else:
            out["roofline"] = env_roof
        out.update(extras)
        if last_stats:
            out["last_update_stats"] = last_stats
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample_envs, T, mask=mask0)
            out["cpu_baseline"]["affinity"] = "the job's own mask (%s cpus), not the GPU rank's NUMA pin" % (len(mask0) if mask0 else "?")
        print(json.dumps(out))
        sys.stdout.flush()
    ranks.barrier()
    ranks.close()


if __name__ == "__main__":
    main()
