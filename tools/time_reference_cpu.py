#!/usr/bin/env python3
"""Times the REFERENCE's own CPU path in the build container.

Imports the unmodified reference from /root/reference under the gym/tensorflow stand-ins of tests/golden/_ref_stubs.py (the
same ones the golden fixtures were made with) and measures, per env kind,
  (a) a single-process loop of  step -> auto-reset -> process_state (-> get_local_states for Swarm), and
  (b) the reference's own Runners / GridRunners worker topology (fed_gym/agents/paac/runners.py:11-66,
      emulator_runner.py:38-151) with workers = 8 = every core of this container, random-policy actions, seed 1692.
The reference never ships to the GPU box; this script only runs here.  Output: one JSON document (profiles/r02_reference_cpu.json).

rnn_length = 1 for the Swarm runners: with a shorter history the worker calls tf.keras.preprocessing.sequence.pad_sequences
(emulator_runner.py:141-142), which needs TensorFlow; the Swarm learner does not feed the history anyway (paac.py:293,320).
That choice favours the reference (it copies one 10x84x84x3 float64 block per env-step instead of up to five)."""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests", "golden"))
import _ref_stubs  # noqa: E402

_ref_stubs.install()

import gym  # noqa: E402  (stand-in)
import fed_gym  # noqa: E402,F401
from fed_gym.agents.a3c import worker as a3c_worker  # noqa: E402
from fed_gym.agents.paac import emulator_runner, runners  # noqa: E402
from fed_gym.agents.state_processors import SolowStateProcessor, SwarmStateProcessor  # noqa: E402
from fed_gym.envs import fed_env  # noqa: E402


class Coord(object):
    def should_stop(self):
        return False


def single_swarm(steps=384):
    np.random.seed(1692)
    env = gym.make("Swarm-v0")
    sp = SwarmStateProcessor(grid_size=84)
    s = env.reset()
    emulator_runner.SwarmRunner.get_local_states(sp.process_state(s), sp.positions)
    t0 = time.perf_counter()
    resets = 0
    for _ in range(steps):
        a = emulator_runner.SwarmRunner.transform_actions_for_env(np.random.normal(size=(10, 2)).astype(np.float32))
        s, r, done, _ = env.step(a)
        if done:
            s = env.reset(); resets += 1
        emulator_runner.SwarmRunner.get_local_states(sp.process_state(s), sp.positions)
    dt = time.perf_counter() - t0
    return {"env": "Swarm-v0", "loop": "step + auto-reset + process_state(84) + get_local_states", "steps": steps, "resets": resets,
            "seconds": dt, "env_steps_per_s": steps / dt, "cores": 1}


def single_flat(kind, steps):
    np.random.seed(1692)
    if kind == "solow":
        env, sp = gym.make("Solow-v0"), SolowStateProcessor()
        proc = sp.process_state
        act = lambda: a3c_worker.sigmoid(np.random.normal(size=(1,)))[0]
        name = "Solow-v0 (p=q=1)"
    else:
        n = 16 if kind == "trade16" else 2
        env = gym.wrappers.TimeLimit(fed_env.TradeAR1Env(n_assets=n), max_episode_steps=1024)
        proc = lambda s: a3c_worker.TradeWorker.process_state(None, s) if False else np.concatenate(
            [[np.log(s[0] + 1e-4)], np.log(s[1:] + 1)])      # TradeWorker.process_state (a3c/worker.py:420-431) is an instance method of a TF worker
        act = lambda: np.tanh(np.random.normal(size=(n,)))
        name = "TradeAR1 n=%d" % n
    s = env.reset()
    t0 = time.perf_counter()
    for _ in range(steps):
        s, r, done, _ = env.step(act())
        if done:
            s = env.reset()
        proc(s)
    dt = time.perf_counter() - t0
    return {"env": name, "loop": "step + auto-reset + process_state", "steps": steps, "seconds": dt, "env_steps_per_s": steps / dt, "cores": 1}


def runners_swarm(E=32, workers=8, updates=24, G=84):
    np.random.seed(1692)
    emulators = np.asarray([gym.make("Swarm-v0") for _ in range(E)])
    sp = SwarmStateProcessor(grid_size=G)
    initial_states, idxs = [], []
    for em in emulators:                            # paac.py:245-251
        g = sp.process_state(em.reset())
        initial_states.append(emulator_runner.SwarmRunner.get_local_states(g, sp.positions)); idxs.append(sp.positions)
    initial_states = np.array(initial_states)
    variables = [initial_states, np.zeros((E, 10, 1) + initial_states.shape[-3:]), np.array(idxs), np.zeros((E, 10), np.float32),
                 np.zeros((E, 10), np.float32), np.zeros((E, 10, 2), np.float32)]      # paac.py:263-270, rnn_length = 1
    rs = runners.GridRunners(emulators, workers, variables, emulator_runner.SwarmRunner, Coord(), G)
    rs.start()
    shared = rs.get_shared_variables()
    actions = shared[-1]

    def cycle():
        a = emulator_runner.SwarmRunner.transform_actions_for_env(np.random.normal(size=(E * 10, 2)).astype(np.float32)).reshape(E, 10, 2)
        for i in range(E):
            actions[i] = a[i]
        rs.update_environments(); rs.wait_updated()
    cycle()
    t0 = time.perf_counter()
    for _ in range(updates):
        cycle()
    dt = time.perf_counter() - t0
    rs.stop()
    for r in rs.runners:
        r.join(timeout=30)
    return {"env": "Swarm-v0", "topology": "GridRunners + SwarmRunner (runners.py:57-66, emulator_runner.py:120-151)", "envs": E,
            "workers": workers, "cycles": updates, "seconds": dt, "env_steps_per_s": E * updates / dt, "cores": workers, "rnn_length": 1}


def runners_solow(E, workers=8, updates=200, rnn=5):
    np.random.seed(1692)
    emulators = np.asarray([gym.make("Solow-v0") for _ in range(E)])
    sp = SolowStateProcessor()
    states = np.array([sp.process_state(em.reset()) for em in emulators])
    # paac.py:91-97 with ONE change: the actions variable is (E,) instead of (E, 1).  A (1,)-shaped action row makes
    # SolowEnv._step build np.array([array(1,), scalar]) (fed_env.py:229), which numpy 1.13 (the reference's pin) flattens and
    # numpy >= 1.24 (this image: 2.2) rejects as ragged; scalar actions run the same arithmetic unmodified.
    variables = [states, np.zeros((E, rnn, 2)), np.zeros(E, np.float32), np.zeros(E, np.float32), np.zeros(E, np.float32)]
    rs = runners.Runners(emulators, workers, variables, emulator_runner.SolowRunner, Coord())
    rs.start()
    actions = rs.get_shared_variables()[-1]

    def cycle():
        a = emulator_runner.SolowRunner.transform_actions_for_env(np.random.normal(size=E).astype(np.float32))
        for i in range(E):
            actions[i] = a[i]
        rs.update_environments(); rs.wait_updated()
    cycle()
    t0 = time.perf_counter()
    for _ in range(updates):
        cycle()
    dt = time.perf_counter() - t0
    rs.stop()
    for r in rs.runners:
        r.join(timeout=30)
    return {"env": "Solow-v0 (p=q=1)", "topology": "Runners + SolowRunner (runners.py:11-54, emulator_runner.py:38-79)", "envs": E,
            "workers": workers, "cycles": updates, "seconds": dt, "env_steps_per_s": E * updates / dt, "cores": workers, "rnn_length": rnn}


def main():
    out = {"where": "build container, %d cores, Python %s, numpy %s; reference imported unmodified from /root/reference under "
                    "tests/golden/_ref_stubs.py" % (len(os.sched_getaffinity(0)), sys.version.split()[0], np.__version__),
           "seed": 1692, "single_process": [], "runners": []}
    def add(key, r):
        out[key].append(r)
        sys.stderr.write(json.dumps(r) + "\n"); sys.stderr.flush()
    add("single_process", single_swarm())
    add("single_process", single_flat("solow", 100000))
    add("single_process", single_flat("trade2", 50000))
    add("single_process", single_flat("trade16", 50000))
    add("runners", runners_swarm(32, 8, 24))
    add("runners", runners_swarm(256, 8, 6))
    add("runners", runners_solow(32, 8, 400))
    add("runners", runners_solow(4096, 8, 20))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
