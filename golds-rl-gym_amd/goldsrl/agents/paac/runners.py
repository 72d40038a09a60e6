"""Runners / GridRunners: the drop-in boundary (reference fed_gym/agents/paac/runners.py:7-66).

Same constructor and method set.  `emulators` are facade envs (goldsrl.envs.*) that the learner has
already reset; their state is moved into ONE batch engine of len(emulators) envs.  `workers` is accepted
and ignored (there are no processes).  `variables` keep the reference's list layout; after
wait_updated() the numpy arrays hold the new states / rewards / dones exactly where the reference's
shared-memory arrays would.  Step protocol: the learner writes actions IN PLACE into variables[-1],
calls update_environments() (enqueue on the HIP stream) and wait_updated() (stream sync + copy-back)."""
import numpy as np

from ... import _ffi
from .emulator_runner import SwarmRunner


class Runners(object):
    def __init__(self, emulators, workers, variables, emulator_class, coord):
        if len(emulators) % max(int(workers), 1) != 0:
            # np.split(emulators, workers) in the reference raises for an unequal division
            raise ValueError("array split does not result in an equal division")
        self.variables = [np.array(v) for v in variables]
        self.workers = workers
        self.coord = coord
        self.emulator_class = emulator_class
        self.emulators = emulators
        self.E = len(emulators)
        self._stopped = False
        self._build_engine()

    # -- flat (Solow) engine: state is taken over from the facade envs
    def _build_engine(self):
        e0 = self.emulators[0]._eng
        c = e0.cfg
        self.engine = _ffi.Engine(_ffi.ENV_SOLOW, self.E, device_id=c.device_id, seed=c.seed, flags=c.flags, solow_p=c.solow_p,
                                  solow_q=c.solow_q, solow_tape_len=c.solow_tape_len, solow_sigma=c.solow_sigma,
                                  solow_delta=c.solow_delta, max_episode_steps=c.max_episode_steps,
                                  rnn_length=self.variables[1].shape[1])
        self.engine.reset()
        for f in ("SOLOW_K", "SOLOW_Z", "SOLOW_E", "SOLOW_TAPE", "SOLOW_TAPE_POS", "ELAPSED"):
            self.engine.set_state(f, np.concatenate([em._eng.get_state(f) for em in self.emulators]))
        self.engine.observe()

    def start(self):
        pass

    def stop(self):
        self._stopped = True

    def get_shared_variables(self):
        return self.variables

    def update_environments(self):
        if self.coord is not None and self.coord.should_stop():
            self.stop()
            return
        self.engine.step_async(self.variables[-1])

    def wait_updated(self):
        if self._stopped:
            return
        self.engine.wait()
        self.variables[0][...] = self.engine.read("obs")
        self.variables[1][...] = self.engine.read("history")
        self.variables[2][...] = self.engine.read("reward")
        self.variables[3][...] = self.engine.read("done").astype(np.float32)


class GridRunners(Runners):
    """variables = [states (E,10,G,G,3), histories, positions (E,10,2), rewards (E,10), episode_over (E,10),
    actions (E,10,2)] (paac.py:263-270).  The dense `states` slot is only materialised when it was passed in
    as an array (small E, compat); pass None to keep observations compact on the device."""

    def __init__(self, emulators, workers, variables, emulator_class, coord, grid_size):
        self.grid_size = grid_size
        self._dense = variables[0] is not None
        # the HISTORY slot (E,10,rnn,G,G,3) is dead in the reference's Swarm learner (its feeds are commented out, paac.py:293,
        # 320,355,376) but it is a slot of the boundary: filled when the caller passes the array (needs the dense states too)
        self._hist = self._dense and variables[1] is not None and np.ndim(variables[1]) == 6
        vs = list(variables)
        if vs[0] is None:
            vs[0] = np.zeros((0,))
        if vs[1] is None:
            vs[1] = np.zeros((0,))
        super().__init__(emulators, workers, vs, emulator_class, coord)
        self._nhist = np.zeros(self.E, np.int64)      # len(self.histories[i]) of the worker (emulator_runner.py:23): empty at start

    def _build_engine(self):
        c = self.emulators[0]._eng.cfg
        # SEEDED emulators (SwarmEnv(seed=n), e.g. Swarm-eval-v0) replay the reference's own episode after every reset: the facade
        # drew x, xa, the burn-in actions and the noise tables from MT19937(n) (multiagent.py:46-56) and the learner has just
        # reset it (paac.py:247-251), so its current state IS the reset state -- kept as the batch engine's reset snapshot
        seeded = all(getattr(em, "n_seed", None) for em in self.emulators)
        fresh = all(int(em._eng.get_state("ELAPSED")[0]) == 0 for em in self.emulators)
        flags = c.flags | (_ffi.F_RESET_FROM_SNAPSHOT if (seeded and fresh) else 0)
        self.engine = _ffi.Engine(_ffi.ENV_SWARM, self.E, device_id=c.device_id, seed=c.seed, flags=flags,
                                  grid_size=self.grid_size, max_episode_steps=c.max_episode_steps)
        for f in ("SWARM_X", "SWARM_XA", "SWARM_PNOISE", "SWARM_ANOISE", "ELAPSED", "EPISODE"):
            v = np.concatenate([em._eng.get_state(f) for em in self.emulators])
            self.engine.set_state(f, v)
            if seeded and fresh and f.startswith("SWARM_"):
                self.engine.set_state("RESET_" + f[len("SWARM_"):], v)
        self.engine.observe()

    def wait_updated(self):
        if self._stopped:
            return
        self.engine.wait()
        done = self.engine.read("done")
        if self._dense:
            st = self.variables[SwarmRunner.STATE_IDX]
            d = self.engine.materialize_states()
            if st.dtype == np.float64:
                # the reference's STATE slot is float64 (quirk Q7) with the densities count/80 and count/10 (state_processors.py:
                # 29-42): the device's float32 image holds the same counts, from which the float64 quotients are exact
                st[..., 0] = np.rint(d[..., 0].astype(np.float64) * 80.0) / 80.0
                st[..., 1] = np.rint(d[..., 1].astype(np.float64) * 10.0) / 10.0
                st[..., 2] = d[..., 2]
            else:
                st[...] = d
        if self._hist:
            # SwarmRunner._run's window (emulator_runner.py:129-145) AS IT BEHAVES: the list entries are views of the env's row in
            # the shared STATE array (quirk Q11), so the window is min(n, rnn) copies of the CURRENT state; while n < rnn the worker
            # pads through keras' pad_sequences with its default dtype int32, which truncates the densities toward zero
            # (pinned by tests/golden/swarm_runner.npz; pad_sequences itself is third party, restated)
            H = self.variables[SwarmRunner.HISTORY_IDX]
            rnn = H.shape[2]
            self._nhist = np.where(done != 0, 1, np.minimum(self._nhist + 1, rnn + 1))
            st = self.variables[SwarmRunner.STATE_IDX]
            H[...] = 0
            for e in range(self.E):
                n = int(min(self._nhist[e], rnn))
                cur = st[e] if self._nhist[e] >= rnn else np.trunc(st[e])
                H[e, :, :n] = cur[:, None]
        self.variables[SwarmRunner.AGENT_POSITIONS_IDX][...] = self.engine.read("positions")
        # scalar reward / done broadcast over the 10 agent columns (emulator_runner.py:146-147)
        self.variables[SwarmRunner.REWARD_IDX][...] = self.engine.read("reward")[:, None]
        self.variables[SwarmRunner.DONE_IDX][...] = done.astype(np.float32)[:, None]
