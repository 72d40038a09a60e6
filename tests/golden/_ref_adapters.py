"""numpy-2 adapter used ONLY by gen_golden_learner.py (build container).

The reference's learner hands SolowEnv._step one ROW of its (E, num_actions) shared action array, i.e. a (1,)-shaped
float32 array (paac.py:96,127-128; emulator_runner.py:48-49).  Under the reference's numpy 1.13 that array flows through
_step (fed_env.py:201-236) as a one-element array; under this image's numpy 2.2 `np.array([self.k, z_next])` (fed_env.py:229)
rejects the ragged pair with an ordinary ValueError.  The subclass below unwraps the single element and calls the reference's
_step unchanged -- same values, no arithmetic of its own."""
from fed_gym.envs import fed_env


class SolowEnvRowAction(fed_env.SolowEnv):
    def _step(self, s):
        return super(SolowEnvRowAction, self)._step(s[0])
