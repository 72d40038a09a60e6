"""SolowEnv / TradeAR1Env facades (reference fed_gym/envs/fed_env.py:161-334) over 1-env device engines."""
import numpy as np

from .. import _ffi

registry = {}


def register_solow_env(p, q):
    """fed_env.py:10-27: three ids per (p, q)."""
    registry["Solow-%s-%s-v0" % (p, q)] = (SolowEnv, None, dict(p=p, q=q))
    registry["Solow-%s-%s-finite-v0" % (p, q)] = (SolowEnv, 1024, dict(p=p, q=q))
    registry["Solow-%s-%s-finite-eval-v0" % (p, q)] = (SolowEnv, 1024, dict(p=p, q=q, seed=1692))


class SolowEnv(object):
    """Classic Solow model with ARMA(p,q) TFP shock and log-consumption reward (fed_env.py:161-250).
    State is float32 on the device (BASELINE north_star), observations come back as float64 arrays."""

    def __init__(self, delta=0.02, sigma=0.1, p=1, q=1, T=None, seed=None, max_episode_steps=None, device_id=0, _extra_flags=0):
        self.delta, self.sigma, self.alpha = delta, sigma, 0.33
        self.seed_value = seed
        self.T = T if T else 2048
        self.p, self.q = p, q
        flags = (_ffi.F_RESEED_EACH_RESET if seed else 0) | _extra_flags
        self._eng = _ffi.Engine(_ffi.ENV_SOLOW, 1, device_id=device_id, seed=int(seed if seed else 1692), flags=flags,
                                solow_p=p, solow_q=q, solow_tape_len=self.T + (self.T & 1), solow_sigma=sigma, solow_delta=delta,
                                max_episode_steps=int(max_episode_steps or 0))

    def _k_ss(self, savings):
        return (savings / self.delta) ** (1 / (1 - self.alpha))

    def reset(self):
        self._eng.reset()
        return self._eng.read("obs_raw")[0].astype(np.float64)

    def step(self, s):
        self._eng.step(np.array([[float(np.asarray(s).reshape(-1)[0])]], np.float32))
        obs = self._eng.read("obs_raw")[0].astype(np.float64)
        return obs, float(self._eng.read("reward")[0]), bool(self._eng.read("done")[0]), {}

    def seed(self, seed=None):
        return []


class SolowSSEnv(SolowEnv):
    """fed_env.py:253-265: sigma = 0.02, p = 1, q = 0; reset puts z = 0 (not a draw), e = 0, k = k_ss(alpha)."""

    def __init__(self, delta=0.02, sigma=0.02, T=None, max_episode_steps=None, device_id=0):
        super(SolowSSEnv, self).__init__(delta, sigma, p=1, q=0, T=T, max_episode_steps=max_episode_steps, device_id=device_id,
                                         _extra_flags=_ffi.F_SOLOW_SS_RESET)


class TradeAR1Env(object):
    """n-asset portfolio env with log-AR(1) prices (fed_env.py:268-334)."""

    def __init__(self, starting_balance=10., base_rate=0.05, n_assets=2, std_p=0.05, max_episode_steps=None, device_id=0, seed=1692):
        self.MIN_CASH = 1.
        self.starting_balance = starting_balance
        self.r = base_rate                      # kept and never used, as in the reference (fed_env.py:275)
        self.n_assets = n_assets
        self.rho_p = 0.9
        self.std_e = np.sqrt((std_p ** 2) * (1 - self.rho_p ** 2))
        self._eng = _ffi.Engine(_ffi.ENV_TRADE, 1, device_id=device_id, seed=int(seed), n_assets=n_assets, trade_std_p=std_p,
                                trade_starting_balance=float(starting_balance),
                                max_episode_steps=int(max_episode_steps or 0))

    def reset(self):
        self._eng.reset()
        return self._eng.read("obs_raw")[0].astype(np.float64)

    def step(self, action):
        a = np.asarray(action, dtype=np.float32).reshape(1, self.n_assets)
        self._eng.step_async(a)
        try:
            self._eng.wait()
        except _ffi.GrlError as e:
            if e.code == _ffi.E_ACTION_RANGE:
                raise AssertionError("action outside Box(-1, 1)")      # `assert self.action_space.contains(action)`
            raise
        obs = self._eng.read("obs_raw")[0].astype(np.float64)
        return obs, float(self._eng.read("reward")[0]), bool(self._eng.read("done")[0]), {}

    def seed(self, seed=None):
        return []


class TickerEnv(object):
    """Two-asset account trading an asset and its mirror along a historical price table with a bid/ask spread
    (fed_env.py:89-158).  action = [choices, fractions]: choices[i] in {0 hold, BUY_IDX, SELL_IDX}, fractions[i] the share
    of cash to spend (buys are rescaled to sum to at most 1) or of the position to sell.  The account lives on the
    device in float64; the env samples its 1024-row window there."""
    BUY_IDX = 1
    SELL_IDX = 2

    def __init__(self, starting_balance=10., inverse_asset=True, n_assets=2, sampler=None, ticker='IEF', max_episode_steps=None,
                 device_id=0, seed=1692, window_start=None):
        if starting_balance != 10. or n_assets != 2:
            raise NotImplementedError("only the reference defaults starting_balance=10, n_assets=2 are supported")
        from .data.sampler import OpenCloseSampler
        self.MIN_CASH = 1.
        self.n_assets = n_assets
        self.spread = 0.006
        self.data = sampler if sampler is not None else OpenCloseSampler(ticker=ticker, inverse_asset=inverse_asset)
        flags = 0 if window_start is None else _ffi.F_RESET_FROM_SNAPSHOT
        self._eng = _ffi.Engine(_ffi.ENV_TICKER, 1, device_id=device_id, seed=int(seed), flags=flags,
                                max_episode_steps=int(1023 if max_episode_steps is None else max_episode_steps))
        self._eng.ticker_set_table(self.data.data_matrix)
        if window_start is not None:
            self._eng.set_state("TICKER_START0", np.array([window_start], dtype=np.int32))

    def reset(self):
        self._eng.reset()
        return self._eng.read("obs_raw")[0].astype(np.float64)

    def step(self, action):
        choices = np.asarray(action[0]).reshape(self.n_assets)
        fractions = np.asarray(action[1], dtype=np.float64).reshape(self.n_assets)
        a = np.concatenate([choices.astype(np.float32), fractions.astype(np.float32)]).reshape(1, 4)
        self._eng.step_async(a)
        try:
            self._eng.wait()
        except _ffi.GrlError as e:
            if e.code == _ffi.E_STATE:
                raise IndexError("index 1024 is out of bounds for axis 0 with size 1024")       # price_vol_data[data_idx]
            raise
        obs = self._eng.read("obs_raw")[0].astype(np.float64)
        return obs, float(self._eng.read("reward_f64")[0]), bool(self._eng.read("done")[0]), {}

    def seed(self, seed=None):
        return []
