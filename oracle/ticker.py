"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT PATH.

Batched numpy (float64) restatement of the reference's Ticker path: the price-table
sampler, TickerEnv and its state processor.  Only tests/ may import this module, and only as the checker.

Pinning: tests/test_oracle_golden.py::test_ticker_* check every function against tests/golden/ticker.npz, captured
from the UNMODIFIED reference by tests/golden/gen_golden_ticker.py (which documents the two stand-ins it needs: the
sampler is pandas-0.x code, and the env's constructor reads a CSV).  The one input the reference draws and a device
cannot replay, `random.randint` for the window start (envs/data/sampler.py:38), is an explicit argument here.

All paths cited are relative to /root/reference/.
"""
import numpy as np

SPREAD = 0.006            # envs/fed_env.py:105 ("6 basis points" in the comment, 0.6 % in the arithmetic)
MIN_CASH = 1.0            # envs/fed_env.py:96
STARTING_BALANCE = 10.0   # envs/fed_env.py:93
WINDOW = 1024             # envs/fed_env.py:147
BUY, SELL = 1, 2          # envs/fed_env.py:90-91


def get_inverse(prices):
    """OpenCloseSampler._get_inverse (envs/data/sampler.py:30-37): the price path whose log returns are the negatives."""
    returns = np.log(prices[1:]) - np.log(prices[:-1])
    cum_neg = np.concatenate([[0.0], np.cumsum(-returns)])
    return prices[0] * np.exp(cum_neg)


def open_close_to_sequence(opens, closes, volume, inverse_asset=True):
    """OpenCloseSampler.open_close_to_sequence (envs/data/sampler.py:15-28): rows alternate open, close of each day;
    columns [price, inverse price, log volume ratio, log volume ratio].  (`inverse_asset` is accepted and ignored by
    the reference as well.)"""
    joined = np.stack([np.asarray(opens, np.float64), np.asarray(closes, np.float64)], axis=1).reshape(-1)
    inverse = get_inverse(joined)
    vol = np.stack([np.asarray(volume, np.float64)] * 2, axis=1).reshape(-1)
    vol = np.log(vol) - np.log(vol[0])
    return np.stack([joined, inverse, vol, vol], axis=1)


def ticker_reset(matrix, start):
    """TickerEnv._reset (envs/fed_env.py:144-156) for a batch of window starts.
    Returns state dict and the (E,7) observation [cash, q0, q1, p0, p1, v0, v1]."""
    start = np.asarray(start, dtype=np.int64)
    E = start.shape[0]
    st = {"cash": np.full(E, STARTING_BALANCE), "assets": np.full(E, STARTING_BALANCE), "qty": np.zeros((E, 2)),
          "idx": np.zeros(E, dtype=np.int64), "start": start.copy()}
    return st, ticker_obs(matrix, st)


def ticker_obs(matrix, st):
    row = matrix[st["start"] + st["idx"]]
    return np.concatenate([st["cash"][:, None], st["qty"], row[:, :2], row[:, 2:]], axis=1)


def ticker_step(matrix, st, disc, cont):
    """TickerEnv._step (envs/fed_env.py:110-142), batched over envs, same operation order.  disc (E,2) in {0,1,2},
    cont (E,2) fractions.  Updates `st` in place; returns obs (E,7), reward (E,), done (E,).
    Like the reference it does not reset a finished env and does not check the end of the 1024-row window."""
    disc = np.asarray(disc)
    cont = np.array(cont, dtype=np.float64)                    # the reference rescales the caller's array in place
    prices = matrix[st["start"] + st["idx"]][:, :2]
    buy, sell = disc == BUY, disc == SELL
    bsum = np.where(buy[:, 0], cont[:, 0], 0.0) + np.where(buy[:, 1], cont[:, 1], 0.0)
    # `.sum()` of the masked entries: one term, or c0 + c1
    bsum = np.where(buy[:, 0] & buy[:, 1], cont[:, 0] + cont[:, 1], np.where(buy[:, 0], cont[:, 0], np.where(buy[:, 1], cont[:, 1], 0.0)))
    denom = np.maximum(bsum, 1.0)
    cont = np.where(buy, cont / denom[:, None], cont)
    q_buy = cont * st["cash"][:, None] / (prices * (1 + SPREAD))
    q_sell = -cont * st["qty"]
    q_add = np.where(buy, q_buy, np.where(sell, q_sell, 0.0))
    st["qty"] = st["qty"] + q_add
    cb = q_add * prices * (1 + SPREAD)
    cs = q_add * (prices * (1 - SPREAD))

    def msum(v, m):      # numpy .sum() over the selected entries, in index order
        return np.where(m[:, 0] & m[:, 1], v[:, 0] + v[:, 1], np.where(m[:, 0], v[:, 0], np.where(m[:, 1], v[:, 1], 0.0)))

    st["cash"] = st["cash"] + (-msum(cb, buy) - msum(cs, sell))
    old = st["assets"]
    st["assets"] = st["cash"] + (st["qty"][:, 0] * prices[:, 0] + st["qty"][:, 1] * prices[:, 1])
    done = st["assets"] < MIN_CASH
    st["idx"] = st["idx"] + 1
    reward = np.log(st["assets"] + 1e-4) - np.log(old + 1e-4)
    return ticker_obs(matrix, st), reward, done


def ticker_process_state(raw, n_assets=2):
    """TickerTraderStateProcessor.process_state (agents/state_processors.py:50-63):
    [log(cash+1e-4), log(q+1)..., log(p)..., volumes...]."""
    raw = np.asarray(raw, np.float64)
    n = n_assets
    return np.concatenate([np.log(raw[..., :1] + 1e-4), np.log(raw[..., 1:1 + n] + 1), np.log(raw[..., 1 + n:1 + 2 * n]),
                           raw[..., 1 + 2 * n:]], axis=-1)


def ticker_process_temporal_states(history, n_assets=2):
    """TickerTraderStateProcessor.process_temporal_states (agents/state_processors.py:65-66): prices and volumes only."""
    return np.vstack(history)[:, 1 + n_assets:]


def ticker_transform_raw_action(disc, cont):
    """TickerGatedTraderWorker.transform_raw_action (agents/a3c/worker.py:491-494): choices pass, fractions = sigmoid."""
    cont = np.asarray(cont, np.float64)
    return np.asarray(disc), 1.0 / (1.0 + np.exp(-cont))
