"""OpenCloseSampler (reference fed_gym/envs/data/sampler.py:8-41): a daily Open/Close/Volume table laid out as one
row per price print -- open, close, open, close ... -- with the mirrored ("inverse") asset beside it.

The matrix is built with numpy only (the reference's pandas-0.x indexing, sampler.py:18, does not run on pandas 2);
`sample(n)` keeps the reference's host-side contract, while the device env (envs/fed_env.py TickerEnv, csrc/ticker.hip)
takes the whole matrix once and draws each window start on the GPU."""
import csv
import os
import random

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def table_path(ticker):
    for d in (os.environ.get("GOLDSRL_DATA_DIR"), _HERE):
        if d and os.path.exists(os.path.join(d, "%s.csv" % ticker)):
            return os.path.join(d, "%s.csv" % ticker)
    raise FileNotFoundError("no price table %s.csv in $GOLDSRL_DATA_DIR or %s (columns Open, Close, Volume)" % (ticker, _HERE))


def read_table(path, columns=("Open", "Close", "Volume")):
    with open(path, newline="") as f:
        rows = list(csv.DictReader(f))
    if not rows:
        raise ValueError("%s holds no rows" % path)
    return {c: np.array([float(r[c]) for r in rows], dtype=np.float64) for c in columns}


def inverse_prices(prices):
    """Price path with the negated log returns of `prices`, starting at the same level (sampler.py:30-37)."""
    steps = np.diff(np.log(prices))
    return prices[0] * np.exp(np.concatenate([[0.0], np.cumsum(-steps)]))


def build_matrix(opens, closes, volume):
    """(2*days, 4): [price, inverse price, log(volume / first volume), same]  (sampler.py:15-28)."""
    prints = np.column_stack([opens, closes]).ravel()
    vol = np.repeat(np.asarray(volume, dtype=np.float64), 2)
    vol = np.log(vol) - np.log(vol[0])
    m = np.column_stack([prints, inverse_prices(prints), vol, vol])
    assert m[1, 0] == closes[0]
    return m


class OpenCloseSampler(object):
    def __init__(self, ticker=None, inverse_asset=True, path=None, table=None):
        """`ticker` resolves <ticker>.csv as the reference does; `path` names a CSV directly; `table` passes the three
        columns as a dict of arrays (tests, synthetic data)."""
        if table is None:
            table = read_table(path or table_path(ticker))
        self.data_matrix = build_matrix(table["Open"], table["Close"], table["Volume"])
        self.T = len(self.data_matrix)

    def sample(self, n):
        start = random.randint(0, self.T - n)
        return self.data_matrix[start:start + n]
