"""A/B of the flat PAAC rollout: ONE persistent kernel (default) vs the hipGraph of launches (GRL_FLAT_ROLLOUT=graph), BASELINE
configs[1] (Solow 4 096 envs) and the per-GPU share of configs[4] (TradeAR1-16, 8 192 envs).  Usage: python tools/bench_flat_modes.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "golds-rl-gym_amd"))
import bench  # noqa: E402

for mode in ("persistent", "graph"):
    if mode == "graph":
        os.environ["GRL_FLAT_ROLLOUT"] = "graph"
    else:
        os.environ.pop("GRL_FLAT_ROLLOUT", None)
    for kind, E in (("solow", 4096), ("trade", 8192), ("trade", 65536)):
        r = bench.flat_config_block(kind, E, 20, 0, steps=20)
        print(json.dumps({"mode": mode, "kind": kind, "E": E, "ms_per_rollout": round(r["ms_per_rollout"], 4),
                          "ms_per_update": round(r["ms_per_update"], 4)}))
        sys.stdout.flush()
