"""The two flat-net legs of bench.py on their own (rollout and full update, ms): python tools/flat_update_times.py [repeats]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "golds-rl-gym_amd"))
import bench  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for kind, E in (("solow", 4096), ("trade", 8192)):
    rows = [bench.flat_config_block(kind, E, 20, 0) for _ in range(reps)]
    print(kind, E, "ms_per_rollout", ["%.3f" % r["ms_per_rollout"] for r in rows], "ms_per_update", ["%.3f" % r["ms_per_update"] for r in rows], flush=True)
