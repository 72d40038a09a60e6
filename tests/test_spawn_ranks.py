"""CPU: the N>1 launch path of bench.py without a launcher -- `spawn_local_ranks` starts fresh rank processes with
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, a failing rank stops the job with its exit code, and bench.py refuses a
WORLD_SIZE that contradicts --gpus.  No GPU is touched (the children are tiny scripts / the refusal happens before any import
of the engine)."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "golds-rl-gym_amd"))


def test_spawn_sets_rank_environment_and_forms_a_gloo_group(tmp_path):
    from goldsrl import distributed as D
    child = tmp_path / "child.py"
    child.write_text(
        "import os, sys\n"
        "sys.path.insert(0, %r)\n"
        "from goldsrl import distributed as D\n"
        "r = D.Ranks().init(timeout_s=120)\n"
        "assert r.world == 2 and r.local_rank == r.rank and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
        "assert r.max(1.0 + r.rank) == 2.0 and r.sum(1.0 + r.rank) == 3.0\n"
        "r.barrier()\n"
        "open(os.path.join(%r, 'rank%%d' %% r.rank), 'w').write(os.environ['WORLD_SIZE'])\n"
        "r.close()\n" % (os.path.join(ROOT, "golds-rl-gym_amd"), str(tmp_path)))
    rc = D.spawn_local_ranks([sys.executable, str(child)], 2)
    assert rc == 0
    assert sorted(p.name for p in tmp_path.iterdir() if p.name.startswith("rank")) == ["rank0", "rank1"]


def test_spawn_failing_rank_stops_the_others(tmp_path):
    from goldsrl import distributed as D
    child = tmp_path / "child.py"
    child.write_text("import os, sys, time\n"
                     "if os.environ['RANK'] == '1':\n"
                     "    sys.exit(7)\n"
                     "time.sleep(120)\n")
    t0 = time.time()
    rc = D.spawn_local_ranks([sys.executable, str(child)], 3)
    assert rc == 7
    assert time.time() - t0 < 60        # ranks 0 and 2 were terminated, not waited for


def test_bench_refuses_world_size_that_contradicts_gpus():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr and p.stdout.strip() == ""
