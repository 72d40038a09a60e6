"""CPU: the N>1 host layer (goldsrl/distributed.py: standard library only) and the launch path of bench.py without a launcher -- `spawn_local_ranks` starts fresh rank processes with
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, a failing rank stops the job with its exit code, and bench.py refuses a
WORLD_SIZE that contradicts --gpus.  No GPU is touched (the children are tiny scripts / the refusal happens before any import
of the engine)."""
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "golds-rl-gym_amd"))


def test_spawn_sets_rank_environment_and_the_ranks_meet_over_the_store(tmp_path):
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


_STORE_CHILD = (
    "import os, sys, struct\n"
    "import numpy as np\n"
    "sys.path.insert(0, %r)\n"
    "from goldsrl import distributed as D\n"
    "assert 'torch' not in sys.modules\n"
    "r = D.Ranks().init(timeout_s=120)\n"
    "w = r.world\n"
    "assert r.max(1.0 + r.rank) == float(w) and r.min(1.0 + r.rank) == 1.0 and r.sum(1.0 + r.rank) == w * (w + 1) / 2\n"
    "parts = r.allgather_bytes(bytes([r.rank]) * (r.rank + 1))\n"
    "assert parts == [bytes([k]) * (k + 1) for k in range(w)]\n"
    "uid = np.arange(128, dtype=np.uint8) if r.rank == 0 else np.zeros(128, np.uint8)\n"
    "assert np.array_equal(np.frombuffer(r.broadcast_bytes(uid.tobytes(), 0), np.uint8), np.arange(128, dtype=np.uint8))\n"
    "g = np.full(2210213, 0.25 * (r.rank + 1), np.float32)\n"          # the conv net's flat gradient size
    "s = r.allreduce_sum_f32(g)\n"
    "assert s.shape == g.shape and (s == 0.25 * w * (w + 1) / 2).all()\n"
    "p = r.broadcast_array(np.arange(7, dtype=np.float32) * (1 + r.rank), 0)\n"
    "assert np.array_equal(p, np.arange(7, dtype=np.float32))\n"
    "r.barrier()\n"
    "assert 'torch' not in sys.modules\n"
    "open(os.path.join(%r, 'ok%%d' %% r.rank), 'w').write('ok')\n"
    "r.close()\n")


def test_store_collectives_world_3(tmp_path):
    from goldsrl import distributed as D
    child = tmp_path / "child.py"
    child.write_text(_STORE_CHILD % (os.path.join(ROOT, "golds-rl-gym_amd"), str(tmp_path)))
    assert D.spawn_local_ranks([sys.executable, str(child)], 3) == 0
    assert sorted(p.name for p in tmp_path.iterdir() if p.name.startswith("ok")) == ["ok0", "ok1", "ok2"]


def test_store_on_a_fixed_port(tmp_path):
    """GRL_STORE_PORT: the multi-node form (every rank connects to MASTER_ADDR on a given port, no rendezvous file)."""
    import socket
    from goldsrl import distributed as D
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    child = tmp_path / "child.py"
    child.write_text(_STORE_CHILD % (os.path.join(ROOT, "golds-rl-gym_amd"), str(tmp_path)))
    assert D.spawn_local_ranks([sys.executable, str(child)], 2, extra_env={"GRL_STORE_PORT": str(port)}) == 0
    assert sorted(p.name for p in tmp_path.iterdir() if p.name.startswith("ok")) == ["ok0", "ok1"]


def test_store_under_the_drivers_elastic_launcher(tmp_path):
    """The driver starts bench.py's ranks with `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P`: MASTER_PORT is then held by the launcher's own store, and the ranks (which never import it)
    must still meet -- through the ephemeral port rank 0 publishes for its siblings."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    child = tmp_path / "child.py"
    child.write_text(_STORE_CHILD % (os.path.join(ROOT, "golds-rl-gym_amd"), str(tmp_path)))
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "GRL_RDZV_KEY", "GRL_STORE_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(child)], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    assert sorted(q.name for q in tmp_path.iterdir() if q.name.startswith("ok")) == ["ok0", "ok1"]


class _FakeNet(object):
    """Stands for a ConvNet / FlatNet binding in the phase logic of attach_gradient_exchange (no GPU here)."""

    class _Lib(object):
        @staticmethod
        def grl_comm_unique_id_bytes():
            return 128

    def __init__(self, rank, fail_init_on):
        import numpy as np
        self.lib, self.rank, self.fail_init_on = self._Lib(), rank, fail_init_on
        self.params = np.full(5, float(rank + 1), np.float32)
        self.calls = []

    def comm_unique_id(self):
        import numpy as np
        self.calls.append("uid")
        return np.arange(128, dtype=np.uint8)

    def comm_init(self, uid, rank, world):
        import numpy as np
        assert np.array_equal(np.asarray(uid), np.arange(128, dtype=np.uint8))
        self.calls.append("init")
        if rank in self.fail_init_on:
            raise RuntimeError("ncclCommInitRank: refused (test)")

    def comm_info(self):
        return {"rccl_ranks": int(os.environ["WORLD_SIZE"]), "rccl_user_rank": self.rank}

    def comm_broadcast_params(self, root):
        self.calls.append("bcast")

    def comm_destroy(self):
        self.calls.append("destroy")

    def get_params(self):
        return self.params

    def set_params(self, p):
        self.params = p


def test_exchange_setup_phases_keep_the_ranks_together(tmp_path):
    """attach_gradient_exchange: one rank failing ncclCommInitRank sends EVERY rank to the host path (the rank that formed a
    communicator destroys it), parameters come from rank 0, and the host sum is world x the mean."""
    from goldsrl import distributed as D
    child = tmp_path / "child.py"
    child.write_text(
        "import os, sys\n"
        "import numpy as np\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from goldsrl import distributed as D\n"
        "from test_spawn_ranks import _FakeNet\n"
        "r = D.Ranks().init(timeout_s=120)\n"
        "class Roll(object):\n"
        "    host_allreduce = None\n"
        "roll = Roll(); roll.net = _FakeNet(r.rank, fail_init_on={1})\n"
        "assert D.attach_gradient_exchange(roll, r) == 'host-store-fallback'\n"
        "assert roll.net.calls == (['uid', 'init', 'destroy'] if r.rank == 0 else ['init'])\n"
        "assert np.array_equal(roll.net.params, np.full(5, 1.0, np.float32))\n"
        "g, world = roll.host_allreduce(np.full(3, r.rank + 1.0, np.float32))\n"
        "assert world == 2 and (g == 3.0).all()\n"
        "roll2 = Roll(); roll2.net = _FakeNet(r.rank, fail_init_on=set())\n"
        "assert D.attach_gradient_exchange(roll2, r) == 'rccl' and roll2.net.calls[-1] == 'bcast' and roll2.host_allreduce is None\n"
        "roll3 = Roll(); roll3.net = _FakeNet(r.rank, fail_init_on=set())\n"
        "assert D.attach_gradient_exchange(roll3, r, prefer='host') == 'host-store-fallback' and 'init' not in roll3.net.calls\n"
        "open(os.path.join(%r, 'ok%%d' %% r.rank), 'w').write('ok')\n"
        "r.close()\n" % (os.path.join(ROOT, "golds-rl-gym_amd"), os.path.join(ROOT, "tests"), str(tmp_path)))
    assert D.spawn_local_ranks([sys.executable, str(child)], 2) == 0
    assert sorted(p.name for p in tmp_path.iterdir() if p.name.startswith("ok")) == ["ok0", "ok1"]


def test_host_exchange_update_is_taken_or_left_by_every_rank_together(tmp_path):
    """goldsrl.rollout._GradientExchange._update on the host path, two ranks over the TCP store: a local gradient pass that raises
    on ONE rank makes both raise before the exchange; a rank whose rollout left the fp16 range (update_skipped in range_info) makes
    BOTH return NaN statistics without applying anything; otherwise both apply the mean of the ranks' gradients."""
    from goldsrl import distributed as D
    child = tmp_path / "child.py"
    child.write_text(
        "import math, os, sys\n"
        "import numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from goldsrl import distributed as D\n"
        "from goldsrl.rollout import _GradientExchange\n"
        "r = D.Ranks().init(timeout_s=120)\n"
        "class Net(object):\n"
        "    def __init__(self, fail, skipped):\n"
        "        self.fail, self.skipped, self.applied, self.g = fail, skipped, None, np.full(4, r.rank + 1.0, np.float32)\n"
        "    def train_rollout_grads(self):\n"
        "        if self.fail: raise RuntimeError('local pass failed (test)')\n"
        "    def range_info(self):\n"
        "        return {'gemm_f32': self.skipped, 'fallbacks': int(self.skipped), 'update_skipped': self.skipped}\n"
        "    def get_grads(self): return self.g\n"
        "    def set_grads(self, g): self.g = g\n"
        "    def apply_grads(self, lr, scale):\n"
        "        self.applied = self.g * scale\n"
        "        return {'loss': 1.0}\n"
        "def make(fail=False, skipped=False):\n"
        "    x = _GradientExchange(); x.net = Net(fail, skipped); x.lr = 1e-3; x.ranks = r\n"
        "    x.host_allreduce = lambda g: (r.allreduce_sum_f32(g), r.world)\n"
        "    return x\n"
        "x = make(fail=(r.rank == 1))\n"
        "try:\n"
        "    x._update(); raise SystemExit('no error on rank %%d' %% r.rank)\n"
        "except RuntimeError as e:\n"
        "    assert ('test' in str(e)) == (r.rank == 1) and x.net.applied is None\n"
        "x = make(skipped=(r.rank == 0))\n"
        "s = x._update()\n"
        "assert math.isnan(s['loss']) and x.net.applied is None\n"
        "x = make()\n"
        "s = x._update()\n"
        "assert s == {'loss': 1.0} and np.array_equal(x.net.applied, np.full(4, 1.5, np.float32))\n"
        "open(os.path.join(%r, 'ok%%d' %% r.rank), 'w').write('ok')\n"
        "r.close()\n" % (os.path.join(ROOT, "golds-rl-gym_amd"), str(tmp_path)))
    assert D.spawn_local_ranks([sys.executable, str(child)], 2) == 0
    assert sorted(p.name for p in tmp_path.iterdir() if p.name.startswith("ok")) == ["ok0", "ok1"]


@pytest.mark.parametrize("answer", [b"\x05\x00\x00\x00\x00\x00\x00\x00hello", b"HTTP/1.1 400 Bad Request\r\n\r\n", b""],
                         ids=["framed", "raw-http", "silent"])
def test_a_stale_rendezvous_file_pointing_at_a_foreign_listener_is_rejected(tmp_path, answer):
    """A crashed job can leave its port file behind and the port may since belong to somebody else (round-3 advisor finding): rank 1
    reads the stale file first, reaches a listener that does not answer with this job's nonce -- a well-formed frame, raw bytes whose
    first eight read as a length of ~5e18 (round-4 advisor finding: bounded before anything is allocated), or nothing -- drops it and keeps polling; rank 0
    removes the stale file, publishes its own port + nonce, and the two meet.  The file lives in a 0700 directory of this user."""
    import socket
    import stat
    import threading
    from goldsrl import distributed as D
    key = "stale-test-%d" % os.getpid()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); mport = s.getsockname()[1]; s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(mport), GRL_RDZV_KEY=key, WORLD_SIZE="2")
    env.pop("GRL_STORE_PORT", None)
    # the foreign listener: accepts, reads, answers garbage
    foreign = socket.socket(); foreign.bind(("127.0.0.1", 0)); foreign.listen(4); foreign.settimeout(0.2)
    hits, stop = [], threading.Event()

    def serve():
        while not stop.is_set():
            try:
                c, _ = foreign.accept()
            except socket.timeout:
                continue
            hits.append(1)
            try:
                c.settimeout(1.0); c.recv(64); c.sendall(answer)
            except OSError:
                pass
            c.close()
    th = threading.Thread(target=serve, daemon=True); th.start()
    path = D._rendezvous_file(env)
    d = os.path.dirname(path)
    st = os.lstat(d)
    assert stat.S_ISDIR(st.st_mode) and (st.st_mode & 0o077) == 0 and st.st_uid == os.getuid()
    with open(path, "w") as f:
        f.write("%d old-nonce\n" % foreign.getsockname()[1])
    out = {}

    def rank(r):
        rk = D.Ranks(dict(env, RANK=str(r), LOCAL_RANK=str(r)))
        if r == 0:
            time.sleep(1.0)      # rank 1 meets the stale file first
        rk.init(timeout_s=60)
        out[r] = rk.sum(1.0 + r)
        rk.close()
    ts = [threading.Thread(target=rank, args=(r,)) for r in (0, 1)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(90)
    stop.set(); th.join(2); foreign.close()
    assert out == {0: 3.0, 1: 3.0}
    assert hits, "rank 1 never tried the stale port (the test did not exercise the rejection)"
    assert not os.path.exists(path)      # rank 0 removes its file on close


def test_rank_0_drops_a_client_that_does_not_speak_the_handshake():
    """The other direction of the same finding: something that is not a rank connects to rank 0's port and sends raw bytes (their
    first eight read as a frame length of ~5e18).  Rank 0 must close that connection and go on accepting -- not die allocating."""
    import socket
    import threading
    from goldsrl import distributed as D
    key = "scan-test-%d" % os.getpid()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); mport = s.getsockname()[1]; s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(mport), GRL_RDZV_KEY=key, WORLD_SIZE="2")
    env.pop("GRL_STORE_PORT", None)
    path = D._rendezvous_file(env)
    if os.path.exists(path):
        os.unlink(path)
    out, scanned = {}, threading.Event()

    def scanner():
        deadline = time.time() + 30
        while time.time() < deadline:
            try:
                with open(path) as f:
                    port = int(f.read().split()[0])
                break
            except (OSError, ValueError, IndexError):
                time.sleep(0.02)
        for raw in (b"GET / HTTP/1.1\r\nHost: x\r\n\r\n", b"\xff" * 8, b"abc"):
            c = socket.create_connection(("127.0.0.1", port), timeout=5)
            c.sendall(raw)
            time.sleep(0.2)
            c.close()
        scanned.set()

    def rank(r):
        rk = D.Ranks(dict(env, RANK=str(r), LOCAL_RANK=str(r)))
        if r == 1:
            assert scanned.wait(60)
        rk.init(timeout_s=60)
        out[r] = rk.sum(1.0 + r)
        rk.close()
    ts = [threading.Thread(target=f, args=a) for f, a in ((scanner, ()), (rank, (0,)), (rank, (1,)))]
    for t in ts:
        t.start()
    for t in ts:
        t.join(120)
    assert out == {0: 3.0, 1: 3.0}
