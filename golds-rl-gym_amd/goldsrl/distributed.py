"""One process per GPU: rendezvous and the per-rollout gradient exchange for a sharded env batch.

The reference is single-device (fed_gym/agents/paac/actor_learner.py:70-75); its only parallelism is
`np.split(emulators, workers)` over OS processes (paac/runners.py:18-19) talking through multiprocessing queues
(runners.py:34-54).  Here each rank owns a contiguous block of env ids (goldsrl/sharding.py) and a full replica of the
(small) policy; the ONE collective is the sum of the flat gradient once per rollout:

  * RCCL over xGMI (`grl_net_comm_*` / `grl_fnet_comm_*`): the all-reduce runs inside `train_rollout` on the
    handle's stream; or, when any rank cannot form the communicator (e.g. two ranks rehearsing on one GPU),
  * the host path: `train_rollout_grads -> sum over ranks through the store below -> set_grads -> apply_grads(lr, 1/world)`.

Either way: sum, scale by 1/world (the loss is a mean over the WHOLE batch, policy_v_network.py:54,62,246-251),
clip_by_global_norm after the reduction, Adam replicated, parameters broadcast from rank 0 once.

Host-side plumbing is the standard library only (no PyTorch, north_star): `Ranks` is a small TCP store -- rank 0 listens,
every other rank keeps one socket to it, and the ranks call the same collectives in the same order (SPMD).  It carries the
128-byte RCCL unique id, barriers, scalar max / min / sum, the parameter broadcast and (fallback only) the gradient sum.

Where rank 0 listens.  Under the driver's elastic launcher MASTER_PORT itself belongs to the launcher's own store, so the
ranks cannot bind it.  Rank 0 therefore binds GRL_STORE_PORT if set (multi-node: every rank connects to MASTER_ADDR on it), else
an ephemeral port on MASTER_ADDR which it publishes in a file keyed by (MASTER_PORT, parent pid) -- the launcher is the common
parent of all ranks of a one-node job, which is what bench.py's contract and the reference's process model are.
"""
import os
import socket
import struct
import sys
import tempfile
import time

import numpy as np

_HDR = struct.Struct("<Q")


def _send(sock, payload):
    sock.sendall(_HDR.pack(len(payload)) + payload)


def _recv_exact(sock, n):
    buf = bytearray(n)
    view, got = memoryview(buf), 0
    while got < n:
        k = sock.recv_into(view[got:], n - got)
        if k == 0:
            raise ConnectionError("rendezvous peer closed the connection")
        got += k
    return bytes(buf)


def _recv(sock, limit=None):
    """One frame.  `limit` bounds the length header BEFORE anything is allocated: during the handshake the other end may be anybody
    (an HTTP server answering 'HTTP/1.1 400', a scanner sending 'GET / HT' read as a length of ~6e18)."""
    (n,) = _HDR.unpack(_recv_exact(sock, _HDR.size))
    if limit is not None and n > limit:
        raise ConnectionError("rendezvous: a frame of %d bytes where at most %d can be this job's" % (n, limit))
    return _recv_exact(sock, n)


def _rendezvous_dir():
    """A directory only this user can write: <tmp>/goldsrl-<uid>, mode 0700, owned by us and not a symlink -- the port file is read
    by every rank, so nobody else must be able to plant one."""
    d = os.path.join(tempfile.gettempdir(), "goldsrl-%d" % os.getuid())
    try:
        os.mkdir(d, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(d)
    import stat
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise RuntimeError("rendezvous: %s is not a private directory of this user (mode %o, uid %d)" % (d, st.st_mode & 0o777, st.st_uid))
    return d


def _rendezvous_file(env):
    key = "%s_%s" % (env.get("MASTER_PORT", "29500"), env.get("GRL_RDZV_KEY", str(os.getppid())))
    return os.path.join(_rendezvous_dir(), "rdzv_%s" % key)


_HELLO, _ACK = b"goldsrl-rank", b"goldsrl-store"


class Ranks(object):
    """rank / world / local device from the launcher's environment (the driver's elastic launcher or bench.py's own spawn) and the
    collectives the host side needs.  world == 1: every collective is the identity and nothing is opened."""

    def __init__(self, env=None):
        self.env = dict(os.environ if env is None else env)
        self.rank = int(self.env.get("RANK", "0"))
        self.world = int(self.env.get("WORLD_SIZE", "1"))
        self.local_rank = int(self.env.get("LOCAL_RANK", str(self.rank)))
        self._peers = None      # rank 0: sockets of ranks 1..world-1 (index r-1)
        self._sock = None       # other ranks: socket to rank 0
        self._listener = None
        self._file = None

    # ---------------------------------------------------------------- rendezvous
    def init(self, timeout_s=600):
        if self.world <= 1 or self._peers is not None or self._sock is not None:
            return self
        addr = self.env.get("MASTER_ADDR", "127.0.0.1")
        fixed = self.env.get("GRL_STORE_PORT")
        deadline = time.time() + timeout_s
        if self.rank == 0:
            ls = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            ls.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            ls.bind((addr, int(fixed) if fixed else 0))
            ls.listen(self.world)
            self._listener = ls
            # the job's nonce: every peer must present it and gets it echoed, so a peer that read a stale file (a crashed job's) and
            # reached somebody else's listener rejects it and keeps polling, and rank 0 drops connections that are not its peers'
            nonce = self.env.get("GRL_RDZV_KEY", "") if fixed else "%s-%s" % (self.env.get("GRL_RDZV_KEY", ""), os.urandom(8).hex())
            if not fixed:
                self._file = _rendezvous_file(self.env)
                try:
                    os.unlink(self._file)            # a file left by a crashed job with the same key
                except FileNotFoundError:
                    pass
                tmp = "%s.%d" % (self._file, os.getpid())
                try:
                    os.unlink(tmp)
                except FileNotFoundError:
                    pass
                fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)
                with os.fdopen(fd, "w") as f:
                    f.write("%d %s\n" % (ls.getsockname()[1], nonce))
                os.replace(tmp, self._file)          # atomic: a reader sees the whole line or no file
            peers = [None] * (self.world - 1)
            ls.settimeout(1.0)
            while any(p is None for p in peers):
                if time.time() > deadline:
                    raise TimeoutError("rendezvous: %d of %d ranks connected" % (sum(p is not None for p in peers) + 1, self.world))
                try:
                    c, _ = ls.accept()
                except socket.timeout:
                    continue
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                c.settimeout(5.0)
                want = _HELLO + nonce.encode()
                try:
                    hello = _recv(c, len(want) + 4)
                except (OSError, ConnectionError):      # not framed like a rank's hello (or nothing at all): not a peer
                    c.close()
                    continue
                if len(hello) != len(want) + 4 or hello[:len(want)] != want:
                    c.close()                        # not a rank of this job (port scanner, a peer of another job): ignore it
                    continue
                r = struct.unpack("<i", hello[len(want):])[0]
                if not (1 <= r < self.world) or peers[r - 1] is not None:
                    c.close()
                    raise RuntimeError("rendezvous: unexpected rank %d" % r)
                _send(c, _ACK + nonce.encode())
                c.settimeout(timeout_s)
                peers[r - 1] = c
            self._peers = peers
        else:
            last = None
            while True:
                if time.time() > deadline:
                    raise TimeoutError("rendezvous: rank %d could not reach rank 0 (%s)" % (self.rank, last))
                s = None
                try:
                    if fixed:
                        port, nonce = int(fixed), self.env.get("GRL_RDZV_KEY", "")
                    else:
                        with open(_rendezvous_file(self.env)) as f:
                            port, _, nonce = f.read().strip().partition(" ")
                        port = int(port)
                    s = socket.create_connection((addr, port), timeout=5.0)
                    s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    s.settimeout(5.0)
                    _send(s, _HELLO + nonce.encode() + struct.pack("<i", self.rank))
                    if _recv(s, len(_ACK) + len(nonce.encode())) != _ACK + nonce.encode():
                        raise ConnectionError("the listener on port %d is not this job's rank 0" % port)
                    break
                except (OSError, ValueError, ConnectionError) as e:   # file not there yet, stale port, rank 0 not listening yet, foreign listener
                    last = e
                    if s is not None:
                        s.close()
                    time.sleep(0.05)
            s.settimeout(timeout_s)
            self._sock = s
        self.barrier()
        return self

    # ---------------------------------------------------------------- collectives (star through rank 0)
    def allgather_bytes(self, payload):
        """Every rank contributes `payload`; every rank receives the list of all payloads in rank order."""
        if self.world <= 1:
            return [bytes(payload)]
        if self._peers is None and self._sock is None:
            raise RuntimeError("Ranks.init() has not run")
        if self.rank == 0:
            parts = [bytes(payload)] + [_recv(p) for p in self._peers]
            blob = b"".join(_HDR.pack(len(x)) + x for x in parts)
            for p in self._peers:
                _send(p, blob)
            return parts
        _send(self._sock, bytes(payload))
        blob, parts, off = _recv(self._sock), [], 0
        while off < len(blob):
            (n,) = _HDR.unpack_from(blob, off)
            off += _HDR.size
            parts.append(blob[off:off + n])
            off += n
        return parts

    def reduce_bytes(self, payload, combine):
        """Every rank contributes `payload`; rank 0 folds them with combine(list) -> bytes and everyone gets the result
        (the payload crosses each socket once in each direction: what the gradient fallback uses)."""
        if self.world <= 1:
            return combine([bytes(payload)])
        if self.rank == 0:
            out = combine([bytes(payload)] + [_recv(p) for p in self._peers])
            for p in self._peers:
                _send(p, out)
            return out
        _send(self._sock, bytes(payload))
        return _recv(self._sock)

    def barrier(self):
        self.allgather_bytes(b"")

    def _scalars(self, value):
        return [struct.unpack("<d", x)[0] for x in self.allgather_bytes(struct.pack("<d", float(value)))]

    def max(self, value):
        return max(self._scalars(value))

    def min(self, value):
        return min(self._scalars(value))

    def sum(self, value):
        return float(sum(self._scalars(value)))

    def broadcast_bytes(self, payload, src=0):
        return self.allgather_bytes(payload if self.rank == src else b"")[src]

    def broadcast_array(self, arr, src=0):
        a = np.ascontiguousarray(arr)
        out = np.frombuffer(self.broadcast_bytes(a.tobytes(), src), dtype=a.dtype)
        return out.reshape(a.shape).copy()

    def allreduce_sum_f32(self, arr):
        """Sum of a float32 array over ranks, added in rank order on rank 0 (every rank receives the same bits)."""
        a = np.ascontiguousarray(arr, dtype=np.float32)

        def combine(parts):
            acc = np.frombuffer(parts[0], np.float32).copy()
            for x in parts[1:]:
                acc += np.frombuffer(x, np.float32)
            return acc.tobytes()
        return np.frombuffer(self.reduce_bytes(a.tobytes(), combine), np.float32).reshape(a.shape).copy()

    def close(self):
        for s in (self._peers or []) + [self._sock, self._listener]:
            if s is not None:
                try:
                    s.close()
                except OSError:
                    pass
        self._peers = self._sock = self._listener = None
        if self._file:
            try:
                os.remove(self._file)
            except OSError:
                pass
            self._file = None


def attach_gradient_exchange(roll, ranks, prefer="rccl"):
    """Make `roll` (goldsrl.rollout.ConvPolicyRollout / FlatPolicyRollout, or the learner's update object) exchange gradients
    over `ranks` and start every rank from rank 0's parameters.  Returns the exchange that will run:
    "none" | "rccl" | "host-store-fallback".

    The setup is a sequence of phases every rank walks through together, whatever fails on whichever rank, so that no rank
    is left alone inside a collective:
      1. agree (min) that RCCL is wanted and the id call works on rank 0 -- before any rank enters ncclCommInitRank;
      2. broadcast the 128-byte unique id unconditionally (zeros if phase 1 said no);
      3. ncclCommInitRank on every rank (RCCL itself blocks until all `world` ranks arrive or one of them errors);
      4. agree (min) on the result; all keep the communicator or all drop it and take the host path.
    """
    if ranks.world <= 1:
        return "none"
    net, rank, world = roll.net, ranks.rank, ranks.world
    nbytes = int(net.lib.grl_comm_unique_id_bytes())
    uid, want = np.zeros(nbytes, np.uint8), 1 if prefer == "rccl" else 0
    if want and rank == 0:
        try:
            uid = net.comm_unique_id()
        except Exception as e:       # noqa: BLE001
            sys.stderr.write("rank 0: ncclGetUniqueId failed (%s)\n" % e)
            want = 0
    want = int(ranks.min(want))                                   # phase 1
    uid = np.frombuffer(ranks.broadcast_bytes(uid.tobytes(), 0), np.uint8)        # phase 2
    ok = 0
    if want:
        try:
            net.comm_init(uid, rank, world)                       # phase 3
            ok = 1
        except Exception as e:       # noqa: BLE001 -- any failure means "no device communicator on this rank"
            sys.stderr.write("rank %d: RCCL communicator unavailable (%s)\n" % (rank, e))
    if int(ranks.min(ok)) == 1:                                   # phase 4
        info = net.comm_info()
        if info["rccl_ranks"] != world or info["rccl_user_rank"] != rank:
            raise RuntimeError("RCCL reports %r for rank %d of %d" % (info, rank, world))
        net.comm_broadcast_params(0)
        return "rccl"
    if ok:
        net.comm_destroy()
    net.set_params(ranks.broadcast_array(net.get_params(), 0))

    def host_allreduce(g):
        return ranks.allreduce_sum_f32(g), world
    roll.host_allreduce = host_allreduce
    roll.ranks = ranks
    return "host-store-fallback"


def params_equal_across_ranks(net, ranks):
    """Cross-rank check that the replicas hold bit-identical parameters (CRC of the flat vector, compared on every rank)."""
    import zlib
    crc = zlib.crc32(np.ascontiguousarray(net.get_params()).tobytes())
    crcs = [struct.unpack("<I", x)[0] for x in ranks.allgather_bytes(struct.pack("<I", crc))]
    return all(c == crcs[0] for c in crcs)


def spawn_local_ranks(argv, n, extra_env=None, port=None, poll_s=0.2):
    """Start `n` fresh rank processes of `argv` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set) from a parent that has NOT
    touched the GPU, wait for them, and stop the others as soon as one fails (a rank that dies would otherwise leave its
    peers waiting in the rendezvous).  Children inherit stdout/stderr.  Returns the first non-zero exit code, or 0.

    HSA_ENABLE_IPC_MODE_LEGACY=0 is passed on (defaulting to 0): RCCL's intra-node transport and any cross-process device-memory
    sharing go through hipIpcGetMemHandle, and this pool's host driver only supports dmabuf IPC -- with the legacy mode the
    handle export fails with `invalid argument` (platform note of the GPU image; the driver exports the same value in its own
    launches).  A caller that sets the variable keeps its value."""
    import subprocess
    if port is None:
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), GRL_RDZV_KEY=str(os.getpid()),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen(list(argv), env=env))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(poll_s)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in live:          # exact PIDs we started, never a pattern
                    q.terminate()
    for p in procs:
        try:
            p.wait(timeout=30)
        except subprocess.TimeoutExpired:
            p.kill()
    return rc
