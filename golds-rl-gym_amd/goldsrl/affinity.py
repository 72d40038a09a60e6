"""Pin a rank's host threads to the cores next to its GPU (one process per GPU: bench.py, PAACLearner._ranks).

The reference splits its envs over worker PROCESSES and leaves their placement to the OS (fed_gym/agents/paac/runners.py:18-54).
Here a rank's host side is one thread that enqueues ~10^4 kernel launches per PAAC update on four HIP streams plus RCCL's proxy
threads; on an 8-GPU host with two sockets a rank that runs on the far socket pays the inter-socket hop on every doorbell and
every completion signal, and eight unpinned ranks migrate over each other.  `pin_to_gpu(ordinal)` restricts the CALLING THREAD (and
every thread created after the call, which inherits the mask: the HIP runtime's and RCCL's, when it runs before they start) to
the cores of the GPU's NUMA node -- found in sysfs, WITHOUT any HIP call.  Threads that exist already keep their mask:
`os.sched_setaffinity(0, ...)` is per thread on Linux, so e.g. the BLAS / OpenMP pools numpy created at import stay where they were.

  /sys/class/kfd/kfd/topology/nodes/<k>/properties   `simd_count` > 0 marks a GPU node (node order = HIP ordinal order),
                                                     `domain`, `location_id` = its PCI address (bus << 8 | device << 3 | function)
  /sys/bus/pci/devices/<dddd:bb:dd.f>/local_cpulist   the cores of that device's NUMA node ("0-47,96-143")

ROCR_VISIBLE_DEVICES, then HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES are applied when they are plain index lists (a UUID list makes
the mapping unknowable here: no pinning).  Nothing is pinned when the information is missing, when the node's cores and the cores
the process may use (affinity mask, i.e. what a cgroup cpuset or the launcher left) do not intersect, or with GRL_PIN_CPUS=off.
"""
import os


def parse_cpulist(text):
    """'0-3,8,10-11' -> {0, 1, 2, 3, 8, 10, 11}"""
    cpus = set()
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def _visible(indices, env, names):
    """Apply a *_VISIBLE_DEVICES list of plain indices to `indices`; None if it is set but not an index list."""
    for name in names:
        v = env.get(name)
        if v is None or v.strip() == "":
            continue
        try:
            sel = [int(x) for x in v.split(",") if x.strip() != ""]
        except ValueError:
            return None
        if any(i < 0 or i >= len(indices) for i in sel):
            return None
        return [indices[i] for i in sel]
    return indices


def gpu_pci_addresses(sysfs="/sys", env=None):
    """PCI addresses of the GPUs in HIP ordinal order, or None when sysfs does not say."""
    env = os.environ if env is None else env
    base = os.path.join(sysfs, "class", "kfd", "kfd", "topology", "nodes")
    try:
        nodes = sorted((int(n) for n in os.listdir(base) if n.isdigit()))
    except OSError:
        return None
    gpus = []
    for n in nodes:
        props = {}
        try:
            with open(os.path.join(base, str(n), "properties")) as f:
                for line in f:
                    k, _, v = line.strip().partition(" ")
                    props[k] = v
            if int(props.get("simd_count", "0")) <= 0:
                continue
            loc, dom = int(props["location_id"]), int(props.get("domain", "0"))
        except (OSError, ValueError, KeyError):
            return None
        gpus.append("%04x:%02x:%02x.%x" % (dom, (loc >> 8) & 0xFF, (loc >> 3) & 0x1F, loc & 0x7))
    gpus = _visible(gpus, env, ("ROCR_VISIBLE_DEVICES",))
    if gpus is None:
        return None
    return _visible(gpus, env, ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"))


def gpu_local_cpus(ordinal, sysfs="/sys", env=None):
    """(cores of the NUMA node of HIP device `ordinal`, its PCI address, its NUMA node) or (None, reason, None)."""
    gpus = gpu_pci_addresses(sysfs, env)
    if gpus is None:
        return None, "no KFD topology in sysfs (or a *_VISIBLE_DEVICES list that is not plain indices)", None
    if not (0 <= ordinal < len(gpus)):
        return None, "device ordinal %d beyond the %d GPUs sysfs lists" % (ordinal, len(gpus)), None
    dev = os.path.join(sysfs, "bus", "pci", "devices", gpus[ordinal])
    try:
        with open(os.path.join(dev, "local_cpulist")) as f:
            cpus = parse_cpulist(f.read())
    except (OSError, ValueError):
        return None, "no local_cpulist for %s" % gpus[ordinal], None
    node = None
    try:
        with open(os.path.join(dev, "numa_node")) as f:
            node = int(f.read().strip())
    except (OSError, ValueError):
        pass
    if not cpus:
        return None, "empty local_cpulist for %s" % gpus[ordinal], node
    return cpus, gpus[ordinal], node


def pci_local_cpus(pci, sysfs="/sys"):
    """(cores, NUMA node) next to the PCI device `pci` ('dddd:bb:dd.f'), or (None, None)."""
    dev = os.path.join(sysfs, "bus", "pci", "devices", pci)
    try:
        with open(os.path.join(dev, "local_cpulist")) as f:
            cpus = parse_cpulist(f.read())
    except (OSError, ValueError):
        return None, None
    node = None
    try:
        with open(os.path.join(dev, "numa_node")) as f:
            node = int(f.read().strip())
    except (OSError, ValueError):
        pass
    return (cpus or None), node


def pin_to_pci(pci, sysfs="/sys", env=None, apply=True):
    """The same restriction for a caller that already knows its GPU's PCI address -- from the HIP runtime
    (_ffi.device_pci_address) where the KFD topology is not readable (containers that hide /sys/class/kfd's GPU nodes).  The
    runtime is up by then, so only the CALLING thread moves (the one that enqueues the launches); its helper threads keep the mask
    they were born with."""
    env = os.environ if env is None else env
    rep = {"pinned": False, "pci": pci, "via": "hipDeviceGetPCIBusId"}
    if env.get("GRL_PIN_CPUS", "on").lower() in ("off", "0", "no"):
        rep["reason"] = "GRL_PIN_CPUS=off"
        return rep
    if not pci or not hasattr(os, "sched_setaffinity"):
        rep["reason"] = "no PCI address / no sched_setaffinity"
        return rep
    cpus, node = pci_local_cpus(pci, sysfs)
    if cpus is None:
        rep["reason"] = "no local_cpulist for %s" % pci
        return rep
    rep["numa_node"] = node
    try:
        allowed = os.sched_getaffinity(0)
    except OSError as e:
        rep["reason"] = "sched_getaffinity: %s" % e
        return rep
    use = cpus & allowed
    rep["allowed_cpus"], rep["node_cpus"] = len(allowed), len(cpus)
    if not use:
        rep["reason"] = "the GPU's cores and the cores this process may use do not intersect"
        return rep
    if use != allowed and apply:
        try:
            os.sched_setaffinity(0, use)
        except OSError as e:
            rep["reason"] = "sched_setaffinity: %s" % e
            return rep
    rep.update(pinned=True, cpus=len(use))
    return rep


def pin_to_gpu(ordinal, sysfs="/sys", env=None, apply=True):
    """Restrict this process to the cores of the GPU's NUMA node.  Call BEFORE the first HIP call (threads the runtime starts
    afterwards inherit the mask).  Returns a small report for the bench line / the log; never raises."""
    env = os.environ if env is None else env
    rep = {"pinned": False, "device": int(ordinal)}
    if env.get("GRL_PIN_CPUS", "on").lower() in ("off", "0", "no"):
        rep["reason"] = "GRL_PIN_CPUS=off"
        return rep
    if not hasattr(os, "sched_setaffinity"):
        rep["reason"] = "no sched_setaffinity on this platform"
        return rep
    cpus, what, node = gpu_local_cpus(ordinal, sysfs, env)
    if cpus is None:
        rep["reason"] = what
        return rep
    rep.update(pci=what, numa_node=node)
    try:
        allowed = os.sched_getaffinity(0)
    except OSError as e:
        rep["reason"] = "sched_getaffinity: %s" % e
        return rep
    use = cpus & allowed
    rep["allowed_cpus"], rep["node_cpus"] = len(allowed), len(cpus)
    if not use:
        rep["reason"] = "the GPU's cores and the cores this process may use do not intersect"
        return rep
    if use == allowed:
        rep.update(pinned=True, cpus=len(use), reason="already inside the GPU's NUMA node")
        return rep
    if apply:
        try:
            os.sched_setaffinity(0, use)
        except OSError as e:
            rep["reason"] = "sched_setaffinity: %s" % e
            return rep
    rep.update(pinned=True, cpus=len(use))
    return rep
