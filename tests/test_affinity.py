"""CPU: goldsrl/affinity.py finds the cores next to a GPU in sysfs without any HIP call (a fake sysfs tree here), honours the
*_VISIBLE_DEVICES index lists, and never pins to an empty set."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "golds-rl-gym_amd"))


def _fake_sysfs(tmp_path, gpus):
    """gpus: list of (domain, bus, dev, fn, cpulist, numa) in KFD node order, behind two CPU nodes."""
    nodes = tmp_path / "class" / "kfd" / "kfd" / "topology" / "nodes"
    for k in range(2):      # CPU nodes: simd_count 0
        d = nodes / str(k); d.mkdir(parents=True)
        (d / "properties").write_text("cpu_cores_count 48\nsimd_count 0\nlocation_id 0\ndomain 0\n")
    for i, (dom, bus, dev, fn, cpus, numa) in enumerate(gpus):
        d = nodes / str(2 + i); d.mkdir(parents=True)
        (d / "properties").write_text("cpu_cores_count 0\nsimd_count 1024\nlocation_id %d\ndomain %d\n" % ((bus << 8) | (dev << 3) | fn, dom))
        p = tmp_path / "bus" / "pci" / "devices" / ("%04x:%02x:%02x.%x" % (dom, bus, dev, fn)); p.mkdir(parents=True)
        (p / "local_cpulist").write_text(cpus + "\n")
        (p / "numa_node").write_text("%d\n" % numa)
    return str(tmp_path)


def test_parse_cpulist():
    from goldsrl import affinity as A
    assert A.parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    assert A.parse_cpulist("") == set()


def test_gpu_cores_from_a_fake_sysfs_and_visible_devices(tmp_path):
    from goldsrl import affinity as A
    sysfs = _fake_sysfs(tmp_path, [(0, 0x05, 0, 0, "0-47,96-143", 0), (0, 0x15, 0, 0, "0-47,96-143", 0),
                                   (0, 0x85, 0, 0, "48-95,144-191", 1), (1, 0x95, 0, 0, "48-95,144-191", 1)])
    assert A.gpu_pci_addresses(sysfs, {}) == ["0000:05:00.0", "0000:15:00.0", "0000:85:00.0", "0001:95:00.0"]
    cpus, pci, node = A.gpu_local_cpus(2, sysfs, {})
    assert cpus == set(range(48, 96)) | set(range(144, 192)) and pci == "0000:85:00.0" and node == 1
    # ROCR filters first, HIP indexes what is left
    env = {"ROCR_VISIBLE_DEVICES": "2,3,0", "HIP_VISIBLE_DEVICES": "1"}
    assert A.gpu_pci_addresses(sysfs, env) == ["0001:95:00.0"]
    assert A.gpu_local_cpus(0, sysfs, env)[2] == 1
    # a UUID list cannot be mapped here: no answer, no pinning
    assert A.gpu_pci_addresses(sysfs, {"ROCR_VISIBLE_DEVICES": "GPU-abcdef"}) is None
    rep = A.pin_to_gpu(0, sysfs, {"ROCR_VISIBLE_DEVICES": "GPU-abcdef"})
    assert rep["pinned"] is False and "VISIBLE_DEVICES" in rep["reason"]
    assert A.gpu_local_cpus(7, sysfs, {})[0] is None


def test_pin_intersects_with_the_allowed_cores_and_can_be_switched_off(tmp_path):
    from goldsrl import affinity as A
    allowed = sorted(os.sched_getaffinity(0))
    half = allowed[:max(1, len(allowed) // 2)]
    sysfs = _fake_sysfs(tmp_path, [(0, 5, 0, 0, ",".join(str(c) for c in half), 0), (0, 6, 0, 0, "100000-100003", 1)])
    rep = A.pin_to_gpu(0, sysfs, {}, apply=False)      # apply=False: report only, the test process keeps its mask
    assert rep["pinned"] and rep["cpus"] == len(half) and rep["numa_node"] == 0 and rep["pci"] == "0000:05:00.0"
    rep = A.pin_to_gpu(1, sysfs, {}, apply=False)      # cores this process may not use: leave the mask alone
    assert not rep["pinned"] and "intersect" in rep["reason"]
    assert A.pin_to_gpu(0, sysfs, {"GRL_PIN_CPUS": "off"})["reason"] == "GRL_PIN_CPUS=off"
    assert A.pin_to_gpu(0, str(tmp_path / "nothing"), {})["pinned"] is False
    # applied for real in a child: the mask shrinks to the node's cores
    import subprocess
    code = ("import os, sys; sys.path.insert(0, %r); from goldsrl import affinity as A; r = A.pin_to_gpu(0, %r, {}); "
            "assert r['pinned'], r; assert sorted(os.sched_getaffinity(0)) == %r; print('ok')" % (os.path.join(ROOT, "golds-rl-gym_amd"), sysfs, half))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr


def test_pin_by_pci_address_fallback(tmp_path):
    """Where sysfs hides the KFD GPU nodes the PCI address comes from the runtime (grl_device_pci_address); the rest is the same."""
    from goldsrl import affinity as A
    allowed = sorted(os.sched_getaffinity(0))
    sysfs = _fake_sysfs(tmp_path, [(0, 0x75, 0, 0, ",".join(str(c) for c in allowed[:1]), 1)])
    rep = A.pin_to_pci("0000:75:00.0", sysfs, {}, apply=False)
    assert rep["pinned"] and rep["cpus"] == 1 and rep["numa_node"] == 1 and rep["via"] == "hipDeviceGetPCIBusId"
    assert A.pin_to_pci("0000:76:00.0", sysfs, {}, apply=False)["pinned"] is False
    assert A.pin_to_pci(None, sysfs, {})["pinned"] is False
