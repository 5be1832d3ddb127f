"""ark_vrf_amd/numa.py: the CPUs next to a rank's GPU, from a FAKE sysfs tree of a two-socket 8-GPU node (SURVEY.md 8e,
BASELINE configs[3]: one rank per GPU): KFD topology -> PCI address -> numa_node -> cpulist, the split among the ranks of a
socket, the *_VISIBLE_DEVICES filters, and the fallbacks when sysfs says nothing."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ark_vrf_amd import numa  # noqa: E402

BUSES = [0x05, 0x15, 0x65, 0x75, 0x85, 0x95, 0xe5, 0xf5]          # eight GPUs, the first four on socket 0


def make_tree(tmp_path, numa_nodes=(0, 0, 0, 0, 1, 1, 1, 1), cpulists=("0-47,96-143", "48-95,144-191")):
    root = tmp_path
    nodes = root / "sys/class/kfd/kfd/topology/nodes"
    for i in range(2):                                              # two CPU nodes first, as KFD lists them
        d = nodes / str(i); d.mkdir(parents=True)
        (d / "properties").write_text("cpu_cores_count 96\nsimd_count 0\nlocation_id 0\ndomain 0\n")
    for g, bus in enumerate(BUSES):
        d = nodes / str(2 + g); d.mkdir(parents=True)
        (d / "properties").write_text("cpu_cores_count 0\nsimd_count 1024\nlocation_id %d\ndomain 0\ndrm_render_minor %d\n" % (bus << 8, 128 + g))
        p = root / ("sys/bus/pci/devices/0000:%02x:00.0" % bus); p.mkdir(parents=True)
        (p / "numa_node").write_text("%d\n" % numa_nodes[g])
    for k, cl in enumerate(cpulists):
        d = root / ("sys/devices/system/node/node%d" % k); d.mkdir(parents=True)
        (d / "cpulist").write_text(cl + "\n")
    return str(root)


def test_cpulist_parsing():
    assert numa.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert numa.parse_cpulist("") == []


def test_device_to_node_and_cpus(tmp_path):
    root = make_tree(tmp_path)
    assert numa.gpu_pci_addresses(root) == ["0000:%02x:00.0" % b for b in BUSES]
    assert numa.device_numa(0, root, env={}) == ("0000:05:00.0", 0, numa.parse_cpulist("0-47,96-143"))
    assert numa.device_numa(5, root, env={})[:2] == ("0000:95:00.0", 1)
    assert numa.device_numa(8, root, env={}) == (None, None, [])
    # visibility filters compose: ROCR first, then HIP, as the runtime applies them
    env = {"ROCR_VISIBLE_DEVICES": "4,5,6,7", "HIP_VISIBLE_DEVICES": "1,0"}
    assert numa.device_numa(0, root, env=env)[0] == "0000:95:00.0" and numa.device_numa(1, root, env=env)[0] == "0000:85:00.0"


def test_hip_and_cuda_visible_devices_set_to_the_same_list(tmp_path):
    """launchers often export HIP_VISIBLE_DEVICES and CUDA_VISIBLE_DEVICES with the same list: HIP reads the CUDA name only when its own is
    unset, so the list is applied ONCE (applied twice, "4,5,6,7" selects nothing and "1,0" swaps back)"""
    assert numa.visible_devices(8, {"HIP_VISIBLE_DEVICES": "4,5,6,7", "CUDA_VISIBLE_DEVICES": "4,5,6,7"}) == [4, 5, 6, 7]
    assert numa.visible_devices(8, {"HIP_VISIBLE_DEVICES": "1,0", "CUDA_VISIBLE_DEVICES": "1,0"}) == [1, 0]
    assert numa.visible_devices(8, {"CUDA_VISIBLE_DEVICES": "2,3"}) == [2, 3]                       # the alias alone
    assert numa.visible_devices(8, {"HIP_VISIBLE_DEVICES": "", "CUDA_VISIBLE_DEVICES": "2,3"}) == [2, 3]
    assert numa.visible_devices(8, {"ROCR_VISIBLE_DEVICES": "4,5,6,7", "HIP_VISIBLE_DEVICES": "1,0", "CUDA_VISIBLE_DEVICES": "3"}) == [5, 4]
    root = make_tree(tmp_path)
    pci, node, cpus = numa.device_numa(0, root, {"HIP_VISIBLE_DEVICES": "4,5,6,7", "CUDA_VISIBLE_DEVICES": "4,5,6,7"})
    assert node == 1 and pci is not None


def test_ranks_split_their_socket(tmp_path):
    root = make_tree(tmp_path)
    allowed = set(range(192))
    got = [numa.rank_cpus(r, 8, allowed, root, env={}) for r in range(8)]
    assert [g["node"] for g in got] == [0, 0, 0, 0, 1, 1, 1, 1]
    node0, node1 = set(numa.parse_cpulist("0-47,96-143")), set(numa.parse_cpulist("48-95,144-191"))
    for r, g in enumerate(got):
        assert len(g["cpus"]) == 24 and set(g["cpus"]) <= (node0 if r < 4 else node1)
    for a in range(8):
        for b in range(a + 1, 8):
            assert not set(got[a]["cpus"]) & set(got[b]["cpus"])    # no two ranks share a CPU
    # a 16-CPU cgroup that lies on socket 1 only: ranks of socket 0 keep their mask, ranks of socket 1 split the 16
    small = set(range(48, 64))
    g0, g5 = numa.rank_cpus(0, 8, small, root, env={}), numa.rank_cpus(5, 8, small, root, env={})
    assert g0["cpus"] == sorted(small) and "none of the allowed" in g0["source"]
    assert g5["cpus"] == [52, 53, 54, 55]
    assert numa.rank_cpus(5, 8, allowed, root, env={}, per_rank=2)["cpus"] == [50, 51]      # second rank of socket 1, two CPUs each


def test_fallbacks(tmp_path):
    # numa_node = -1 (single-socket box, or firmware silent) and a tree without KFD: the mask is left alone
    root = make_tree(tmp_path / "a", numa_nodes=(-1,) * 8)
    r = numa.rank_cpus(3, 8, {0, 1, 2, 3}, root, env={})
    assert r["node"] is None and r["cpus"] == [0, 1, 2, 3]
    empty = tmp_path / "b"; empty.mkdir()
    r = numa.rank_cpus(0, 1, {0, 1}, str(empty), env={})
    assert r["node"] is None and r["pci"] is None and r["cpus"] == [0, 1]


def test_bind_rank_on_this_box():
    """on the build container (no KFD, or one node) bind_rank must not change or break the affinity"""
    before = os.sched_getaffinity(0)
    r = numa.bind_rank(0, 1)
    assert os.sched_getaffinity(0) == before or r["bound"]
    if r["bound"]:
        os.sched_setaffinity(0, before)


def test_c_abi_lookup_agrees(tmp_path):
    """avrf_numa_cpus_of_pci (csrc/host_numa.h, what avrf_pool_create and avrf_host_alloc use) reads the same tree the same way"""
    import ctypes as C
    from ark_vrf_amd import _native as nat
    root = make_tree(tmp_path)
    L = nat.lib()
    L.avrf_numa_cpus_of_pci.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_int32), C.c_size_t, C.POINTER(C.c_int32)]
    L.avrf_numa_cpus_of_pci.restype = C.c_int
    buf = (C.c_int32 * 512)(); node = C.c_int32(-7)
    k = L.avrf_numa_cpus_of_pci(b"0000:95:00.0", root.encode(), buf, 512, C.byref(node))
    assert node.value == 1 and list(buf[:k]) == numa.parse_cpulist("48-95,144-191")
    k = L.avrf_numa_cpus_of_pci(b"0000:05:00.0", root.encode(), buf, 4, C.byref(node))      # cap respected
    assert node.value == 0 and list(buf[:k]) == [0, 1, 2, 3]
    assert L.avrf_numa_cpus_of_pci(b"0000:aa:00.0", root.encode(), buf, 512, C.byref(node)) == 0 and node.value == -1
    assert L.avrf_numa_cpus_of_pci(b"0000:95:00.0".upper(), root.encode(), buf, 512, C.byref(node)) > 0 and node.value == 1
