"""Which host CPUs sit next to a GPU -- read from sysfs alone, so a rank can bind itself BEFORE anything touches the device
(BASELINE configs[3]: one rank per GPU on a two-socket 8-GPU node).  The pool's worker threads hash the batch
verifier's weight transcripts and stage page-locked buffers (csrc/pool.hip): threads created after the rank set its affinity
inherit it, and pages pinned by those threads are allocated on their node (first touch under the default local policy), so
binding the rank's process places both.

    HIP device i  ->  the i-th GPU node of /sys/class/kfd/kfd/topology/nodes (those with simd_count > 0, in node order,
                      filtered by ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES when set)
                  ->  PCI address domain:bus:dev.fn from that node's `domain` and `location_id` properties
                  ->  /sys/bus/pci/devices/<bdf>/numa_node  ->  /sys/devices/system/node/node<k>/cpulist

`root` replaces "/" (tests build a fake tree).  Everything degrades to "no information": (None, []) -- the caller then keeps
the affinity it has."""
import os


def parse_cpulist(text):
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-")
            out.extend(range(int(a), int(b) + 1))
        else:
            out.append(int(part))
    return sorted(set(out))


def _read(path):
    try:
        with open(path) as f:
            return f.read()
    except OSError:
        return None


def gpu_pci_addresses(root="/"):
    """PCI addresses of the GPU nodes of the KFD topology, in HIP's enumeration order before any *_VISIBLE_DEVICES filter"""
    base = os.path.join(root, "sys/class/kfd/kfd/topology/nodes")
    try:
        nodes = sorted((int(n) for n in os.listdir(base) if n.isdigit()))
    except OSError:
        return []
    out = []
    for n in nodes:
        txt = _read(os.path.join(base, str(n), "properties"))
        if txt is None:
            continue
        props = {}
        for line in txt.splitlines():
            kv = line.split()
            if len(kv) == 2 and kv[1].lstrip("-").isdigit():
                props[kv[0]] = int(kv[1])
        if props.get("simd_count", 0) <= 0:
            continue                                                   # a CPU node
        loc, dom = props.get("location_id", 0), props.get("domain", 0)
        out.append("%04x:%02x:%02x.%x" % (dom, (loc >> 8) & 0xff, (loc >> 3) & 0x1f, loc & 7))
    return out


def visible_devices(n_total, env=None):
    """indices (into the unfiltered enumeration) of the devices HIP will show, in HIP's order: ROCR_VISIBLE_DEVICES filters first (the
    runtime below HIP), then exactly ONE of HIP_VISIBLE_DEVICES or -- only when that is unset -- its alias CUDA_VISIBLE_DEVICES (a
    launcher that sets both to the same list must not see the list applied twice)"""
    env = os.environ if env is None else env
    idx = list(range(n_total))
    hip = env.get("HIP_VISIBLE_DEVICES")
    second = "HIP_VISIBLE_DEVICES" if (hip is not None and hip.strip() != "") else "CUDA_VISIBLE_DEVICES"
    for var in ("ROCR_VISIBLE_DEVICES", second):
        v = env.get(var)
        if v is None or v.strip() == "":
            continue
        try:
            pick = [int(x) for x in v.split(",") if x.strip() != ""]
        except ValueError:
            return idx                                                 # UUID form: no sysfs mapping attempted
        idx = [idx[i] for i in pick if 0 <= i < len(idx)]
    return idx


def numa_of_pci(bdf, root="/"):
    """(node or None, CPUs of that node) of a PCI device"""
    txt = _read(os.path.join(root, "sys/bus/pci/devices", bdf, "numa_node"))
    if txt is None:
        return None, []
    try:
        node = int(txt.strip())
    except ValueError:
        return None, []
    if node < 0:
        return None, []                                                # single-node machine, or the firmware does not say
    cl = _read(os.path.join(root, "sys/devices/system/node/node%d/cpulist" % node))
    return node, (parse_cpulist(cl) if cl else [])


def device_numa(device, root="/", env=None):
    """(PCI address or None, NUMA node or None, CPUs of the node) of HIP device `device`"""
    gpus = gpu_pci_addresses(root)
    vis = visible_devices(len(gpus), env)
    if device < 0 or device >= len(vis):
        return None, None, []
    bdf = gpus[vis[device]]
    node, cpus = numa_of_pci(bdf, root)
    return bdf, node, cpus


def rank_cpus(local_rank, world_local, allowed, root="/", env=None, per_rank=None):
    """CPUs for the rank that drives HIP device `local_rank` of `world_local` ranks on this host: the allowed CPUs (the job's
    affinity mask) on the GPU's NUMA node, split evenly among the ranks whose GPUs share that node, at most `per_rank` each.
    -> dict(cpus, node, pci, source)."""
    allowed = sorted(allowed)
    bdf, node, node_cpus = device_numa(local_rank, root, env)
    fallback = dict(cpus=allowed, node=None, pci=bdf, source="affinity mask unchanged (no NUMA information for the device)")
    if node is None:
        return fallback
    local = [c for c in allowed if c in set(node_cpus)]
    if not local:
        return dict(fallback, node=node, source="affinity mask unchanged (none of the allowed CPUs is on the device's node)")
    mates = [r for r in range(world_local) if device_numa(r, root, env)[1] == node]      # ranks sharing the node, this one included
    k = mates.index(local_rank) if local_rank in mates else 0
    share = max(1, len(local) // max(1, len(mates)))
    if per_rank:
        share = max(1, min(share, int(per_rank)))
    mine = local[k * share: (k + 1) * share] or local[-share:]
    return dict(cpus=mine, node=node, pci=bdf, source="CPUs of NUMA node %d (the device's), share %d of %d ranks on it" % (node, k, len(mates)))


def bind_rank(local_rank, world_local, root="/", per_rank=None):
    """sched_setaffinity of this process to rank_cpus(...); returns the dict, with `bound` saying whether the mask changed"""
    try:
        allowed = os.sched_getaffinity(0)
    except (AttributeError, OSError):
        return dict(cpus=[], node=None, pci=None, source="no sched_getaffinity on this platform", bound=False)
    r = rank_cpus(local_rank, world_local, allowed, root, per_rank=per_rank)
    r["bound"] = False
    if r["node"] is not None and r["cpus"] and set(r["cpus"]) != set(allowed):
        try:
            os.sched_setaffinity(0, r["cpus"])
            r["bound"] = True
        except OSError as e:
            r["source"] += " (sched_setaffinity failed: %r)" % (e,)
    return r
