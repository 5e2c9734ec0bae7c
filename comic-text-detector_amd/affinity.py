"""CPU placement of a rank's host side next to its GPU (one process per GPU, `bench.py --gpus N`, `dist.py`).

The reference is one process with one thread of control (`inference.py:141`); this build runs, per GPU, a launching
thread, 2-4 tail workers with up to 8 native geometry threads each and 2 loader threads -- 3-7 cores of CPU time per wall
second (`config.host_cpu_cores_used`).  Eight such ranks on a two-socket host are ~55 busy cores: left to the scheduler
they migrate across sockets, and a tail worker's page-locked result buffers (allocated on the node it first ran on) end
up on the far side of the inter-socket link from the GPU that DMAs into them.  `rank_cpus` gives every rank a fixed,
disjoint set of logical CPUs on the NUMA node of ITS GPU; `apply` binds the calling thread (threads it creates afterwards
inherit the mask: call it before the pools exist).

Topology sources, in order: an explicit `topology` dict (tests), sysfs (`/sys/bus/pci/devices/<bdf>/numa_node` of the
device's PCI address, `/sys/devices/system/node/node<k>/cpulist`), else a contiguous split of the usable CPUs.
"""
from __future__ import annotations

import glob
import os
from typing import Dict, Iterable, List, Optional, Sequence


def parse_cpulist(text: str) -> List[int]:
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the kernel's cpulist format)."""
    out: List[int] = []
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-", 1)
            out.extend(range(int(a), int(b) + 1))
        else:
            out.append(int(part))
    return sorted(set(out))


def _usable() -> List[int]:
    try:
        return sorted(os.sched_getaffinity(0))
    except AttributeError:                                    # not Linux
        return list(range(os.cpu_count() or 1))


def gpu_numa_node(index: int) -> int:
    """NUMA node of HIP device `index` from sysfs (-1: unknown / single node)."""
    bdf = None
    try:
        import torch
        p = torch.cuda.get_device_properties(index)
        bdf = f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
    except Exception:
        pass
    paths = [f"/sys/bus/pci/devices/{bdf}/numa_node"] if bdf else []
    cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/numa_node"),
                   key=lambda s: int("".join(ch for ch in s.split("/card")[1].split("/")[0] if ch.isdigit()) or 0))
    if not bdf and index < len(cards):
        paths.append(cards[index])
    for path in paths:
        try:
            with open(path) as f:
                return int(f.read().strip())
        except (OSError, ValueError):
            continue
    return -1


def system_topology(n_gpus: int) -> Dict[str, dict]:
    """{'gpu_node': {gpu: node}, 'node_cpus': {node: [cpus]}} from sysfs, restricted to the CPUs this process may use."""
    usable = set(_usable())
    node_cpus: Dict[int, List[int]] = {}
    for d in glob.glob("/sys/devices/system/node/node[0-9]*"):
        try:
            k = int(d.rsplit("node", 1)[1])
            with open(os.path.join(d, "cpulist")) as f:
                cpus = [c for c in parse_cpulist(f.read()) if c in usable]
            if cpus:
                node_cpus[k] = cpus
        except (OSError, ValueError):
            continue
    return {"gpu_node": {g: gpu_numa_node(g) for g in range(n_gpus)}, "node_cpus": node_cpus}


def rank_cpus(local_rank: int, local_world: int, gpu_of_rank: Optional[Sequence[int]] = None,
              topology: Optional[dict] = None, usable: Optional[Iterable[int]] = None) -> dict:
    """The logical CPUs of rank `local_rank` of `local_world` ranks on this host.

    Ranks whose GPUs sit on the same NUMA node split that node's CPUs into equal contiguous runs (in rank order); a rank
    whose GPU's node is unknown, or whose node has fewer CPUs than ranks, falls into the contiguous split of ALL usable CPUs
    (`source: "contiguous"`).  Disjoint across ranks by construction.  Returns {'cpus', 'node', 'source'}."""
    gpu_of_rank = list(gpu_of_rank) if gpu_of_rank is not None else list(range(local_world))
    usable = sorted(set(usable)) if usable is not None else _usable()
    topo = topology if topology is not None else system_topology(max(gpu_of_rank) + 1)
    gpu_node, node_cpus = topo.get("gpu_node", {}), {k: [c for c in v if c in set(usable)] for k, v in topo.get("node_cpus", {}).items()}
    node_of = [gpu_node.get(g, -1) for g in gpu_of_rank]
    ok = all(n in node_cpus for n in node_of)
    if ok:
        for n in set(node_of):
            if len(node_cpus[n]) < node_of.count(n):
                ok = False
    if not ok:                                                # one pool, cut into `local_world` contiguous runs
        k = len(usable) // max(1, local_world)
        if k == 0:
            return {"cpus": list(usable), "node": -1, "source": "shared (fewer CPUs than ranks)"}
        return {"cpus": usable[local_rank * k: (local_rank + 1) * k], "node": -1, "source": "contiguous"}
    n = node_of[local_rank]
    mates = [r for r in range(local_world) if node_of[r] == n]
    cpus = node_cpus[n]
    k = len(cpus) // len(mates)
    j = mates.index(local_rank)
    return {"cpus": cpus[j * k: (j + 1) * k], "node": n, "source": "numa"}


def apply(cpus: Sequence[int]) -> bool:
    """Binds the calling thread (and every thread it creates from now on) to `cpus`; False when the platform refuses."""
    if not cpus:
        return False
    try:
        os.sched_setaffinity(0, set(int(c) for c in cpus))
        return True
    except (AttributeError, OSError):
        return False
