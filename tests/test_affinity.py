"""CPU placement of the ranks of `bench.py --gpus N` (comic-text-detector_amd/affinity.py): every rank gets a fixed,
disjoint set of logical CPUs on the NUMA node of its GPU.  Checked on fake topologies for 1 / 2 / 4 / 8 ranks -- no GPU, no
sysfs -- and once against this machine's real sysfs (whatever it is, the sets must be disjoint and non-empty)."""
import os

import pytest

from conftest import pkg


def topo_two_sockets(n_gpus=8, cpus_per_node=64):
    """GPUs 0..n/2-1 on node 0, the rest on node 1; hyperthread siblings numbered after the physical cores, as Linux does."""
    half = cpus_per_node // 2
    node_cpus = {0: list(range(0, half)) + list(range(2 * half, 3 * half)),
                 1: list(range(half, 2 * half)) + list(range(3 * half, 4 * half))}
    return {"gpu_node": {g: (0 if g < max(1, n_gpus // 2) else 1) for g in range(n_gpus)}, "node_cpus": node_cpus}


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_ranks_get_disjoint_cpus_of_their_gpus_node(world):
    A = pkg().affinity
    topo = topo_two_sockets(8, 64)
    usable = range(128)
    sets = [A.rank_cpus(r, world, topology=topo, usable=usable) for r in range(world)]
    seen = set()
    for r, s in enumerate(sets):
        assert s["source"] == "numa" and s["cpus"], s
        assert s["node"] == topo["gpu_node"][r]
        assert set(s["cpus"]) <= set(topo["node_cpus"][s["node"]])          # on the GPU's node
        assert not (seen & set(s["cpus"]))                                   # disjoint from every other rank
        seen |= set(s["cpus"])
    mates = max(sum(1 for r in range(world) if topo["gpu_node"][r] == n) for n in (0, 1))
    assert all(len(s["cpus"]) == 64 // mates for s in sets)                  # equal shares of a node
    assert sets == [A.rank_cpus(r, world, topology=topo, usable=usable) for r in range(world)]   # deterministic


def test_unknown_topology_falls_back_to_a_contiguous_split():
    A = pkg().affinity
    usable = list(range(3, 51))                                              # a cgroup's odd CPU range
    for topo in ({"gpu_node": {}, "node_cpus": {}}, {"gpu_node": {0: -1, 1: -1, 2: -1, 3: -1}, "node_cpus": {0: usable}}):
        sets = [A.rank_cpus(r, 4, topology=topo, usable=usable) for r in range(4)]
        assert all(s["source"] == "contiguous" and len(s["cpus"]) == 12 for s in sets)
        flat = [c for s in sets for c in s["cpus"]]
        assert flat == usable[:48] and len(set(flat)) == 48


def test_more_ranks_than_cpus_share_and_one_device_rehearsal_splits_one_node():
    A = pkg().affinity
    s = A.rank_cpus(3, 8, topology={"gpu_node": {}, "node_cpus": {}}, usable=[0, 1, 2])
    assert s["cpus"] == [0, 1, 2] and s["source"].startswith("shared")
    topo = topo_two_sockets(8, 64)                                           # all ranks on GPU 0 (bench's rehearsal)
    sets = [A.rank_cpus(r, 4, gpu_of_rank=[0, 0, 0, 0], topology=topo, usable=range(128)) for r in range(4)]
    assert all(x["node"] == 0 and len(x["cpus"]) == 16 for x in sets)
    assert len({c for x in sets for c in x["cpus"]}) == 64


def test_cpulist_parser_and_the_real_machine():
    A = pkg().affinity
    assert A.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11] and A.parse_cpulist("") == []
    sets = [A.rank_cpus(r, 2, gpu_of_rank=[0, 0]) for r in range(2)]          # sysfs or the fallback, whichever this box has
    assert all(s["cpus"] for s in sets)
    if len(os.sched_getaffinity(0)) >= 2:
        assert not (set(sets[0]["cpus"]) & set(sets[1]["cpus"]))
    before = os.sched_getaffinity(0)
    try:
        assert A.apply(sets[0]["cpus"])
        assert os.sched_getaffinity(0) == set(sets[0]["cpus"])
    finally:
        os.sched_setaffinity(0, before)
