"""Host logic of the N > 1 path that the 2-rank gloo tests cannot reach: the shard plan and the prediction-gather
row permutation for world sizes 3, 4 and 8 (the driver's scaling run), checked without any process group by emulating
what the per-round all-gathers deliver."""
import numpy as np
import pytest
import torch

from chromegcn_amd import synth
from chromegcn_amd.dist import plan_shards
from chromegcn_amd.finetune import GCNStage


class _Tiny(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.p = torch.nn.Parameter(torch.zeros(1))


def _stage_with(sizes, C, world, rank):
    st = GCNStage(_Tiny(), None, "none", "cpu", hip_graphs=False)
    for nm, n in sizes.items():
        st.add_chromosome(nm, {"forward": torch.zeros(n, 8), "backward": torch.zeros(n, 8), "target": torch.zeros(n, C)})
    st.world, st.rank = world, rank
    return st


@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_gather_index_restores_reference_order(world):
    C = 3
    names = [c for c in synth.HG19_LEN if synth.split_of(c) == "train"]
    sizes = {c: max(2, synth.chrom_nodes(c) // 97) for c in names}          # genome proportions, scaled down
    stages = [_stage_with(sizes, C, world, r) for r in range(world)]
    plan = plan_shards({nm: stages[0].chroms[nm].cost for nm in names}, world)
    assert len(plan.rounds) == -(-len(names) // world)
    gps = [st._gather_plan(names, plan, C) for st in stages]
    # what rank r would put into its send slab of round k: rows tagged (chromosome index, row index)
    def rows_of(nm):
        i = names.index(nm)
        return torch.stack([torch.full((sizes[nm],), float(i)), torch.arange(sizes[nm], dtype=torch.float32),
                            torch.full((sizes[nm],), 7.0)], 1)
    want = torch.cat([rows_of(nm) for nm in names], 0)                         # finetune.py:52 order
    for r in range(world):                                                      # every rank must reconstruct the same thing
        gp = gps[r]
        for k, group in enumerate(plan.rounds):
            recv = gp["recv"][k].view(world, -1, C)
            for src in range(world):                                            # emulate all_gather_into_tensor
                slab = torch.zeros_like(gps[src]["send"][k])
                nm = group[src] if src < len(group) else None
                if nm is not None:
                    slab[:sizes[nm]] = rows_of(nm)
                recv[src].copy_(slab)
        got = gp["recv_all"].index_select(0, gp["index"])
        assert torch.equal(got, want), "rank %d of %d" % (r, world)


def test_genome_plan_is_balanced_at_eight_ranks():
    names = [c for c in synth.HG19_LEN if synth.split_of(c) == "train"]
    d = 128
    cost = {c: (synth.chrom_nodes(c) + 500000.0) * d + 3.0 * synth.chrom_nodes(c) * d * d / 16.0 for c in names}
    plan = plan_shards(cost, 8)
    assert len(plan.rounds) == 2 and all(len(r) == 8 and all(g is not None for g in r) for r in plan.rounds)
    load = np.array(plan.load)
    assert load.max() / load.mean() < 1.10                                     # LPT: heaviest rank within 10 % of the mean
    # each round synchronises at its all-reduce: the round's span is its slowest member
    span = sum(max(cost[g] for g in r) for r in plan.rounds)
    assert span / (sum(cost.values()) / 8) < 1.35


def test_copy_groups_of_a_to_cpu_split_keep_the_copies_ahead_of_the_compute():
    """GCNStage._copy_groups (host logic, no device): a split that returns CPU predictions is replayed as a few graphs, each
    followed by one copy of its rows.  Every group must hold at least half of the rows still to come (so the PCIe copies,
    2-3x faster per row than the compute, never fall behind), the groups must partition the names in order, and without
    epoch graphs every chromosome is its own group."""
    from chromegcn_amd import synth
    from chromegcn_amd.finetune import GCNStage
    st = GCNStage.__new__(GCNStage)
    names = [c for c in synth.HG19_LEN if synth.split_of(c) == "train"]
    st._meta = {c: (synth.chrom_nodes(c), 103, 1.0) for c in names}
    st.epoch_graph, st.hip_graphs = True, True
    groups = st._copy_groups(names)
    assert [c for g in groups for c in g] == names and [len(g) for g in groups] == [6, 4, 3, 2, 1]
    left = sum(st._meta[c][0] for c in names)
    for g in groups:
        rows = sum(st._meta[c][0] for c in g)
        assert 2 * rows >= left
        left -= rows
    assert left == 0
    assert st._copy_groups(names[:1]) == [names[:1]] and st._copy_groups([]) == []
    st.epoch_graph = False
    assert st._copy_groups(names) == [[c] for c in names]
