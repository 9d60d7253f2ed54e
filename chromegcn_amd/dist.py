"""Chromosome -> rank sharding for the multi-GPU GCN stage.

The reference's GCN stage is single-GPU (finetune.py:29-49, README.md:45); sharding is new here and
exact because Hi-C graphs are intra-chromosomal (data/7create_graph_new.py:145): chromosomes are
independent units, so ranks exchange nothing on the data path.  The only collective is one all-reduce
of the flat parameter-gradient buffer per step group (finetune.GCNStage.train_group)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional


@dataclass
class ShardPlan:
    rounds: List[List[Optional[str]]]  # rounds[r][rank] = chromosome processed by `rank` in step group r (or None)
    owner: Dict[str, int]              # chromosome -> rank that processes it
    load: List[float]                  # summed cost per rank


def plan_shards(cost: Dict[str, float], world: int) -> ShardPlan:
    """Longest-processing-time-first: heaviest chromosome to the least-loaded rank, with at most
    ceil(k/world) chromosomes per rank so the number of step groups (optimizer steps) is minimal.
    Deterministic: ties break on the order of `cost` (dict order = the reference's chromosome order)."""
    names = list(cost)
    order = sorted(range(len(names)), key=lambda i: (-cost[names[i]], i))
    cap = -(-len(names) // world) if names else 0
    load = [0.0] * world
    per_rank: List[List[str]] = [[] for _ in range(world)]
    for i in order:
        cands = [r for r in range(world) if len(per_rank[r]) < cap]
        r = min(cands, key=lambda q: (load[q], q))
        per_rank[r].append(names[i])
        load[r] += cost[names[i]]
    rounds = []
    for k in range(cap):
        rounds.append([per_rank[r][k] if k < len(per_rank[r]) else None for r in range(world)])
    owner = {nm: r for r in range(world) for nm in per_rank[r]}
    return ShardPlan(rounds=rounds, owner=owner, load=load)
