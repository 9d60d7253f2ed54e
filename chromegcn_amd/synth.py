"""Seeded synthetic stand-ins for the data the hot path consumes (the real GM12878/K562 files need
a 13 GB download, README.md:16-21).  Shapes follow SURVEY.md section 8(d) / Appendix C:

  * graphs: symmetric {0,1} float64 CSR with zero diagonal -- the on-disk contract of
    data/7create_graph_new.py:108-120 -- with `pairs` undirected contact pairs per chromosome
    (hic_edges/2, data/create_data.py:27, data/7create_graph_new.py:168);
  * node features / targets: the chrom_feature_dict contract of utils/util_methods.py:183-199."""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np
import scipy.sparse as sp
import torch

# hg19 autosome lengths (UCSC hg19.chrom.sizes; the reference reads the same file, data/create_data.py:32)
HG19_LEN = {
    "chr1": 249250621, "chr2": 243199373, "chr3": 198022430, "chr4": 191154276, "chr5": 180915260,
    "chr6": 171115067, "chr7": 159138663, "chr8": 146364022, "chr9": 141213431, "chr10": 135534747,
    "chr11": 135006516, "chr12": 133851895, "chr13": 115169878, "chr14": 107349540, "chr15": 102531392,
    "chr16": 90354753, "chr17": 81195210, "chr18": 78077248, "chr19": 59128983, "chr20": 63025520,
    "chr21": 48129895, "chr22": 51304566,
}
VALID_CHROMS = ["chr3", "chr12", "chr17"]  # data/create_data.py:44
TEST_CHROMS = ["chr1", "chr8", "chr21"]    # data/create_data.py:45
PEAK_FRACTION = 0.12   # assumed share of 1 kb windows overlapping a peak
PAIRS_PER_CHROM = 250000  # -hicsize 500000 => 250k undirected pairs (README.md:45)
N_LABELS = 103         # assumed label count (data dependent in the reference, main.py:35)


def chrom_nodes(chrom: str, fraction: float = PEAK_FRACTION) -> int:
    return int(round(fraction * HG19_LEN[chrom] / 1000.0))


def split_of(chrom: str) -> str:
    return "valid" if chrom in VALID_CHROMS else "test" if chrom in TEST_CHROMS else "train"


def chrom_seed(chrom: str) -> int:
    return int(chrom[3:])


GENERATORS = ("uniform", "hic_like", "hub")


def contact_graph(n: int, pairs: int, seed: int, hic_like=False) -> sp.csr_matrix:
    """`pairs` undirected contact pairs on n windows.  hic_like: False / "uniform" = uniform-random pairs; True /
    "hic_like" = pairs whose genomic distance |i-j| follows a truncated 1/k law, so contacts concentrate near the
    diagonal like real Hi-C; "hub" = a top-K-style graph: the reference keeps the `hic_edges/2` HIGHEST-scoring pairs of
    a chromosome (data/7create_graph_new.py:93-104), and normalised contact scores are dominated by a few
    high-coverage bins, so the kept edges pile up on them -- a heavy-tailed degree distribution with hubs of thousands
    of neighbours.  Modelled as: 8 hubs with 2 000 ... min(10 000, n/2) neighbours each (uniform random partners),
    the rest of the budget drawn with power-law endpoint propensities (w_i ~ rank^-0.6, random rank order) -- degrees
    from 1 to 10^4 on the same edge budget as the other generators."""
    kind = hic_like if isinstance(hic_like, str) else ("hic_like" if hic_like else "uniform")
    if kind not in GENERATORS:
        raise ValueError("unknown contact generator %r" % (kind,))
    rng = np.random.RandomState(seed)
    if kind == "hic_like":
        kmax = max(2, n - 1)
        u = rng.random_sample(pairs)
        dist = np.clip(np.floor(np.exp(u * math.log(kmax))).astype(np.int64), 1, n - 1)
        i = (rng.random_sample(pairs) * (n - dist)).astype(np.int64)
        j = i + dist
    elif kind == "hub":
        n_hubs = min(8, max(1, n // 64))
        hubs = rng.choice(n, n_hubs, replace=False)
        hi_deg = max(2, min(10000, n // 2))
        lo_deg = max(1, min(2000, hi_deg // 2))
        ii, jj = [], []
        for h in hubs:
            deg = int(rng.randint(lo_deg, hi_deg + 1))
            nb = rng.choice(n, deg, replace=False)
            ii.append(np.full(deg, h, dtype=np.int64))
            jj.append(nb.astype(np.int64))
        used = sum(a.size for a in ii)
        rest = max(0, pairs - used)
        w = np.arange(1, n + 1, dtype=np.float64) ** -0.6
        w = w[rng.permutation(n)]
        w /= w.sum()
        ii.append(rng.choice(n, rest, p=w).astype(np.int64))
        jj.append(rng.choice(n, rest, p=w).astype(np.int64))
        i, j = np.concatenate(ii), np.concatenate(jj)
    else:
        i = rng.randint(0, n, pairs)
        j = rng.randint(0, n, pairs)
    keep = i != j
    i, j = i[keep], j[keep]
    a = sp.coo_matrix((np.ones(i.size), (i, j)), shape=(n, n)).tocsr()
    a = a + a.T
    a.data[:] = 1.0
    a.sort_indices()
    return sp.csr_matrix(a, dtype=np.float64)


def chrom_features(n: int, d: int, n_labels: int, seed: int, positive_rate: float = 0.05) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    return {"forward": torch.randn(n, d, generator=g), "backward": torch.randn(n, d, generator=g),
            "target": (torch.rand(n, n_labels, generator=g) < positive_rate).float()}


def synthetic_chromosome(chrom: str, d: int = 128, n_labels: int = N_LABELS, pairs: int = PAIRS_PER_CHROM,
                         hic_like: bool = False, n: int = None) -> Tuple[Dict[str, torch.Tensor], sp.csr_matrix]:
    n = chrom_nodes(chrom) if n is None else n
    seed = chrom_seed(chrom)
    return chrom_features(n, d, n_labels, 1000 + seed), contact_graph(n, pairs, seed, hic_like)


def config1(d: int = 128, n_labels: int = N_LABELS):
    """BASELINE.json configs[0]: one chromosome, 5k nodes, ~1 % density."""
    n = 5000
    return chrom_features(n, d, n_labels, 1000), contact_graph(n, int(0.01 * n * n / 2), 0)
