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


def contact_graph(n: int, pairs: int, seed: int, hic_like: bool = False) -> sp.csr_matrix:
    """Uniform-random pairs, or (hic_like) pairs whose genomic distance |i-j| follows a truncated
    1/k law so contacts concentrate near the diagonal like real Hi-C."""
    rng = np.random.RandomState(seed)
    if hic_like:
        kmax = max(2, n - 1)
        u = rng.random_sample(pairs)
        dist = np.clip(np.floor(np.exp(u * math.log(kmax))).astype(np.int64), 1, n - 1)
        i = (rng.random_sample(pairs) * (n - dist)).astype(np.int64)
        j = i + dist
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
