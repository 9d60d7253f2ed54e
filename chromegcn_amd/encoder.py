"""Window encoder of BASELINE.json config 5 -- stock PyTorch-ROCm ops (MIOpen convolutions, rocBLAS linear), as the
north star prescribes: the encoder is NOT on the hand-written path, only its OUTPUT CONTRACT matters to it:

    tokens [B, L] int64 in {0..4}  ->  x_feat [B, 128] (pre-ReLU linear output), logits [B, C]

for the window and for its reverse complement (SURVEY.md section 2 'Window encoders' / 'Strand wrapper', Appendix B;
reference: models/WindowModels.py:9-87 'Expecto', models/NonStrandSpecific.py:81-94).  `WindowEncoder` is an
ExPecto-shaped network (Zhou et al. 2018) built from a stage table instead of a literal layer list: three stages of
two valid k=8 convolutions, max-pooling by 4 after the first two, BatchNorm after each, dropout 0 / 0.2 / 0.5, then
Linear(960 * n_pos -> 128), ReLU, BatchNorm, Linear(128 -> C).  Module names and positions follow the reference's so
that a reference Expecto state_dict loads unchanged (`src_word_emb`, `conv_net.<i>`, `linear`, `batch_norm`,
`classifier`); pinned by golden G7 (tests/golden/make_golden.py builds the reference encoder under a seed and records
its outputs; tests/test_encoder.py builds this one under the same seed)."""
from __future__ import annotations

from typing import Dict, Optional, Sequence, Tuple

import torch
import torch.nn as nn

# (channels, pool after the stage, dropout after the stage's BatchNorm)
EXPECTO_STAGES = ((320, True, 0.0), (480, True, 0.2), (960, False, 0.5))
KERNEL, POOL, ALPHABET, FEATURE_WIDTH = 8, 4, 5, 128


def positions_after_convs(seq_length: int, stages=EXPECTO_STAGES, kernel: int = KERNEL, pool: int = POOL) -> int:
    """sequence positions left after the stage stack: every stage removes 2 (kernel - 1), pooling floors by `pool`"""
    n = seq_length
    for _, pooled, _ in stages:
        n -= 2 * (kernel - 1)
        if pooled:
            n //= pool
    return n


class WindowEncoder(nn.Module):
    def __init__(self, nclass: int, seq_length: int = 2000, stages=EXPECTO_STAGES):
        super().__init__()
        self.src_word_emb = nn.Embedding(ALPHABET, ALPHABET)       # 5 symbols A, C, G, T, N -> 5 channels
        layers = []
        c_in = ALPHABET
        for c_out, pooled, p_drop in stages:
            for _ in range(2):
                layers += [nn.Conv1d(c_in, c_out, kernel_size=KERNEL), nn.ReLU(inplace=True)]
                c_in = c_out
            if pooled:
                layers.append(nn.MaxPool1d(kernel_size=POOL, stride=POOL))
            layers.append(nn.BatchNorm1d(c_out))
            if p_drop > 0:
                layers.append(nn.Dropout(p=p_drop))
        self.conv_net = nn.Sequential(*layers)
        self.n_positions = positions_after_convs(seq_length, stages)
        if self.n_positions < 1:
            raise ValueError("seq_length %d is too short for the convolution stack" % seq_length)
        self.flat_width = c_in * self.n_positions
        self.linear = nn.Linear(self.flat_width, FEATURE_WIDTH)
        self.batch_norm = nn.BatchNorm1d(FEATURE_WIDTH)
        self.classifier = nn.Linear(FEATURE_WIDTH, nclass)

    def forward(self, tokens: torch.Tensor):
        """tokens [B, L] -> (x_feat [B,128], logits [B,C], None) -- the reference's return triple"""
        x = self.src_word_emb(tokens).transpose(1, 2)              # [B, 5, L]
        x = self.conv_net(x)
        x_feat = self.linear(x.flatten(1))                         # the node feature the GCN stage consumes
        logits = self.classifier(self.batch_norm(torch.relu(x_feat)))
        return x_feat, logits, None


def complement_table(src_dict: Optional[Dict[str, int]] = None, alphabet: int = ALPHABET) -> torch.Tensor:
    """token -> complementary token (a<->t, c<->g, everything else unchanged; models/NonStrandSpecific.py:30-43).
    src_dict: the reference's letter -> index vocabulary (lower-case keys); default a,c,g,t,n = 0..4."""
    d = src_dict if src_dict is not None else {"a": 0, "c": 1, "g": 2, "t": 3, "n": 4}
    size = max(alphabet, max(d.values()) + 1)
    tab = torch.arange(size, dtype=torch.long)
    for x, y in (("a", "t"), ("c", "g")):
        if x in d and y in d:
            tab[d[x]], tab[d[y]] = d[y], d[x]
    return tab


class StrandPair(nn.Module):
    """Runs the encoder on a window and on its reverse complement; returns (x_feat_f, x_feat_r, mean logits, None, None)
    like models/NonStrandSpecific.py:81-94 in 'mean' mode."""

    def __init__(self, model: nn.Module, src_dict: Optional[Dict[str, int]] = None):
        super().__init__()
        self.model = model
        self.register_buffer("_complement", complement_table(src_dict), persistent=False)

    def reverse_complement(self, tokens: torch.Tensor) -> torch.Tensor:
        return self._complement[tokens.flip(1)]

    def forward(self, src: torch.Tensor, src_dict=None):
        x_f, y_f, _ = self.model(src)
        x_r, y_r, _ = self.model(self.reverse_complement(src))
        return x_f, x_r, (y_f + y_r) / 2, None, None


@torch.no_grad()
def extract_features(pair: StrandPair, tokens: torch.Tensor, targets: torch.Tensor, locs: Sequence, collector,
                     batch_size: int = 64):
    """The `-save_feats` pass of pretrain.py:24-63 (eval mode, shuffle off, batch 64): every batch's (loc, x_f, x_r,
    target) goes into a handoff.FeatureCollector on the device instead of being concatenated on the host."""
    pair.eval()
    for i in range(0, tokens.shape[0], batch_size):
        x_f, x_r, _, _, _ = pair(tokens[i:i + batch_size])
        collector.add(locs[i:i + batch_size], x_f, x_r, targets[i:i + batch_size])
    return collector
