#!/usr/bin/env python3
"""The training head in accumulate mode against records mode, in isolation (tuning tool): k_head_fused_rs alone
(cgcn_debug_head_train_phases, phase 2) at the genome's mean chromosome size; accumulate mode is timed as (zero the totals +
head) - (zero the totals), because every launch adds to them.  CHROMEGCN_LIB selects a (decomposition) variant."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from chromegcn_amd import _lib
from tools.kbench import timeit


def main():
    dev = torch.device("cuda"); lib = _lib.load()
    d, S, C = 128, 2, 103
    for n in [int(a) for a in (sys.argv[1:] or ["5776", "15182", "29910"])]:
        x = torch.randn(S, n, d, device=dev)
        bn_w = torch.rand(d, device=dev) + 0.5; bn_b = torch.randn(d, device=dev) * 0.1
        rm = torch.zeros(d, device=dev); rv = torch.ones(d, device=dev); nbt = torch.zeros(1, dtype=torch.int64, device=dev)
        W = torch.randn(C, d, device=dev) / d ** 0.5; b = torch.zeros(C, device=dev)
        tgt = (torch.rand(n, C, device=dev) < 0.05).float()
        rng = torch.tensor([1, 0], dtype=torch.int64, device=dev)
        wsb = lib.cgcn_head_workspace_bytes(n, S, d, C); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        probs = torch.empty(n, C, device=dev); loss = torch.empty(1, device=dev)
        sm = torch.empty(S, d, device=dev); si = torch.empty(S, d, device=dev)
        P = _lib.ptr; st = _lib.stream_ptr
        rows = ctypes.c_int(0)
        tiles = lib.cgcn_layer_fwd_colstats_plan(n, S, d, _lib.COLSTATS_ACCUMULATE, ctypes.byref(rows))
        cs = torch.zeros(tiles, S, d, 2, device=dev)
        def head(ph, acc):
            return lib.cgcn_debug_head_train_phases(st(), n, S, d, C, P(x), P(bn_w), P(bn_b), P(rm), P(rv), P(nbt), 0.1, 1e-5, P(W), P(b), P(tgt), 0.2, P(rng),
                                                    P(probs), P(loss), P(sm), P(si), P(cs) if acc else None, tiles if acc else 0, -1 if acc else 0, P(ws), wsb, ph)
        assert head(7, False) == 0
        t_rec = timeit(lambda: head(2, False), reps=100)
        def pair():
            cs.zero_()
            return head(2, True)
        assert pair() == 0
        t_zero = timeit(lambda: cs.zero_(), reps=100)
        t_pair = timeit(pair, reps=100)
        print(json.dumps({"n": n, "records_us": round(t_rec, 2), "accumulate_us": round(t_pair - t_zero, 2), "zero_us": round(t_zero, 2)}))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
