"""The BENCHMARK'S OWN workload against the oracle: one epoch of the GCN stage over the 16 train chromosomes
of the synthetic GM12878-shaped genome at full size (242 908 windows, 250 000 contact pairs each; bench.py's default
workload = BASELINE.json's metric), in the configuration bench.py measures -- reference semantics (all four aggregations,
d loss / d features), the whole split captured as ONE HIP graph, the engine's default statistics mode (accumulate where the
features are in range: every chromosome here), the SGD step fused into the last backward launch, 16 sequential optimizer
steps -- versus oracle.finetune_epoch on the host (finetune.py:29-53 restated), chromosome by chromosome.  Dropout 0 (the
device RNG is not torch's); everything else as bench.py.
Checked: every chromosome's loss, sigmoid(pred) on a strided sample of the rows of every chromosome, every parameter after the
epoch at atol = rtol = 1e-4, the BatchNorm running statistics at 1e-5, num_batches_tracked; then the evaluation split
(finetune.py:10-13: eval mode) on the three valid chromosomes.  The full-size single-chromosome cases of
test_gpu_fullsize_oracle.py tie the kernels to the oracle step by step; this one ties the EPOCH the headline number times."""
import numpy as np
import pytest
import torch

import chromegcn_amd as C
from chromegcn_amd import synth
from chromegcn_amd.finetune import GCNStage
from oracle import chromegcn_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
D = 128


@pytest.mark.timeout(1800)
def test_one_epoch_of_the_train_genome_matches_the_oracle():
    threads = torch.get_num_threads()
    torch.set_num_threads(8)     # (the host oracle's fp32 sums depend on the thread count: test_gpu_fullsize_oracle.py)
    try:
        _epoch_case()
    finally:
        torch.set_num_threads(threads)


def _epoch_case():
    names = [c for c in synth.HG19_LEN if synth.split_of(c) == "train"]
    valid = [c for c in synth.HG19_LEN if synth.split_of(c) == "valid"]
    assert len(names) == 16 and len(valid) == 3
    feats, graphs = {}, {}
    for c in names + valid:
        feats[c], graphs[c] = synth.synthetic_chromosome(c, d=D)
    assert sum(feats[c]["forward"].shape[0] for c in names) == 242908
    torch.manual_seed(0)
    orc = O.GatedGCNOracle(D, synth.N_LABELS, 0.0, 2)
    with torch.no_grad():   # the reference init (gain 0.02) leaves tanh / the gates in their linear range: a conditioned scale-up
        for k in (1, 2):
            getattr(orc, "GC%d" % k).weight.mul_(8.0)
            getattr(orc, "W%d" % k).weight.mul_(3.0)
    model = C.ChromeGCN(D, D, synth.N_LABELS, 0.0, True, 2)
    model.load_state_dict(orc.state_dict())
    model.to(DEV)
    opt = torch.optim.SGD(model.parameters(), lr=0.25, momentum=0.9, weight_decay=1e-6)
    # bench.py's stage, argument by argument (single rank)
    stage = GCNStage(model, opt, "hic", DEV, hip_graphs=True, input_grad=True, cache_input_aggregation=False)
    for c in names:
        stage.add_chromosome(c, feats[c], graphs[c])
    assert stage.epoch_graph and all(stage.chroms[c].stat_acc for c in names)      # one graph per split, accumulate mode
    preds, targets, total = stage.run_split("train", names, to_cpu=False)
    torch.cuda.synchronize()
    assert (tuple(names), "epoch") in stage._graphs          # the whole split was ONE captured graph, replayed once
    rows = stage._arena["rows"]
    loss_hip = {c: float(stage._arena["slots"][c]["loss"].item()) for c in names}
    preds = preds.cpu().numpy()
    # ---- the oracle: the same epoch on the host, one chromosome at a time (finetune.py:29 iterates the dict in order)
    oopt = O.make_sgd(orc, 0.25)
    cache = {}
    worst_p, worst_l, off = 0.0, 0.0, 0
    for c in names:
        po, _, lo = O.finetune_epoch(orc, {c: feats[c]}, {c: graphs[c]}, oopt, "train", "hic", adj_cache=cache)
        n = feats[c]["forward"].shape[0]
        assert abs(loss_hip[c] - lo) <= 1e-4 + 1e-4 * abs(lo), (c, loss_hip[c], lo)
        worst_l = max(worst_l, abs(loss_hip[c] - lo))
        r0, r1 = rows[c]
        assert (r0, r1) == (off, off + n)
        a, b = preds[r0:r1:37], po.numpy()[::37]
        np.testing.assert_allclose(a, b, atol=1e-4, rtol=1e-4, err_msg="sigmoid(pred) of %s" % c)
        worst_p = max(worst_p, float(np.abs(a - b).max()))
        off += n
    assert abs(total - sum(loss_hip.values())) <= 1e-4 * len(names)
    osd = orc.state_dict()
    worst = {}
    for k, v in model.state_dict().items():
        a, b = v.detach().cpu().numpy(), osd[k].numpy()
        if k.endswith("num_batches_tracked"):
            assert int(a) == int(b) == 2 * len(names), (k, a, b)     # one BatchNorm call per strand and chromosome
            continue
        tol = 1e-5 if "running" in k else 1e-4
        np.testing.assert_allclose(a, b, atol=tol, rtol=tol, err_msg=k)
        worst[k] = float(np.abs(a - b).max())
    print("\nepoch vs oracle: worst |loss diff| %.2e, worst |prob diff| %.2e, parameters: %s" % (
        worst_l, worst_p, ", ".join("%s %.1e" % kv for kv in sorted(worst.items()))))
    # ---- the evaluation split (eval mode, running statistics) on the three valid chromosomes
    for c in valid:
        stage.add_chromosome(c, feats[c], graphs[c])
    pv, tv, lv = stage.run_split("valid", valid, to_cpu=True)
    po, to_, lo = O.finetune_epoch(orc, {c: feats[c] for c in valid}, {c: graphs[c] for c in valid}, None, "valid", "hic")
    assert abs(lv - lo) <= 1e-4 * len(valid) + 1e-4 * abs(lo), (lv, lo)
    np.testing.assert_allclose(pv.numpy()[::53], po.numpy()[::53], atol=1e-4, rtol=1e-4)
    assert torch.equal(tv.float().cpu(), to_)
