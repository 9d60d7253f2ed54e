"""The real multi-rank engine on the hardware there is: two fresh child processes share cuda:0 and run the HIP
GCNStage.run_split("train") -- shard plan, captured fwd+bwd HIP graphs, one flat-gradient all-reduce per step group,
fused SGD step, BatchNorm-statistics sync, split-end prediction gather -- over five chromosomes with the gloo backend
(RCCL refuses two ranks on one device), against a single-process emulation on the same GPU that averages the same
chromosomes' gradients per step group (the semantics DESIGN.md section 6 documents).  Parameters: 1e-4; rank 0 vs
rank 1: bitwise.  Children are started with the `spawn` method (a new interpreter each; the parent is never re-exec'd)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu
D, NC = 128, 23
SIZES = {"chr2": 900, "chr4": 610, "chr5": 1200, "chr6": 750, "chr7": 330}
EPOCHS = 2
GENOME = False   # set by the genome-scale test (module globals travel to the spawned children through the args)


def make_data(genome=False):
    from chromegcn_amd import synth
    feats, graphs = {}, {}
    if genome:  # BASELINE.json configs[2]: the 16 train chromosomes of the synthetic GM12878-shaped genome, full size
        for c in synth.HG19_LEN:
            if synth.split_of(c) == "train":
                f, g = synth.synthetic_chromosome(c, d=D, n_labels=NC)
                feats[c], graphs[c] = f, g
        return feats, graphs
    for i, (c, n) in enumerate(SIZES.items()):
        feats[c] = synth.chrom_features(n, D, NC, 50 + i, positive_rate=0.2)
        graphs[c] = synth.contact_graph(n, 6 * n, 60 + i)
    return feats, graphs


def make_model(dev):
    import chromegcn_amd as C
    torch.manual_seed(3)
    m = C.ChromeGCN(D, D, NC, 0.0, True, 2)
    with torch.no_grad():
        m.GC1.weight.mul_(40)
        m.GC2.weight.mul_(40)
    return m.to(dev)


def worker(rank, world, port, q, genome=False, epochs=EPOCHS):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    from chromegcn_amd.finetune import GCNStage
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    feats, graphs = make_data(genome)
    rows = sum(f["forward"].shape[0] for f in feats.values())
    m = make_model("cuda:0")
    opt = torch.optim.SGD(m.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-6)
    stage = GCNStage(m, opt, "hic", "cuda:0", hip_graphs=True, input_grad=True, group=dist.group.WORLD,
                     cache_input_aggregation=False)
    stage.load(feats, graphs, defer=True)   # a rank uploads only what the shard plan gives it
    tot = []
    for _ in range(epochs):
        preds, targets, t = stage.run_split("train")
        tot.append(t)
        assert preds.shape == (rows, NC) and targets.shape == preds.shape
    owned = sum(1 for g in __import__("chromegcn_amd.dist", fromlist=["plan_shards"]).plan_shards(
        {c: stage._meta[c][2] for c in feats}, world).owner.values() if g == rank)
    assert len(stage.chroms) == owned < len(feats)
    pd, td, tv = stage.run_split("valid", to_cpu=False)          # device-resident gather path
    assert pd.is_cuda and pd.shape == (rows, NC)
    if genome:   # 243 k x 23 predictions: send checksums and a strided sample instead of everything
        preds, pdc = preds[::97].contiguous(), pd[::97].cpu()
    else:
        pdc = pd.cpu()
    q.put((rank, {k: v.cpu().numpy() for k, v in m.state_dict().items()}, tot, preds.numpy(), pdc.numpy(), tv,
           m._rng_state.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def emulate(world, genome=False, epochs=EPOCHS):
    """single process, same GPU, same kernels: per step group, run fwd+bwd of each member eagerly, average the flat
    gradient buffers, take one fused optimizer step"""
    from chromegcn_amd.dist import plan_shards
    from chromegcn_amd.finetune import GCNStage
    feats, graphs = make_data(genome)
    m = make_model("cuda")
    opt = torch.optim.SGD(m.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-6)
    stage = GCNStage(m, opt, "hic", "cuda", hip_graphs=False, input_grad=True, cache_input_aggregation=False)
    stage.load(feats, graphs)
    plan = plan_shards({c: stage.chroms[c].cost for c in feats}, world)
    m.train()
    tot = []
    for _ in range(epochs):
        t = 0.0
        for group in plan.rounds:
            names = [g for g in group if g is not None]
            acc = None
            for nm in names:
                stage._ensure_flat_grad()
                loss, _, _ = stage._fwd_bwd(stage.chroms[nm])
                t += loss.item()
                acc = stage._flat_grad.clone() if acc is None else acc + stage._flat_grad
            stage._flat_grad.copy_(acc)
            stage._optimizer_step(1.0 / len(names))
        tot.append(t)
    preds, _, ev = stage.run_split("valid")
    if genome:
        preds = preds[::97].contiguous()
    return {k: v.cpu().numpy() for k, v in m.state_dict().items()}, tot, preds.numpy(), ev, plan


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("world,genome", [(2, False), (2, True), (8, False), (8, True)],
                         ids=["2_ranks_five_small_chromosomes", "2_ranks_full_size_train_genome_configs2",
                              "8_ranks_five_small_chromosomes_idle_ranks", "8_ranks_full_size_train_genome_two_step_groups"])
def test_ranks_on_one_gpu_match_the_gradient_averaging_emulation(world, genome):
    """genome=True is BASELINE.json configs[2]'s code path at full size -- the 16 train chromosomes (242 908 windows,
    250 000 contact pairs each) sharded by plan_shards -- on ranks that share the one GPU a test box has (gloo: RCCL refuses
    two ranks per device).  world 8 is the driver's 8-GPU shape: two chromosomes per rank = exactly two step groups, seven
    ranks' rows gathered; with five small chromosomes three ranks hold nothing at all."""
    import torch.multiprocessing as mp
    epochs = 1 if genome else EPOCHS
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, q, genome, epochs)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=900) for _ in range(world)]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    ref_sd, ref_tot, ref_preds, ref_ev, plan = emulate(world, genome, epochs)
    if genome:
        assert len(plan.rounds) == 16 // world and all(g is not None for r in plan.rounds for g in r)
    elif world == 2:
        assert len(plan.rounds) == 3 and all(any(g is None for g in r) is (i == 2) for i, r in enumerate(plan.rounds))
    else:
        assert len(plan.rounds) == 1 and sum(g is None for g in plan.rounds[0]) == world - 5
    results.sort(key=lambda r: r[0])
    for rank, sd, tot, preds, preds_dev, ev, rng in results:
        # dropout RNG state: the seed is untouched by the statistics sync, the step counter advanced once per step group
        assert int(rng[0]) == 0x5DEECE66D and int(rng[1]) == len(plan.rounds) * epochs, rng
        np.testing.assert_allclose(tot, ref_tot, rtol=1e-4, atol=1e-5)
        for k in ref_sd:
            if "running" in k or "num_batches" in k:
                continue  # per-rank BatchNorm statistics are averaged across ranks (documented deviation)
            np.testing.assert_allclose(sd[k], ref_sd[k], rtol=1e-4, atol=1e-4, err_msg=k)
        np.testing.assert_array_equal(preds_dev.shape, ref_preds.shape)
        # (evaluation predictions are not compared with the emulation: they depend on the BatchNorm running statistics,
        #  which the multi-rank run averages across ranks -- the documented deviation; rank 0 == rank 1 is checked below)
    # every rank holds the identical model (incl. the averaged BatchNorm buffers) and identical full predictions
    for other in results[1:]:
        for k in results[0][1]:
            np.testing.assert_array_equal(results[0][1][k], other[1][k], err_msg=k)
        np.testing.assert_array_equal(results[0][3], other[3])
        np.testing.assert_array_equal(results[0][4], other[4])
        assert abs(results[0][5] - other[5]) < 1e-6
