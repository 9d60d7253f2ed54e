"""The engine's multi-rank code path over the real RCCL backend, on the one GPU a test box has: a child process
initialises torch.distributed with backend "nccl" (= RCCL on ROCm) and world_size 1 and runs GCNStage with
force_collectives=True, so the shard plan, the flat-gradient all-reduce, the asynchronous all_gather_into_tensor of
every round's predictions and the statistics/loss all-reduce are all issued through RCCL on the device (the two-rank
test next door has to use gloo: RCCL refuses two ranks on one device).  With one rank every collective is the
identity, so the result must match a single process that runs the same chromosomes' fwd+bwd and optimizer steps with
no process group at all."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

pytestmark = pytest.mark.gpu
EPOCHS = 2


def worker(port, q, group_graph=True):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    import test_gpu_two_rank as T
    from chromegcn_amd.finetune import GCNStage
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    assert dist.get_backend() == "nccl"
    feats, graphs = T.make_data()
    m = T.make_model(dev)
    opt = torch.optim.SGD(m.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-6)
    stage = GCNStage(m, opt, "hic", dev, hip_graphs=True, input_grad=True, group=dist.group.WORLD,
                     cache_input_aggregation=False, force_collectives=True, group_graph=group_graph,
                     prediction_gather="all" if group_graph else "rank0")
    assert stage.multi and stage.world == 1
    assert stage.aux_group is not stage.group   # the eager gathers / statistics all-reduce run on a second RCCL communicator
    stage.load(feats, graphs, defer=True)      # registered only: uploaded when the shard plan hands them to this rank
    assert not stage.chroms and len(stage._meta) == len(feats)
    tot = []
    for _ in range(EPOCHS):
        preds, targets, t = stage.run_split("train")
        tot.append(t)
    assert len(stage.chroms) == len(feats)
    kinds = {k[1] for k in stage._graphs}
    if group_graph:   # the whole step group -- RCCL all-reduce and fused step included -- was captured and replayed
        assert stage._group_graph_ok and "group" in kinds and "fwdbwd" not in kinds, kinds
    else:
        assert "fwdbwd" in kinds and "group" not in kinds, kinds
    pd, td, tv = stage.run_split("valid", to_cpu=False)
    with pytest.raises(RuntimeError):
        stage.train_step(next(iter(feats)))
    dist.barrier()
    torch.cuda.synchronize()
    q.put(({k: v.cpu().numpy() for k, v in m.state_dict().items()}, tot, preds.numpy(), pd.cpu().numpy(), tv))
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("group_graph", [True, False], ids=["step_group_as_one_hip_graph", "captured_fwdbwd_then_collective"])
def test_engine_collectives_run_over_rccl_with_one_rank(group_graph):
    import torch.multiprocessing as mp
    import test_gpu_two_rank as T
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=worker, args=(T.free_port(), q, group_graph))
    p.start()
    sd, tot, preds, vpreds, tv = q.get(timeout=600)
    p.join(120)
    assert p.exitcode == 0

    # the same semantics without torch.distributed: eager fwd+bwd per chromosome, then the optimizer step on the
    # flat buffers (test_gpu_two_rank.emulate with a one-rank plan) -- same kernels in the same order
    ref_sd, ref_tot, ref_vpreds, ref_tv, plan = T.emulate(1, False, EPOCHS)
    assert all(len(r) == 1 for r in plan.rounds)
    print("train loss", tot, ref_tot, "valid loss", tv, ref_tv, "max |dpred|", np.abs(vpreds - ref_vpreds).max())
    np.testing.assert_allclose(tot, ref_tot, rtol=1e-6)
    np.testing.assert_allclose(tv, ref_tv, rtol=1e-6)
    np.testing.assert_allclose(vpreds, ref_vpreds, atol=1e-6, rtol=1e-5)
    for k, v in ref_sd.items():
        np.testing.assert_allclose(sd[k], v, atol=1e-6, rtol=1e-5, err_msg=k)
