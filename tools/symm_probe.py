"""Probe (GPU box): does torch's CUDA symmetric memory work here between two processes sharing cuda:0?"""
import os, socket, sys, traceback
import torch
import torch.multiprocessing as mp


def worker(rank, world, port, backend):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    try:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        import torch.distributed._symmetric_memory as sm
        print(rank, "backend", backend, "symm backends:", getattr(sm, "get_backend", lambda d: None)(dev), flush=True)
        buf = sm.empty(1024, dtype=torch.float32, device=dev)
        buf.fill_(float(rank + 1))
        hdl = sm.rendezvous(buf, dist.group.WORLD.group_name)
        print(rank, "rendezvous ok: rank", hdl.rank, "world", hdl.world_size, flush=True)
        hdl.barrier()
        peers = [hdl.get_buffer(r, (1024,), torch.float32) for r in range(world)]
        tot = peers[0] + peers[1]
        torch.cuda.synchronize()
        print(rank, "peer-read sum", float(tot[0]), flush=True)
        hdl.barrier()
        try:
            out = torch.ops.symm_mem.one_shot_all_reduce(buf, "sum", dist.group.WORLD.group_name)
            torch.cuda.synchronize()
            print(rank, "one_shot_all_reduce", float(out[0]), flush=True)
        except Exception as e:
            print(rank, "one_shot_all_reduce failed:", repr(e)[:300], flush=True)
    except Exception:
        print(rank, "FAILED", flush=True)
        traceback.print_exc()
    try:
        dist.destroy_process_group()
    except Exception:
        pass


if __name__ == "__main__":
    backend = sys.argv[1] if len(sys.argv) > 1 else "gloo"
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    ps = [ctx.Process(target=worker, args=(r, 2, port, backend)) for r in range(2)]
    for p in ps: p.start()
    for p in ps: p.join(180)
    print("exit codes", [p.exitcode for p in ps])
