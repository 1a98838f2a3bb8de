"""Can RCCL run two ranks on ONE device on this stack?  (If yes, the data-parallel tests can use the real backend.)"""
import os, sys
import torch, torch.distributed as dist, torch.multiprocessing as mp

def worker(rank, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    try:
        dist.init_process_group("nccl", rank=rank, world_size=2)
        x = torch.full((4,), float(rank + 1), device="cuda")
        dist.all_reduce(x)
        torch.cuda.synchronize()
        print("rank", rank, "all_reduce ->", x.tolist(), flush=True)
        out = torch.zeros(2, device="cuda", dtype=torch.int32); inp = torch.arange(4, device="cuda", dtype=torch.int32) + 10 * rank
        dist.reduce_scatter_tensor(out, inp)
        torch.cuda.synchronize()
        print("rank", rank, "reduce_scatter ->", out.tolist(), flush=True)
        dist.destroy_process_group()
    except Exception as e:
        print("rank", rank, "FAILED:", str(e)[:300], flush=True)

if __name__ == "__main__":
    mp.start_processes(worker, args=(29611,), nprocs=2, join=True, start_method="spawn")
