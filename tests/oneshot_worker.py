"""Child process of tests/test_gpu_oneshot_allreduce.py: one rank of a one-shot all-reduce between processes that share cuda:0.
Started fresh (before any GPU call) with RANK / WORLD_SIZE / MASTER_* in the environment; writes <out>.<rank>.pt."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as td  # noqa: E402

from pime_amd import dist as pdist  # noqa: E402

out, n = sys.argv[1], int(sys.argv[2])
mode = sys.argv[3] if len(sys.argv) > 3 else "parity"
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda", 0)      # every rank on the one GPU of the box; gloo carries the handle exchange and the reference sums
torch.cuda.set_device(0)
td.init_process_group(backend="gloo", rank=rank, world_size=world)
ar = pdist.OneShotAllReduce(rank, world, n, dev)
assert ar.fine_grained or ar.same_device

if mode == "absent":   # rank 1 maps the regions and then never calls: rank 0's launch must give up and the check must raise
    import time
    if rank == 1:
        time.sleep(8.0)              # keep the region mapped while rank 0 spins out (~2 s) and reports
        sys.exit(0)
    x = torch.ones(n, device=dev)
    ar(x)
    torch.cuda.synchronize()
    dp = pdist.DataParallel(rank, world, 0, dev)
    dp._oneshot[n] = ar
    dp.check()                       # raises PimeError: the process exits non-zero with the message on stderr
    sys.exit(0)                      # (not reached)


def vec(k):
    g = torch.Generator().manual_seed(1000 * k + rank)
    return torch.randn(n, generator=g) * (1.0 + k)


got, want = [], []
for k in range(7):                      # eager calls: both parities several times
    x = vec(k).to(dev)
    ar(x)
    torch.cuda.synchronize()
    ref = vec(k)
    td.all_reduce(ref, op=td.ReduceOp.SUM)
    got.append(x.cpu())
    want.append(ref / world)
static = torch.zeros(n, device=dev)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
static.copy_(vec(100).to(dev))
with torch.cuda.graph(g):               # the launch is capturable: the call sequence number lives in device memory
    ar(static)
for k in range(101, 105):               # (the capture itself launched nothing)
    static.copy_(vec(k).to(dev))
    g.replay()
    torch.cuda.synchronize()
    ref = vec(k)
    td.all_reduce(ref, op=td.ReduceOp.SUM)
    got.append(static.cpu().clone())
    want.append(ref / world)
status = ar.status()
td.barrier()
torch.save({"got": got, "want": want, "status": status}, f"{out}.{rank}.pt")
ar.close()
td.destroy_process_group()
