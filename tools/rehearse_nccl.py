"""Does RCCL accept two ranks on ONE device?  If it does, the exchange steps of cimg/shard.py (gather_chunks / scatter_chunks /
broadcast_sizes: RCCL send / recv under "nccl") run once with device tensors and are compared byte for byte; if it does not, the
refusal is what this prints.  One GPU is all the pool hands out, so this is the nearest thing to a multi-GPU run there is.
usage (two ranks, one card):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/rehearse_nccl.py"""
import os, sys, time, traceback
sys.path[:0] = [os.path.join(os.getcwd(), "compressed-image_amd"), os.path.join(os.getcwd(), "tests")]
import numpy as np, torch
import torch.distributed as dist
from cimg import shard

rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
torch.cuda.set_device(0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
try:
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
    t = torch.ones(1, device="cuda") * (rank + 1)
    dist.all_reduce(t)                                     # the communicator is built here: a duplicate-device refusal shows up now
    torch.cuda.synchronize()
    assert int(t.item()) == world * (world + 1) // 2
except Exception as ex:                                    # noqa: BLE001 -- whatever RCCL says is the result
    if rank == 0:
        print("nccl rehearsal: RCCL REFUSED %d ranks on one device: %s: %s" % (world, type(ex).__name__, str(ex).splitlines()[0][:300]))
    os._exit(0)

# chunks of random sizes, owned round-robin; every rank's chunks in a device buffer with gaps
rng = np.random.default_rng(7)
n_items, per_group, stride = 24, 3, 70000
sizes = rng.integers(1000, 65536, n_items).astype(np.int64)
mine = shard.partition(n_items, world, rank, per_group)
blobs = {int(g): np.random.default_rng(1000 + int(g)).integers(0, 256, int(sizes[g]), dtype=np.uint8) for g in range(n_items)}
local = np.zeros(len(mine) * stride, np.uint8)
for k, g in enumerate(mine):
    local[k * stride:k * stride + sizes[g]] = blobs[int(g)]
d_local = torch.from_numpy(local).cuda()
sizes_all = shard.gather_sizes(dist, mine, sizes[mine], n_items, device="cuda")
assert sizes_all.tolist() == sizes.tolist()
ok = True
t0 = time.perf_counter()
got = shard.gather_chunks(dist, world, rank, mine, d_local, [k * stride for k in range(len(mine))], sizes_all, n_items, dst=0,
                          items_per_group=per_group, device="cuda", as_tensor=True)
torch.cuda.synchronize()
t_gather = time.perf_counter() - t0
if rank == 0:
    whole, offs, szs = got
    w = whole.cpu().numpy()
    for g in range(n_items):
        ok &= w[int(offs[g]):int(offs[g]) + int(szs[g])].tobytes() == blobs[g].tobytes()
    whole_dev, offs_dev = whole, offs
else:
    whole_dev, offs_dev = None, None
# the decode mirror: rank 0 holds everything, the sizes travel first, every owner gets ITS chunks back
s2 = shard.broadcast_sizes(dist, sizes_all if rank == 0 else None, n_items, src=0, device="cuda")
t0 = time.perf_counter()
back, loff, idx = shard.scatter_chunks(dist, world, rank, whole_dev, offs_dev, s2, n_items, src=0, items_per_group=per_group, device="cuda")
torch.cuda.synchronize()
t_scatter = time.perf_counter() - t0
b = back.cpu().numpy()
for k, g in enumerate(idx):
    ok &= b[int(loff[k]):int(loff[k]) + int(s2[g])].tobytes() == blobs[int(g)].tobytes()
flag = torch.tensor([1 if ok else 0], device="cuda")
dist.all_reduce(flag, op=dist.ReduceOp.MIN)
if rank == 0:
    print("nccl rehearsal: RCCL accepted %d ranks on one device; gather_chunks + broadcast_sizes + scatter_chunks over RCCL send/recv: %s "
          "(%d chunks, %d bytes; gather %.1f ms, scatter %.1f ms -- one card, the times mean nothing)" % (
              world, "bytes identical on every rank" if int(flag.item()) == 1 else "BYTES DIFFER", n_items, int(sizes.sum()), t_gather * 1e3, t_scatter * 1e3))
dist.barrier()
dist.destroy_process_group()
