"""N > 1 path on CPU: two gloo ranks shard the chunks of an image, each runs the codec on its own items
(the oracle stands in for the engine here -- there is no GPU in this container), and the tiny
exchanges of cimg/shard.py reproduce the single-process result.  No data-path collective exists.
"""
import os
import socket

import numpy as np
import pytest

from cimg import shard, synth


def test_partition_covers_every_item_once():
    for n, world, per in [(32, 8, 8), (32, 2, 8), (256, 8, 8), (7, 3, 2), (5, 8, 1), (3, 4, 8)]:
        seen = np.concatenate([shard.partition(n, world, r, per) for r in range(world)])
        assert sorted(seen.tolist()) == list(range(n)), (n, world, per)
    assert shard.partition(32, 4, 3, 8).tolist() == list(range(24, 32))           # config 2 on 4 GPUs: channel 3 -> rank 3
    assert shard.partition(32, 8, 3, 8).tolist() == [3, 11, 19, 27]                # config 2 on 8 GPUs: 4 channels < 8 ranks -> chunk-granular, 4 chunks each
    assert shard.partition(2048, 8, 1, 8).tolist()[:9] == list(range(8, 16)) + [72]  # config 4: 256 channels round-robin
    with pytest.raises(ValueError):
        shard.partition(4, 2, 2)


def _worker(rank, world, port, out):
    import torch.distributed as dist
    import _oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        chans = [synth.tiled_channel(np.uint16, 512, 256, c=c) for c in range(3)]       # 3 channels x 4 chunks of 64 KiB
        raw = np.concatenate([c.view(np.uint8).ravel() for c in chans])
        chunk = 65536
        n = raw.size // chunk
        mine = shard.partition(n, world, rank, items_per_group=4)
        p = O.cparams(2)
        sizes, blobs = [], []
        for i in mine:
            r, c = O.compress(p, raw[i * chunk:(i + 1) * chunk], destsize=chunk + 32)
            back = O.decompress(c)[1]
            assert back.tobytes() == raw[i * chunk:(i + 1) * chunk].tobytes()
            sizes.append(r)
            blobs.append(np.frombuffer(c[:r], dtype=np.uint8))
        full = shard.gather_sizes(dist, mine, sizes, n)
        t = shard.max_over_ranks(dist, 1.0 + rank)
        # the exchange step: every chunk ends up on rank 0, in global order, byte for byte
        stride = chunk + 64
        local = np.zeros(len(mine) * stride, dtype=np.uint8)
        for k, b in enumerate(blobs):
            local[k * stride:k * stride + b.size] = b
        got = shard.gather_chunks(dist, world, rank, mine, local, [k * stride for k in range(len(mine))], full, n,
                                  dst=0, items_per_group=4)
        # the same exchange handing back ONE packed tensor + offsets (what bench.py --config 4 uses): nothing is padded to
        # the largest rank -- rank 0 owns 8 chunks, rank 1 four -- and the bytes are those of the list form
        packed = shard.gather_chunks(dist, world, rank, mine, local, [k * stride for k in range(len(mine))], full, n,
                                     dst=0, items_per_group=4, as_tensor=True)
        if rank == 0:
            whole, offs, szs = packed
            assert whole.numel() == int(np.asarray(full).sum()) and szs.tolist() == list(full)
            for i in range(n):
                assert whole[int(offs[i]):int(offs[i]) + int(szs[i])].numpy().tobytes() == got[i].tobytes()
        else:
            assert packed is None
        digest = None
        if rank == 0:
            assert len(got) == n and all(g is not None for g in got)
            digest = [int(np.frombuffer(g.tobytes(), dtype=np.uint8).astype(np.uint64).sum()) * 1000003 + g.size for g in got]
            for i in range(n):                                    # every gathered chunk decodes to its pixels
                assert O.decompress(got[i])[1].tobytes() == raw[i * chunk:(i + 1) * chunk].tobytes()
        else:
            assert got is None
        # ---- the decode mirror: rank 0 holds the whole image's chunks (just gathered), every owner gets ITS chunks back, exact sizes,
        # and decodes them; the sizes table travels first (a reader of a stored image knows it on one rank only)
        sizes_b = shard.broadcast_sizes(dist, full if rank == 0 else None, n, src=0)
        assert sizes_b.tolist() == list(full)
        if rank == 0:
            whole, offs, _ = packed
            mine_buf, mine_off, mine_idx = shard.scatter_chunks(dist, world, rank, whole, offs, sizes_b, n, src=0, items_per_group=4)
        else:
            mine_buf, mine_off, mine_idx = shard.scatter_chunks(dist, world, rank, None, None, sizes_b, n, src=0, items_per_group=4)
        assert mine_idx.tolist() == mine.tolist()
        assert mine_buf.numel() == int(sizes_b[mine].sum())                 # nothing padded
        for k, g in enumerate(mine_idx):
            c = mine_buf[int(mine_off[k]):int(mine_off[k]) + int(sizes_b[g])].numpy()
            assert c.tobytes() == blobs[k].tobytes()                         # the chunk this rank compressed came back byte for byte
            assert O.decompress(c)[1].tobytes() == raw[g * chunk:(g + 1) * chunk].tobytes()
        out.put((rank, mine.tolist(), full.tolist(), t, digest))
    finally:
        dist.destroy_process_group()


def test_two_gloo_ranks_reproduce_single_process_sizes():
    import torch.multiprocessing as mp
    import _oracle as O
    O.lib()                                    # build before forking
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    chans = [synth.tiled_channel(np.uint16, 512, 256, c=c) for c in range(3)]
    raw = np.concatenate([c.view(np.uint8).ravel() for c in chans])
    want = [O.compress(O.cparams(2), raw[i * 65536:(i + 1) * 65536], destsize=65536 + 32)[0] for i in range(12)]
    assert res[0][1] == [0, 1, 2, 3, 8, 9, 10, 11] and res[1][1] == [4, 5, 6, 7]      # channels 0,2 / channel 1
    blobs = [O.compress(O.cparams(2), raw[i * 65536:(i + 1) * 65536], destsize=65536 + 32) for i in range(12)]
    want_digest = [int(np.frombuffer(c[:r], dtype=np.uint8).astype(np.uint64).sum()) * 1000003 + r for r, c in blobs]
    for rank, mine, full, t, digest in res:
        assert full == want
        assert t == 2.0                                                                 # max over ranks
        if rank == 0:
            assert digest == want_digest                                                # gathered chunks == single-process chunks


def test_four_gloo_ranks_more_ranks_than_channels():
    """World size 4 over 3 channels of 4 chunks: fewer channels than ranks, so the partition goes chunk-granular (SURVEY.md section 8e:
    configs[1] at 8 GPUs is this shape) -- every rank compresses its chunks, the sizes, the gather to rank 0 and the scatter back are
    exact, and the gathered chunks are the single-process ones."""
    import torch.multiprocessing as mp
    import _oracle as O
    O.lib()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 4
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    owned = [m for _, m, _, _, _ in res]
    assert sorted(i for m in owned for i in m) == list(range(12))                      # every chunk exactly once
    assert owned == [shard.partition(12, world, r, 4).tolist() for r in range(world)]
    assert owned[0] == [0, 4, 8]                                                         # 3 channels < 4 ranks: chunk-granular round-robin
    chans = [synth.tiled_channel(np.uint16, 512, 256, c=c) for c in range(3)]
    raw = np.concatenate([c.view(np.uint8).ravel() for c in chans])
    blobs = [O.compress(O.cparams(2), raw[i * 65536:(i + 1) * 65536], destsize=65536 + 32) for i in range(12)]
    want_digest = [int(np.frombuffer(c[:r], dtype=np.uint8).astype(np.uint64).sum()) * 1000003 + r for r, c in blobs]
    for rank, mine, full, t, digest in res:
        assert full == [r for r, _ in blobs]
        assert t == float(world)                                                        # max over ranks of 1 + rank
        if rank == 0:
            assert digest == want_digest
