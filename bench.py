#!/usr/bin/env python3
"""bench.py -- compress+decompress GB/s (uncompressed side) of the blosc2 chunk codec path on MI355X.

Workload (BASELINE.json configs[1]): 4 channels of 4096x4096 float16, lz4 level 9 + byte shuffle,
32 KiB blocks, 4 MiB chunks (8 chunks per channel, 128 blocks per chunk, 2 x 16 KiB streams per
block), seeded synthetic "tiled channel" data (cimg/synth.py).  One *step* =
compress all 32 chunks (one batched call) + decompress them (one batched call), pixels and chunks
resident in HBM.  value = ranks * steps * 2 * N / wall  (N = 128 MiB per rank; weak scaling: every rank
owns its own 4 channels, there is no data-path collective).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.  `roofline` is for the kernel with the largest share of device time
(HIP events recorded on the engine's stream around every launch); `kernels` lists all four.
`cpu_baseline` times the oracle (CPU restatement, oracle/) on the host cores -- it is the checker
being timed, never part of the measured GPU path.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "compressed-image_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402  (first: keeps one HIP runtime in the process)

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md); measured copy peak is ~6300

WIDTH = HEIGHT = 4096
CHANNELS = 4
DTYPE = np.float16
CHUNK = 4 * 1024 * 1024
BLOCK = 32768


def cpu_baseline(raw_channels, budget_s=10.0, threads=16):
    """Oracle compress+decompress of the same chunks on this GPU's share of the host cores (bounded sample)."""
    from concurrent.futures import ThreadPoolExecutor
    import _oracle as O
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(avail, threads)                              # 16 = one GPU's share of an 8-GPU host's cores
    O.lib()
    p = O.cparams(np.dtype(DTYPE).itemsize, clevel=9, blocksize=BLOCK)
    pieces = []
    for ch in raw_channels:                                  # all 4 channels = 32 chunks x 4 MiB
        raw = ch.view(np.uint8).ravel()
        pieces += [raw[o:o + CHUNK] for o in range(0, raw.size, CHUNK)]
    nbytes = sum(x.size for x in pieces)

    def one(piece):
        r, c = O.compress(p, piece, destsize=CHUNK + 32)
        rr, out = O.decompress(c, piece.size)
        return r, rr

    with ThreadPoolExecutor(max_workers=cores) as ex:
        list(ex.map(one, pieces))                           # warm
        t0 = time.perf_counter()
        reps = 0
        while True:
            list(ex.map(one, pieces))
            reps += 1
            if time.perf_counter() - t0 > budget_s:
                break
        dt = time.perf_counter() - t0
    gbps = reps * 2 * nbytes / dt / 1e9
    return {"value": round(gbps, 3), "unit": "GB/s", "cores": cores, "kind": "port",
            "sample": f"oracle (oracle/ CPU restatement, not c-blosc2): the same {len(pieces)} chunks x 4 MiB, "
                      f"compress+decompress per chunk, {reps} passes over a {cores}-thread pool in {dt:.1f} s "
                      f"({avail} hardware threads visible)"}


# HBM traffic per launch comes from rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, KiB), which cannot be collected
# from inside this process: the committed per-launch averages of the same workload are reported instead
# (profiles/tools/collect.sh makes them).  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts half the
# bytes of 16-B-per-lane streaming reads -- the decoder stages compressed bytes that way (x2); the encoder's strided
# byte-plane gather is calibrated against its known read volume (every raw byte exactly once: factor 1).
TIMING_PERIOD = 4
FILTER_TEXT = {"shuffle": "byte shuffle", "bitshuffle": "bitshuffle (one unsplit stream per block)", "none": "no filter"}
PMC_FILES = ("final_pmc_per_launch.json", "mid_pmc_per_launch.json")
FETCH_FACTOR = {"cimg_encode_streams": 1.0, "cimg_decode_blocks": 2.0, "cimg_decode_lean": 2.0}
# the decode entry of the engine timers covers two launches (lean kernel + general kernel behind it)
PMC_KERNELS = {"cimg_decode_blocks": ("cimg_decode_lean", "cimg_decode_blocks")}


def pmc_traffic(kernel):
    here = os.path.dirname(os.path.abspath(__file__))
    for name in PMC_FILES:
        path = os.path.join(here, "profiles", "r01", name)
        if not os.path.exists(path):
            continue
        try:
            with open(path) as f:
                table = json.load(f)
            total, seen = 0.0, False
            for k in PMC_KERNELS.get(kernel, (kernel,)):
                c = table.get(k)
                if c and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                    total += (c["FETCH_SIZE"] * FETCH_FACTOR.get(k, 1.0) + c["WRITE_SIZE"]) * 1024
                    seen = True
            if seen:
                return int(total), "profiles/r01/" + name
        except (OSError, ValueError):
            pass
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--family", default="tiled", choices=["tiled", "natural", "random", "zero"])
    ap.add_argument("--filter", default="shuffle", choices=["shuffle", "bitshuffle", "none"],
                    help="not part of the headline: the reference only uses byte shuffle")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py needs a GPU (MI355X); none visible", file=sys.stderr)
        sys.exit(1)
    # CIMG_BENCH_REHEARSAL=1: every rank on GPU 0 over gloo -- rehearses the N > 1 control flow on a one-GPU box
    # (numbers from such a run mean nothing: the ranks share one card)
    rehearsal = os.environ.get("CIMG_BENCH_REHEARSAL") is not None
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    red_dev = "cpu" if rehearsal else "cuda"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    from cimg import hip, synth

    # ---- inputs: this rank's 4 channels, resident in HBM ------------------------------------------------
    gen = getattr(synth, args.family + "_channel")
    chans = []
    for c in range(CHANNELS):
        if args.family == "zero":
            chans.append(gen(DTYPE, WIDTH, HEIGHT))
        else:
            chans.append(gen(DTYPE, WIDTH, HEIGHT, c=CHANNELS * rank + c))
    host = np.concatenate([c.view(np.uint8).ravel() for c in chans])
    N = host.size
    nchunks = N // CHUNK
    stride = CHUNK + 64
    d_raw = torch.from_numpy(host).cuda()
    d_out = torch.zeros(N, dtype=torch.uint8, device="cuda")
    d_comp = torch.zeros(nchunks * stride, dtype=torch.uint8, device="cuda")
    raw_off = np.arange(nchunks, dtype=np.int64) * CHUNK
    comp_off = np.arange(nchunks, dtype=np.int64) * stride
    nbytes = np.full(nchunks, CHUNK, np.int32)
    destsize = np.full(nchunks, CHUNK + 32, np.int32)          # schunk.h:73: nominal chunk + BLOSC2_MAX_OVERHEAD
    blocksize = np.full(nchunks, BLOCK, np.int32)
    torch.cuda.synchronize()

    eng = hip.Engine(local_rank)
    filt = {"shuffle": hip.SHUFFLE, "bitshuffle": hip.BITSHUFFLE, "none": 0}[args.filter]
    p = hip.cparams(np.dtype(DTYPE).itemsize, clevel=9, blocksize=BLOCK, compcode=hip.LZ4, filters=(0, 0, 0, 0, 0, filt))

    def step():
        cb = eng.compress_device(p, d_raw.data_ptr(), raw_off, nbytes, d_comp.data_ptr(), comp_off, destsize)
        eng.decompress_device(d_comp.data_ptr(), comp_off, nbytes, blocksize, d_out.data_ptr(), raw_off)
        return cb

    cbytes = step()
    if not torch.equal(d_out, d_raw):
        print("bench.py: decompressed pixels differ from the input -- refusing to report a number", file=sys.stderr)
        sys.exit(3)
    for _ in range(args.warmup):
        step()

    # kernel durations: HIP events on the engine's stream around every kernel of every TIMING_PERIOD-th batch call of
    # the timed region (an event record costs ~5 us of stream time; all eight per step were 37 us of an 800 us step)
    eng.enable_timing(0 if os.environ.get("CIMG_BENCH_NO_EVENTS") else TIMING_PERIOD)
    eng.reset_timing()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        csum = torch.tensor([float(cbytes.sum())], dtype=torch.float64, device=red_dev)
        dist.all_reduce(csum, op=dist.ReduceOp.SUM)
        total_c = float(csum.item())
    else:
        total_c = float(cbytes.sum())

    ktimes = [eng.kernel_time(k) for k in range(4)]
    eng.enable_timing(False)
    if not torch.equal(d_out, d_raw):
        print("bench.py: pixels differ after the timed region", file=sys.stderr)
        sys.exit(3)

    if rank == 0:
        C = float(cbytes.sum())
        algo = {hip.K_ENCODE: N + C, hip.K_LAYOUT: 0.0, hip.K_EMIT: 0.0, hip.K_DECODE: C + N}
        kernels = {}
        for k, (ms, n) in enumerate(ktimes):
            avg = ms / n if n else 0.0
            kernels[hip.KERNELS[k]] = {
                "launches": n, "avg_us": round(avg * 1e3, 2),
                "algorithmic_GBps": round(algo[k] / (avg * 1e-3) / 1e9, 1) if avg > 0 and algo[k] else None}
        dom = max(range(4), key=lambda k: ktimes[k][0])
        dom_avg_s = ktimes[dom][0] / max(ktimes[dom][1], 1) * 1e-3
        achieved = algo[dom] / dom_avg_s / 1e9 if dom_avg_s > 0 else 0.0
        dec_avg_s = ktimes[hip.K_DECODE][0] / max(ktimes[hip.K_DECODE][1], 1) * 1e-3
        traffic, traffic_src = pmc_traffic(hip.KERNELS[dom]) if (args.family == "tiled" and args.filter == "shuffle") else (None, None)
        dec_traffic, _ = pmc_traffic(hip.KERNELS[hip.K_DECODE]) if (args.family == "tiled" and args.filter == "shuffle") else (None, None)
        out = {
            "metric": "compress+decompress GB/s (uncompressed side)",
            "value": round(world * args.steps * 2 * N / elapsed / 1e9, 3),
            "unit": "GB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{CHANNELS}x{WIDTH}x{HEIGHT} float16 per GPU, lz4 clevel 9 + {FILTER_TEXT[args.filter]}, "
                                   f"32 KiB blocks, 4 MiB chunks ({nchunks} chunks, {N // BLOCK} blocks, {2 * N // BLOCK} streams), "
                                   f"device-resident, family={args.family}",
                       "uncompressed_bytes_per_gpu": N, "compressed_bytes_per_gpu": int(C),
                       "compression_ratio": round(world * N / total_c, 4) if total_c else None,
                       "roundtrip_GBps": round(world * args.steps * N / elapsed / 1e9, 3),
                       "parallelism": f"chunks sharded by rank x{world}, no data-path collective"},
            "roofline": {"kernel": hip.KERNELS[dom], "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": int(algo[dom]), "avg_launch_us": round(dom_avg_s * 1e6, 2)},
            "roofline_decode": {"kernel": hip.KERNELS[hip.K_DECODE], "bound": "hbm",
                                "achieved": round((C + N) / dec_avg_s / 1e9, 1) if dec_avg_s > 0 else None,
                                "output_side": round(N / dec_avg_s / 1e9, 1) if dec_avg_s > 0 else None,
                                "peak": HBM_PEAK_GBPS, "unit": "GB/s", "traffic": dec_traffic,
                                "frac": round((C + N) / dec_avg_s / 1e9 / HBM_PEAK_GBPS, 4) if dec_avg_s > 0 else None},
            "kernels": kernels,
            "kernel_timing": f"HIP events around every kernel of every {TIMING_PERIOD}th batch call inside the timed region",
        }
        if not args.no_cpu_baseline and world == 1 and args.filter == "shuffle":
            out["cpu_baseline"] = cpu_baseline(chans)
            # the same port with one thread per chunk of the workload (all the chunk-level parallelism there is)
            out["cpu_baseline_one_thread_per_chunk"] = cpu_baseline(chans, budget_s=8.0, threads=nchunks)
        print(json.dumps(out))
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
