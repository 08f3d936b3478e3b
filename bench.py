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
import ctypes as C
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
EXCHANGE_TIMEOUT_S = int(os.environ.get("CIMG_BENCH_EXCHANGE_TIMEOUT", "120"))        # main(), N > 1: how long the gather of the finished chunks may take before the line is printed without it


def _oracle_batch_lib():
    import _oracle as O
    L = O.lib()
    L.orc_bench_compress.argtypes = [C.POINTER(O.CParams), C.c_void_p, C.c_int, C.c_int32, C.c_void_p, C.c_int64, C.c_int32,
                                     C.c_void_p, C.c_int, C.c_int]
    L.orc_bench_compress.restype = C.c_int64
    L.orc_bench_decompress.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_int, C.c_int]
    L.orc_bench_decompress.restype = C.c_int64
    return L, O


def visible_cores():
    return len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)


def cpu_baseline(host, budget_s=10.0, policy="share", typesize=2, compcode=None, chunk=None, threads=None, filt=None, destsize=None):
    """The oracle (C, -O3, OpenMP inside the library: no Python in the timed region) over the same chunks.

    policy "share":     an OpenMP loop over chunks on THIS GPU's share of the host cores (16 of an 8-GPU host's 128 cores / 256
                        threads: what one rank of an 8-rank job can count on).
    policy "all_cores": every hardware thread the process can see -- chunks over min(threads, nchunks) teams, the rest of the
                        threads over the blocks inside each team's chunk (nested OpenMP; two-phase compress, block-parallel
                        decompress).  No cap.
    policy "reference": the reference's own call structure -- chunks one after the other (schunk.h:85-94), compression
                        with hardware_concurrency() / 2 threads over the blocks of a chunk (channel.h:127), decompression
                        on ONE thread (wrapper.h:406)."""
    L, O = _oracle_batch_lib()
    chunk = chunk or CHUNK
    avail = visible_cores()
    nchunks = host.size // chunk
    if policy == "share":
        cores = min(avail, 16)
        enc = (min(cores, nchunks), max(1, cores // min(cores, nchunks)))
        dec = enc
    elif policy == "all_cores":
        cores = threads or avail
        teams = min(cores, nchunks)
        enc = dec = (teams, max(1, cores // teams))
    elif policy == "one":
        cores = 1
        enc = dec = (1, 1)
    elif policy == "blocks16":                                # one chunk at a time, this GPU's share of the host over its blocks
        cores = min(avail, 16)
        enc, dec = (1, cores), (1, cores)
    else:
        cores = max(1, avail // 2)
        enc, dec = (1, cores), (1, 1)
    p = O.cparams(typesize, clevel=9, blocksize=BLOCK, compcode=O.LZ4 if compcode is None else compcode,
                  **({} if filt is None else {"filters": (0, 0, 0, 0, 0, filt)}))
    destsize = destsize or chunk + 32
    stride = max(chunk, destsize) + 64
    comp = np.zeros(nchunks * stride, np.uint8)
    out = np.zeros(host.size, np.uint8)
    cb = np.zeros(nchunks, np.int32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)

    def one_pass():
        t0 = time.perf_counter()
        r = L.orc_bench_compress(C.byref(p), vp(host), nchunks, chunk, vp(comp), stride, destsize, vp(cb), enc[0], enc[1])
        t1 = time.perf_counter()
        d = L.orc_bench_decompress(vp(comp), nchunks, stride, vp(cb), vp(out), chunk, dec[0], dec[1])
        t2 = time.perf_counter()
        assert r > 0 and d == host.size
        return t1 - t0, t2 - t1

    one_pass()                                               # warm
    assert out.tobytes() == host.tobytes()
    te = td = 0.0
    reps = 0
    t_start = time.perf_counter()
    while time.perf_counter() - t_start < budget_s:
        a, b = one_pass()
        te += a; td += b; reps += 1
    n = host.size
    return {"value": round(reps * 2 * n / (te + td) / 1e9, 3), "unit": "GB/s", "cores": cores, "kind": "port",
            "compress_GBps": round(reps * n / te / 1e9, 3), "decompress_GBps": round(reps * n / td / 1e9, 3),
            "policy": policy,
            "sample": f"oracle (oracle/ CPU restatement at -O3, not c-blosc2): the same {nchunks} chunk(s) x {chunk >> 20} MiB, {reps} passes in "
                      f"{te + td:.1f} s; compress: {enc[0]} thread(s) over chunks x {enc[1]} over blocks, decompress: "
                      f"{dec[0]} x {dec[1]} ({avail} hardware threads visible)"}


def cblosc2_baseline(host, budget_s=8.0):
    """The genuine CPU codec, if a c-blosc2 shared library is installed on this box: called exactly as the reference does --
    blosc2_create_cctx with the reference's cparams, one blosc2_compress_ctx per 4 MiB chunk, chunks serial, nthreads = hw / 2
    for compression (channel.h:127) and 1 for decompression (wrapper.h:406).  Returns None when the library is absent."""
    import _cblosc2 as R
    B, name = R.open_blosc2()
    if B is None:
        return None
    try:
        enc_threads = max(1, visible_cores() // 2)
        cctx = R.cctx(B, 2, min(enc_threads, 32767))
        dctx = R.dctx(B, 1)
        nchunks = host.size // CHUNK
        comp = np.zeros(nchunks * (CHUNK + 64), np.uint8)
        out = np.zeros(host.size, np.uint8)
        cb = [0] * nchunks
        te = td = 0.0
        reps = 0
        t_start = time.perf_counter()
        while time.perf_counter() - t_start < budget_s:
            t0 = time.perf_counter()
            for i in range(nchunks):
                cb[i] = B.blosc2_compress_ctx(cctx, host.ctypes.data + i * CHUNK, CHUNK, comp.ctypes.data + i * (CHUNK + 64), CHUNK + 32)
            t1 = time.perf_counter()
            for i in range(nchunks):
                B.blosc2_decompress_ctx(dctx, comp.ctypes.data + i * (CHUNK + 64), 2**31 - 1, out.ctypes.data + i * CHUNK, CHUNK)
            t2 = time.perf_counter()
            te += t1 - t0; td += t2 - t1; reps += 1
        B.blosc2_free_ctx(cctx)
        B.blosc2_free_ctx(dctx)
        if out.tobytes() != host.tobytes():
            return None
        n = host.size
        return {"value": round(reps * 2 * n / (te + td) / 1e9, 3), "unit": "GB/s", "cores": enc_threads, "kind": "reference",
                "compress_GBps": round(reps * n / te / 1e9, 3), "decompress_GBps": round(reps * n / td / 1e9, 3),
                "library": name, "sample": f"{nchunks} chunks x 4 MiB, {reps} passes, blosc2_compress_ctx / blosc2_decompress_ctx per chunk as "
                                           f"blosc2/wrapper.h:139,246 call them (compress nthreads {enc_threads}, decompress 1)"}
    except (OSError, AttributeError):
        return None


# HBM traffic per launch comes from rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, KiB; separate runs, they cannot be
# collected from inside this process).  profiles/tools/collect.sh stores the per-launch averages TOGETHER WITH a hash of
# the kernel sources they were collected on; the figure is reported only when that hash equals the hash of the sources
# this run was built from (else null), or when --pmc-json names a file explicitly.
# gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts HALF the bytes of 16-B-per-lane streaming reads, which
# is how every kernel here reads its bulk input, so FETCH_SIZE is doubled; WRITE_SIZE is exact for 16-B stores.
TIMING_PERIOD = 4
FILTER_TEXT = {"shuffle": "byte shuffle", "bitshuffle": "bitshuffle (one unsplit stream per block)", "none": "no filter"}
FETCH_FACTOR = 2.0
# the decode entry of the engine timers covers two launches (lean kernel + general kernel behind it)
PMC_KERNELS = {"cimg_decode_blocks": ("cimg_decode_lean", "cimg_decode_blocks")}     # (keyed by the engine's timing ids' names, cimg/hip.py KERNELS)


def kernel_source_hash():
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(ROOT, "compressed-image_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".h", ".hip", ".cpp")):
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode()); h.update(f.read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel, explicit=None):
    # (the newest round's counters first: the file names the kernel sources it was collected on, and only a match is reported)
    rounds = sorted((r for r in os.listdir(os.path.join(ROOT, "profiles")) if r[:1] == "r" and r[1:].isdigit()), reverse=True) if os.path.isdir(os.path.join(ROOT, "profiles")) else []
    path = explicit or next((q for q in (os.path.join(ROOT, "profiles", r, "pmc_per_launch.json") for r in rounds) if os.path.exists(q)), "")
    if not os.path.exists(path):
        return None, None
    try:
        with open(path) as f:
            table = json.load(f)
        if not explicit and table.get("_source_hash") != kernel_source_hash():
            return None, f"{os.path.relpath(path, ROOT)} is from other kernel sources ({table.get('_source_hash')}): not reported"
        total, seen = 0.0, False
        for k in PMC_KERNELS.get(kernel, (kernel,)):
            c = table.get(k)
            if c and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                total += (c["FETCH_SIZE"] * FETCH_FACTOR + c["WRITE_SIZE"]) * 1024
                seen = True
        if seen:
            return int(total), f"{os.path.relpath(path, ROOT)} (sources {table.get('_source_hash')}, FETCH_SIZE x 2 per the gfx950 rule)"
    except (OSError, ValueError):
        pass
    return None, None


def measure_copy_peak(nbytes=256 << 20, reps=12):
    """The HBM rate a plain device-to-device copy reaches on THIS card, in this process (fractions are
    quoted against the 8 TB/s specification AND against a measured copy peak): torch's vectorised copy kernel over buffers far
    larger than the caches, HIP events on torch's stream.  bytes moved = read + written."""
    a = torch.empty(nbytes, dtype=torch.uint8, device="cuda").fill_(7)
    b = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    for _ in range(3):
        b.copy_(a)
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            b.copy_(a)
        e1.record()
        e1.synchronize()
        gbps = 2.0 * nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9
        best = gbps if best is None or gbps > best else best
    del a, b
    return round(best, 1)


def _median(xs):
    xs = sorted(xs)
    return 0.5 * (xs[(len(xs) - 1) // 2] + xs[len(xs) // 2]) if xs else 0.0


def run_config1(args, rank, world, local_rank, dist, red_dev):
    """BASELINE configs[0]: ONE 1024 x 1024 uint8 channel, blosclz + byte shuffle, default chunk size -- the reference's own
    CPU-runnable plumbing case (test/src/test_channel.cpp:74-87 shape).  The default chunk size is 4 MiB, so the channel is ONE chunk
    of 1 MiB (32 blocks): a step = compress it + decompress it, device-resident, through the single-chunk form of the batched calls
    (what one blosc2_compress_ctx / blosc2_decompress_ctx of the reference becomes).  32 blocks cannot fill 256 CUs: this line is
    the latency floor of the path, not a throughput claim."""
    import _oracle as O
    from cimg import hip, synth
    W = H = 1024
    codec = args.codec or "blosclz"
    gen = getattr(synth, args.family + "_channel")
    chan = gen(np.uint8, W, H) if args.family == "zero" else gen(np.uint8, W, H, c=rank)
    host = chan.view(np.uint8).ravel()
    N = host.size
    d_raw = torch.from_numpy(host).cuda()
    d_out = torch.zeros(N, dtype=torch.uint8, device="cuda")
    d_comp = torch.zeros(CHUNK + 64, dtype=torch.uint8, device="cuda")
    eng = hip.Engine(local_rank)
    filt = {"shuffle": hip.SHUFFLE, "bitshuffle": hip.BITSHUFFLE, "none": 0}[args.filter or "shuffle"]
    compcode = hip.BLOSCLZ if codec == "blosclz" else hip.LZ4
    p = hip.cparams(1, clevel=9, blocksize=BLOCK, compcode=compcode, filters=(0, 0, 0, 0, 0, filt))
    step = eng.roundtrip_calls(p, d_raw.data_ptr(), [0], [N], d_comp.data_ptr(), [0], [CHUNK + 32], [BLOCK], d_out.data_ptr())
    cb = step().copy()
    if not torch.equal(d_out, d_raw):
        print("bench.py --config 1: decompressed pixels differ from the input", file=sys.stderr)
        sys.exit(3)
    po = O.cparams(1, clevel=9, blocksize=BLOCK, compcode=O.BLOSCLZ if codec == "blosclz" else O.LZ4, filters=(0, 0, 0, 0, 0, filt))
    r, want = O.compress(po, host, destsize=CHUNK + 32)
    same_bytes = int(cb[0]) == r and d_comp[:r].cpu().numpy().tobytes() == want
    for _ in range(args.warmup):
        step()
    eng.enable_timing(1)
    eng.reset_timing()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    per_step = []
    t0 = time.perf_counter()
    for k in range(args.steps):
        if k == args.steps - 1:
            d_out.zero_()                                       # the last step's pixels are checked: they must be written by it
            torch.cuda.synchronize()
        ts = time.perf_counter()
        step()
        per_step.append(time.perf_counter() - ts)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if not torch.equal(d_out, d_raw):
        print("bench.py --config 1: pixels differ after the timed region", file=sys.stderr)
        sys.exit(3)
    enc = eng.kernel_samples(hip.K_ENCODE) * 1e3
    dec = eng.kernel_samples(hip.K_DECODE) * 1e3
    eng.enable_timing(False)
    Cb = int(cb[0])
    if rank == 0:
        e_us, d_us = _median(enc.tolist()), _median(dec.tolist())
        out = {"metric": "compress+decompress GB/s (uncompressed side)", "value": round(world * args.steps * 2 * N / elapsed / 1e9, 3), "unit": "GB/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
               "ms_per_step_median": round(_median(per_step) * 1e3, 4),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
               "config": {"workload": f"BASELINE configs[0]: 1x{W}x{H} uint8, {codec} clevel 9 + {FILTER_TEXT[args.filter or 'shuffle']}, one chunk of 1 MiB in a "
                                      f"4 MiB + 32 buffer (32 blocks of 32 KiB), device-resident, family={args.family}; single-chunk calls",
                          "element_dtype": "uint8", "uncompressed_bytes_per_gpu": N, "compressed_bytes_per_gpu": Cb,
                          "compression_ratio": round(N / Cb, 4) if Cb else None, "bytes_equal_oracle": bool(same_bytes),
                          "us_per_chunk_roundtrip": round(elapsed / args.steps * 1e6, 1)},
               "roofline": {"kernel": "cimg_encode_streams_blosclz" if codec == "blosclz" else "cimg_encode_streams", "bound": "hbm",
                            "achieved": round((N + Cb) / (e_us * 1e-6) / 1e9, 1) if e_us else None, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                            "frac": round((N + Cb) / (e_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 5) if e_us else None, "traffic": None,
                            "algorithmic_bytes_per_launch": N + Cb, "median_launch_us": round(e_us, 2),
                            "note": "32 work items on a device that holds 1280 chains: the launch is one chain's latency"},
               "roofline_decode": {"kernel": "cimg_decode_lean + cimg_decode_blocks", "bound": "hbm", "median_launch_us": round(d_us, 2),
                                   "achieved": round((N + Cb) / (d_us * 1e-6) / 1e9, 1) if d_us else None, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                   "frac": round((N + Cb) / (d_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 5) if d_us else None}}
        if args.family == "tiled" and world == 1:
            # configs[0] as named is incompressible (four noise bits a byte: the chunk comes out memcpyed, ratio 1.000): a second leg on
            # the natural family, where the codec actually codes -- same calls, same geometry
            nat = synth.natural_channel(np.uint8, W, H).view(np.uint8).ravel()
            d_raw.copy_(torch.from_numpy(nat).cuda())
            ncb = step().copy()
            okn = torch.equal(d_out, d_raw)
            rn, wantn = O.compress(po, nat, destsize=CHUNK + 32)
            same_n = int(ncb[0]) == rn and d_comp[:rn].cpu().numpy().tobytes() == wantn
            for _ in range(args.warmup):
                step()
            eng.enable_timing(1); eng.reset_timing()
            torch.cuda.synchronize()
            tn = time.perf_counter()
            nsteps = max(20, args.steps // 4)
            for _ in range(nsteps):
                step()
            torch.cuda.synchronize()
            tn = time.perf_counter() - tn
            ne, nd = eng.kernel_samples(hip.K_ENCODE) * 1e3, eng.kernel_samples(hip.K_DECODE) * 1e3
            eng.enable_timing(False)
            out["compressible"] = {"what": "the same single-chunk round trip on the natural family (the tiled uint8 chunk is memcpyed)", "family": "natural",
                                   "value": round(nsteps * 2 * N / tn / 1e9, 3), "unit": "GB/s", "us_per_chunk_roundtrip": round(tn / nsteps * 1e6, 1),
                                   "compression_ratio": round(N / int(ncb[0]), 4), "encode_us": round(_median(ne.tolist()), 1), "decode_us": round(_median(nd.tolist()), 1),
                                   "bytes_equal_oracle": bool(same_n), "pixels_verified": bool(okn)}
            if not args.no_cpu_baseline:
                out["compressible"]["cpu_baseline"] = cpu_baseline(nat, budget_s=3.0, policy="blocks16", typesize=1, compcode=po.compcode, chunk=N, filt=filt, destsize=CHUNK + 32)
            d_raw.copy_(torch.from_numpy(host).cuda())
        if not args.no_cpu_baseline and world == 1:
            # the reference's own call: one thread team over the blocks of the one chunk (channel.h:127: hw / 2 threads) for compression,
            # ONE thread for decompression (wrapper.h:406)
            # (this GPU's share of the host -- 16 threads over the blocks of the one chunk --, then the reference's hw / 2 threads, which
            # on a box that shows 256 hardware threads to a container scheduled on 16 cores mostly thrash, then one thread)
            out["cpu_baseline"] = cpu_baseline(host, budget_s=4.0, policy="blocks16", typesize=1, compcode=po.compcode, chunk=N, filt=filt, destsize=CHUNK + 32)
            out["cpu_baseline_reference_policy"] = cpu_baseline(host, budget_s=3.0, policy="reference", typesize=1, compcode=po.compcode, chunk=N, filt=filt, destsize=CHUNK + 32)
            out["cpu_baseline_one_thread"] = cpu_baseline(host, budget_s=3.0, policy="one", typesize=1, compcode=po.compcode, chunk=N, filt=filt, destsize=CHUNK + 32)
        print(json.dumps(out))
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def run_config3(args, rank, world, local_rank, dist, red_dev):
    """BASELINE configs[2]: 8192 x 8192 uint16, 3 channels, blosclz + bitshuffle, the random-access loop of the reference's
    examples/lazy_channels/main.cpp:35-52 (channel.h:502-538): visit the chunks in a seeded random permutation (rng 99, SURVEY.md
    section 8d), get_chunk -> every pixel + 1 -> set_chunk.  Device-resident SINGLE-CHUNK calls: one cimg_decompress_batch_device of
    one chunk into a work buffer, the + 1 on the device (on the engine's stream), one cimg_compress_batch_device of one chunk back
    into its slot.  A step = one pass over the permutation of all 96 chunks (4 MiB each); GB/s counts the uncompressed bytes decoded
    + encoded.  128 blocks per call on a 256-CU device: this is the latency-bound face of the path, reported as such."""
    import _oracle as O
    from cimg import hip, synth
    W = H = 8192
    NCH = 3
    codec = args.codec or "blosclz"
    filt_name = args.filter or "bitshuffle"
    gen = getattr(synth, args.family + "_channel")
    chans = [gen(np.uint16, W, H) if args.family == "zero" else gen(np.uint16, W, H, c=NCH * rank + c) for c in range(NCH)]
    host = np.concatenate([c.view(np.uint8).ravel() for c in chans])
    N = host.size
    nchunks = N // CHUNK
    stride = CHUNK + 64
    d_raw = torch.from_numpy(host).cuda()
    d_comp = torch.zeros(nchunks * stride, dtype=torch.uint8, device="cuda")
    d_work = torch.zeros(CHUNK, dtype=torch.uint8, device="cuda")
    raw_off = np.arange(nchunks, dtype=np.int64) * CHUNK
    comp_off = np.arange(nchunks, dtype=np.int64) * stride
    eng = hip.Engine(local_rank)
    compcode = hip.BLOSCLZ if codec == "blosclz" else hip.LZ4
    perm = np.random.default_rng(99).permutation(nchunks)
    ext = torch.cuda.ExternalStream(eng.stream_handle(), device=torch.device("cuda", local_rank))
    work16 = d_work.view(torch.int16)                               # (+ 1 wraps the same way for int16 and uint16 bit patterns)

    def leg(filt_name, steps, warmup, synced):
        """the loop with one filter: the channels compressed as the constructor leaves them, `warmup` + `steps` passes over the permutation,
        then everything decoded in one batch and compared with pixels + passes"""
        filt = {"shuffle": hip.SHUFFLE, "bitshuffle": hip.BITSHUFFLE, "none": 0}[filt_name]
        p = hip.cparams(2, clevel=9, blocksize=BLOCK, compcode=compcode, filters=(0, 0, 0, 0, 0, filt))
        cb0 = eng.compress_device(p, d_raw.data_ptr(), raw_off, [CHUNK] * nchunks, d_comp.data_ptr(), comp_off, [CHUNK + 32] * nchunks)
        get_chunk, set_chunk = eng.single_chunk_calls(p, d_work.data_ptr(), d_comp.data_ptr(), comp_off, CHUNK, CHUNK + 32, BLOCK)
        sizes = np.asarray(cb0, np.int64).copy()

        def one_pass():
            with torch.cuda.stream(ext):
                for i in perm:
                    get_chunk(int(i))
                    work16.add_(1)                                  # on the engine's stream: the compress batch behind it sees the result
                    sizes[i] = set_chunk(int(i))

        passes_done = 0
        for _ in range(max(warmup, 1)):
            one_pass(); passes_done += 1
        eng.enable_timing(1)
        eng.reset_timing()
        if synced and dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        per_step = []
        t0 = time.perf_counter()
        for _ in range(steps):
            ts = time.perf_counter()
            one_pass(); passes_done += 1
            per_step.append(time.perf_counter() - ts)
        torch.cuda.synchronize()
        if synced and dist is not None:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if synced and dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        enc = eng.kernel_samples(hip.K_ENCODE) * 1e3
        dec = eng.kernel_samples(hip.K_DECODE) * 1e3
        eng.enable_timing(False)
        # every pixel went up by one per pass: decode everything in one batch and compare
        d_out = torch.zeros(N, dtype=torch.uint8, device="cuda")
        eng.decompress_device(d_comp.data_ptr(), comp_off, [CHUNK] * nchunks, [BLOCK] * nchunks, d_out.data_ptr(), raw_off)
        want = (host.view(np.uint16) + np.uint16(passes_done)).view(np.uint8)
        if d_out.cpu().numpy().tobytes() != want.tobytes():
            print(f"bench.py --config 3 ({filt_name}): the channels do not hold pixels + passes after the loop", file=sys.stderr)
            sys.exit(3)
        del d_out
        return {"elapsed": elapsed, "per_step": per_step, "enc": enc, "dec": dec, "sizes": sizes, "filt": filt}

    main_leg = leg(filt_name, args.steps, args.warmup, True)
    elapsed, per_step, enc, dec, sizes, filt = (main_leg[k] for k in ("elapsed", "per_step", "enc", "dec", "sizes", "filt"))
    visits = args.steps * nchunks
    Cb = float(sizes.sum())
    if rank == 0:
        e_us, d_us = _median(enc.tolist()), _median(dec.tolist())
        cmean = Cb / nchunks
        out = {"metric": "compress+decompress GB/s (uncompressed side), random-access get_chunk -> +1 -> set_chunk loop",
               "value": round(world * visits * 2 * CHUNK / elapsed / 1e9, 3), "unit": "GB/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
               "ms_per_step_median": round(_median(per_step) * 1e3, 4),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
               "config": {"workload": f"BASELINE configs[2]: {NCH}x{W}x{H} uint16 per GPU, {codec} clevel 9 + {FILTER_TEXT[filt_name]}, 32 KiB blocks, 4 MiB chunks "
                                      f"({nchunks} chunks), seeded random permutation (rng 99), get_chunk -> +1 -> set_chunk per chunk, "
                                      f"single-chunk device-resident calls, family={args.family}; a step = one pass over all {nchunks} chunks",
                          "element_dtype": "uint16", "uncompressed_bytes_per_gpu": N, "compressed_bytes_per_gpu": int(Cb),
                          "compression_ratio": round(N / Cb, 4) if Cb else None,
                          "us_per_chunk_visit": round(elapsed / visits * 1e6, 1), "chunk_visits": visits},
               "roofline": {"kernel": "cimg_encode_streams_blosclz" if codec == "blosclz" else "cimg_encode_streams", "bound": "hbm",
                            "achieved": round((CHUNK + cmean) / (e_us * 1e-6) / 1e9, 1) if e_us else None, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                            "frac": round((CHUNK + cmean) / (e_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 5) if e_us else None, "traffic": None,
                            "algorithmic_bytes_per_launch": int(CHUNK + cmean), "median_launch_us": round(e_us, 2),
                            "note": "one chunk = 128 blocks per launch on a device that holds 1280 encode chains / 2048 decode waves: the launch is "
                                    "a single chain's latency (kernels incl. leftover-free assembly; medians of every launch in the timed region)"},
               "roofline_decode": {"kernel": "cimg_decode_lean + cimg_decode_blocks", "bound": "hbm", "median_launch_us": round(d_us, 2),
                                   "achieved": round((CHUNK + cmean) / (d_us * 1e-6) / 1e9, 1) if d_us else None, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                   "frac": round((CHUNK + cmean) / (d_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 5) if d_us else None}}
        if filt_name == "bitshuffle" and codec == "blosclz" and args.family == "tiled" and world == 1:
            # configs[2] as named comes out memcpyed (BloscLZ 2.3.0 gives up on the noise bit rows: ratio 1.000): one pass of the same
            # loop with BYTE shuffle, where the codec actually codes
            cl = leg("shuffle", 1, 1, False)
            ce, cd = _median(cl["enc"].tolist()), _median(cl["dec"].tolist())
            out["compressible"] = {"what": "one pass of the same loop with byte shuffle (the reference's filter) instead of bitshuffle", "filter": "byte shuffle",
                                   "value": round(nchunks * 2 * CHUNK / cl["elapsed"] / 1e9, 3), "unit": "GB/s",
                                   "us_per_chunk_visit": round(cl["elapsed"] / nchunks * 1e6, 1), "compression_ratio": round(N / float(cl["sizes"].sum()), 4),
                                   "encode_us": round(ce, 1), "decode_us": round(cd, 1), "pixels_verified": True}
            if not args.no_cpu_baseline:
                out["compressible"]["cpu_baseline"] = cpu_random_access(host, perm, compcode=O.BLOSCLZ, filt=hip.SHUFFLE, budget_s=5.0,
                                                                        cthreads=min(visible_cores(), 16), dthreads=min(visible_cores(), 16))
        if not args.no_cpu_baseline and world == 1:
            # the same loop on the host: the reference's call structure (one chunk at a time; compression with hw / 2 threads over the
            # chunk's blocks, channel.h:127; decompression on one thread, wrapper.h:406; the + 1 in numpy) on a bounded sample of chunks
            cc = O.BLOSCLZ if codec == "blosclz" else O.LZ4
            out["cpu_baseline"] = cpu_random_access(host, perm, compcode=cc, filt=filt, budget_s=8.0, cthreads=min(visible_cores(), 16), dthreads=min(visible_cores(), 16))
            out["cpu_baseline_reference_policy"] = cpu_random_access(host, perm, compcode=cc, filt=filt, budget_s=6.0, cthreads=max(1, visible_cores() // 2), dthreads=1)
        print(json.dumps(out))
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def cpu_random_access(host, perm, compcode, filt, budget_s=12.0, cthreads=16, dthreads=16):
    """configs[2]'s loop on the host cores through the oracle (kind "port"): per visited chunk one decompress, + 1, one compress, each
    with its threads over the blocks of the ONE chunk (the reference's call structure: hw / 2 threads for compression, channel.h:127,
    ONE for decompression, wrapper.h:406; or this GPU's 16-thread share of the host for both).  Bounded: as many chunks of the
    permutation as fit the budget."""
    L, O = _oracle_batch_lib()
    avail = visible_cores()
    p = O.cparams(2, clevel=9, blocksize=BLOCK, compcode=compcode, filters=(0, 0, 0, 0, 0, filt))
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    comp = np.zeros(CHUNK + 64, np.uint8)
    cb = np.zeros(1, np.int32)
    work = np.zeros(CHUNK, np.uint8)
    tdec = tadd = tenc = 0.0
    visited = 0
    t_start = time.perf_counter()
    for i in perm:
        src = host[int(i) * CHUNK:(int(i) + 1) * CHUNK]
        r = L.orc_bench_compress(C.byref(p), vp(src), 1, CHUNK, vp(comp), CHUNK + 64, CHUNK + 32, vp(cb), 1, cthreads)     # (the chunk as it sits in the channel: not timed)
        assert r > 0
        t0 = time.perf_counter()
        d = L.orc_bench_decompress(vp(comp), 1, CHUNK + 64, vp(cb), vp(work), CHUNK, 1, dthreads)
        t1 = time.perf_counter()
        w16 = work.view(np.uint16); w16 += np.uint16(1)
        t2 = time.perf_counter()
        r = L.orc_bench_compress(C.byref(p), vp(work), 1, CHUNK, vp(comp), CHUNK + 64, CHUNK + 32, vp(cb), 1, cthreads)
        t3 = time.perf_counter()
        assert d == CHUNK and r > 0
        tdec += t1 - t0; tadd += t2 - t1; tenc += t3 - t2; visited += 1
        if time.perf_counter() - t_start > budget_s:
            break
    total = tdec + tadd + tenc
    return {"value": round(visited * 2 * CHUNK / total / 1e9, 3), "unit": "GB/s", "cores": cthreads, "kind": "port",
            "us_per_chunk_visit": round(total / visited * 1e6, 1), "decompress_us": round(tdec / visited * 1e6, 1), "plus_one_us": round(tadd / visited * 1e6, 1),
            "compress_us": round(tenc / visited * 1e6, 1),
            "sample": f"oracle (CPU restatement, not c-blosc2): {visited} chunks of the same permutation, per chunk decompress on {dthreads} thread(s), "
                      f"+ 1 in numpy, compress on {cthreads} threads over the chunk's blocks; {avail} hardware threads visible"}


def run_config5(args, rank, world, local_rank, dist, red_dev):
    """BASELINE configs[4]: 16384 x 16384 float32, 8 channels, zstd + byte shuffle over 8 GPUs -- ONE channel (1 GiB, 256 chunks
    of 4 MiB, 32768 blocks of 32 KiB) per rank, with the compression ratio beside the rate (docs/concepts/compression.rst:46).

    A step = compress the channel with the engine's zstd encoder (csrc/zstd_encode.h: FORMAT-VALID frames -- raw literals +
    predefined FSE tables over the wave's LZ4 matches -- whose bytes differ from libzstd's by construction; the blosc2 level, 9 as
    in the reference's defaults, decides only the split rule: one frame per 32 KiB block) + decompress it again (cimg_decode_zstd),
    pixels and chunks resident in HBM.  Beside it, outside the timed region: chunks as the REFERENCE writes them (the box's libzstd
    at clevel 9 = ZSTD_maxCLevel() under the checker's chunk layer, oracle/zstd_dl.c) decoded by the same kernel, their ratio, and
    libzstd's own rates on the host."""
    import _oracle as O
    from cimg import hip, synth
    L, _ = _oracle_batch_lib()
    W = H = 16384
    dt = np.float32
    clevel = args.zstd_clevel
    chan = synth.tiled_channel(dt, W, H, c=rank) if args.family == "tiled" else getattr(synth, args.family + "_channel")(dt, W, H)
    host = chan.view(np.uint8).ravel()
    N = host.size
    nchunks = N // CHUNK
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    raw_off = np.arange(nchunks, dtype=np.int64) * CHUNK
    nbytes = np.full(nchunks, CHUNK, np.int32)
    blocksize = np.full(nchunks, BLOCK, np.int32)
    gstride = CHUNK + 64
    goff = np.arange(nchunks, dtype=np.int64) * gstride
    dest = np.full(nchunks, CHUNK + 32, np.int32)
    d_raw = torch.from_numpy(host).cuda()
    d_out = torch.zeros(N, dtype=torch.uint8, device="cuda")
    d_gcomp = torch.zeros(nchunks * gstride, dtype=torch.uint8, device="cuda")
    eng = hip.Engine(local_rank)
    gp = hip.cparams(4, clevel=clevel, blocksize=BLOCK, compcode=hip.ZSTD)

    def step():
        cb = eng.compress_device(gp, d_raw.data_ptr(), raw_off, nbytes, d_gcomp.data_ptr(), goff, dest)
        eng.decompress_device(d_gcomp.data_ptr(), goff, nbytes, blocksize, d_out.data_ptr(), raw_off, comp_size=cb)
        return cb

    gcb = step()
    if not torch.equal(d_out, d_raw):
        print("bench.py --config 5: decompressed pixels differ from the input -- refusing to report a number", file=sys.stderr)
        sys.exit(3)
    have_libzstd = O.zstd_available()
    if have_libzstd:                                               # the real library reads the engine's chunks too (a sample)
        sample = d_gcomp[:3 * gstride].cpu().numpy()
        for i in range(3):
            r, px = O.decompress(sample[i * gstride:i * gstride + int(gcb[i])])
            if r != CHUNK or px.tobytes() != host[i * CHUNK:(i + 1) * CHUNK].tobytes():
                print("bench.py --config 5: libzstd does not decode the GPU encoder's chunk", i, file=sys.stderr)
                sys.exit(3)
    for _ in range(args.warmup):
        step()
    eng.enable_timing(1)
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
    Cb = int(gcb.sum())
    total_c = float(Cb)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        cs = torch.tensor([total_c], dtype=torch.float64, device=red_dev)
        dist.all_reduce(cs, op=dist.ReduceOp.SUM)
        total_c = float(cs.item())
    ems, en = eng.kernel_time(hip.K_ENCODE_ZSTD)
    zms, zn = eng.kernel_time(hip.K_DECODE_ZSTD)

    def read_path_launches():
        """The launches behind the cimg_decode_zstd timing id (the zstd read path of a batch: walk, the two lane decoders, replay;
        cimg_decode_zstd itself only for blocks whose plan did not fit), each with its own event pair."""
        parts = {}
        for kid in (hip.K_ZSTD_WALK, hip.K_ZSTD_LIT, hip.K_ZSTD_SEQ, hip.K_ZSTD_REPLAY, hip.K_ZSTD_FUSED):
            pm, pk = eng.kernel_time(kid)
            if pk:
                parts[hip.KERNELS[kid]] = {"launches": pk, "avg_us": round(pm / pk * 1e3, 1)}
        return parts
    own_parts = read_path_launches()
    eng.enable_timing(False)
    if not torch.equal(d_out, d_raw):
        print("bench.py --config 5: pixels differ after the timed region", file=sys.stderr)
        sys.exit(3)
    e_avg_s, z_avg_s = ems / max(en, 1) * 1e-3, zms / max(zn, 1) * 1e-3
    dom_is_enc = ems >= zms
    out = None
    if rank == 0:
        out = {
            "metric": "compress+decompress GB/s (uncompressed side), zstd", "value": round(world * args.steps * 2 * N / elapsed / 1e9, 3), "unit": "GB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[4] share of one rank: 1x{W}x{H} float32 per GPU, zstd clevel {clevel} "
                                   f"({'split planes' if clevel <= 5 else 'one frame per block'}) + byte shuffle, 32 KiB blocks, 4 MiB chunks "
                                   f"({nchunks} chunks, {N // BLOCK} blocks), device-resident, family={args.family}; the engine's own zstd encoder "
                                   f"(format-valid frames, NOT libzstd's bytes) and decoder",
                       "element_dtype": "float32", "uncompressed_bytes_per_gpu": N, "compressed_bytes_per_gpu": Cb,
                       "compression_ratio": round(world * N / total_c, 4),
                       "compress_GBps": round(N / e_avg_s / 1e9, 3) if e_avg_s > 0 else None,
                       "decompress_GBps": round(N / z_avg_s / 1e9, 3) if z_avg_s > 0 else None,
                       "parallelism": f"channels sharded by rank x{world}, no data-path collective"},
            "roofline": {"kernel": "cimg_encode_streams_zstd" if dom_is_enc else "cimg_decode_zstd", "bound": "hbm",
                         "achieved": round((Cb + N) / (e_avg_s if dom_is_enc else z_avg_s) / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round((Cb + N) / (e_avg_s if dom_is_enc else z_avg_s) / 1e9 / HBM_PEAK_GBPS, 5), "traffic": None,
                         "algorithmic_bytes_per_launch": int(Cb + N), "avg_launch_us": round((e_avg_s if dom_is_enc else z_avg_s) * 1e6, 1)},
            "kernels": {"cimg_encode_streams_zstd": {"launches": en, "avg_us": round(e_avg_s * 1e6, 1)},
                        "cimg_decode_zstd": {"launches": zn, "avg_us": round(z_avg_s * 1e6, 1),
                                             "what": "the zstd read path of a batch, first launch to last (one event pair around them)", "launches_of_the_path": own_parts}},
        }
    # ---- chunks as the reference writes them: libzstd on the host (outside the timed region), decoded by the same kernel -----------
    if rank == 0 and world == 1 and have_libzstd and not args.no_cpu_baseline:
        stride = CHUNK + 64
        comp = np.zeros(nchunks * stride, np.uint8)
        cb = np.zeros(nchunks, np.int32)
        p = O.cparams(4, clevel=clevel, blocksize=BLOCK, compcode=O.ZSTD)
        cores = min(visible_cores(), 64)
        teams = min(cores, nchunks)
        t0 = time.perf_counter()
        piece = 32                                                      # progress every 128 MiB (clevel 9 = zstd level 22 is slow)
        for a in range(0, nchunks, piece):
            n = min(piece, nchunks - a)
            r = L.orc_bench_compress(C.byref(p), vp(host[a * CHUNK:]), n, CHUNK, vp(comp[a * stride:]), stride, CHUNK + 32, vp(cb[a:]),
                                     min(teams, n), max(1, cores // min(teams, n)))
            assert r > 0
            print(f"[bench --config 5] libzstd clevel {clevel} made chunks {a}..{a + n - 1} of {nchunks} ({time.perf_counter() - t0:.1f} s)", file=sys.stderr, flush=True)
        t_make = time.perf_counter() - t0
        d_ref = torch.from_numpy(comp).cuda()
        roff = np.arange(nchunks, dtype=np.int64) * stride
        d_out.zero_()
        eng.decompress_device(d_ref.data_ptr(), roff, nbytes, blocksize, d_out.data_ptr(), raw_off, comp_size=cb)
        ok = torch.equal(d_out, d_raw)
        eng.enable_timing(1)
        eng.reset_timing()
        for _ in range(3):
            eng.decompress_device(d_ref.data_ptr(), roff, nbytes, blocksize, d_out.data_ptr(), raw_off, comp_size=cb)
        rms, rn = eng.kernel_time(hip.K_DECODE_ZSTD)
        ref_parts = read_path_launches()
        eng.enable_timing(False)
        thr = min(visible_cores(), 16)
        outb = np.zeros(N, np.uint8)
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < 6.0:
            d = L.orc_bench_decompress(vp(comp), nchunks, stride, vp(cb), vp(outb), CHUNK, min(thr, nchunks), 1)
            assert d == N
            reps += 1
        td = time.perf_counter() - t0
        out["reference_chunks"] = {
            "what": f"chunks as c-blosc2 would write them: libzstd {O.zstd_version()} at clevel {clevel} (zstd level {L.orc_zstd_level_of_clevel(clevel)}) under the checker's chunk layer",
            "compression_ratio": round(N / float(cb.sum()), 4), "decoded_bit_exact_by_cimg_decode_zstd": bool(ok),
            "gpu_decode_launches": ref_parts,
            "gpu_decode_avg_us": round(rms / max(rn, 1) * 1e3, 1), "gpu_decode_GBps": round(N / (rms / max(rn, 1) * 1e-3) / 1e9, 3) if rn else None,
            "host_libzstd_compress_GBps": round(N / t_make / 1e9, 4), "host_compress_threads": cores,
            "host_libzstd_decompress_GBps": round(reps * N / td / 1e9, 3), "host_decompress_threads": thr}
        out["cpu_baseline"] = {"value": round(2 * N / (t_make + td / reps) / 1e9, 4), "unit": "GB/s", "cores": cores, "kind": "port",
                               "compress_GBps": round(N / t_make / 1e9, 4), "decompress_GBps": round(reps * N / td / 1e9, 3),
                               "sample": f"libzstd {O.zstd_version()} under the oracle's chunk layer on the same channel: one compress pass at clevel {clevel} on {cores} threads "
                                         f"({t_make:.1f} s), {reps} decompress passes on {thr} threads ({td:.1f} s)"}
    if rank == 0:
        # (how many blocks went through the fused fallback -- plans that did not fit their slots -- belongs in the record)
        out["zstd_read_path_stats"] = eng.zstd_stats()
    if rank == 0:
        print(json.dumps(out))
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--family", default="tiled", choices=["tiled", "natural", "random", "zero"])
    ap.add_argument("--zstd-clevel", type=int, default=9, help="--config 5: blosc2 clevel the chunks are made with (9 = the reference's default = zstd level 22)")
    ap.add_argument("--config", type=int, default=2, choices=[1, 2, 3, 4, 5],
                    help="BASELINE.json configuration, counted from 1: 1 = configs[0], one 1024^2 uint8 channel, blosclz + shuffle (one 1 MiB chunk: "
                         "the latency floor); 2 = configs[1], 4 x 4096^2 float16 per rank (the headline); "
                         "3 = configs[2], 3 x 8192^2 uint16 blosclz + bitshuffle, the seeded random-access get_chunk -> +1 -> set_chunk loop "
                         "(single-chunk calls; a step = one pass over the 96 chunks); "
                         "4 = configs[3], 64 such images over 8 GPUs = 8 images per rank, with the gather of the finished chunks to rank 0; "
                         "5 = configs[4], 16384^2 float32 zstd: one channel per rank, decode of libzstd-made chunks + compression ratio")
    ap.add_argument("--codec", default=None, choices=["lz4", "blosclz"], help="default: what the configuration names (lz4 for 2 / 4, blosclz for 1 / 3)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="--config 2 with N > 1: weak = every rank owns its own 4 channels (per-GPU work fixed); strong = the ONE image's 32 chunks are "
                         "split over the ranks as cimg/shard.py partition(32, world, rank, 8) does (total work fixed: 4 chunks per GPU at 8)")
    ap.add_argument("--pmc-json", default=None, help="per-launch PMC averages to take roofline.traffic from (profiles/tools/collect.sh)")
    ap.add_argument("--filter", default=None, choices=["shuffle", "bitshuffle", "none"],
                    help="default: what the configuration names (byte shuffle; bitshuffle for 3)")
    args = ap.parse_args()
    if args.config in (1, 3) and args.steps == 50 and "--steps" not in sys.argv:
        args.steps = 200 if args.config == 1 else 5            # (a step of configs[0] is one 1 MiB chunk, of configs[2] a pass over 96 chunks)
    if args.config in (1, 3) and "--warmup" not in sys.argv:
        args.warmup = 20 if args.config == 1 else 1

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
    # CIMG_BENCH_REHEARSAL=nccl: the same, but over RCCL -- two ranks sharing one device, so that RCCL's send/recv of the exchange
    # step has run at least once where no multi-GPU node is to be had (if RCCL accepts two ranks per device: profiles/r05/nccl_rehearsal.txt)
    rehearsal = os.environ.get("CIMG_BENCH_REHEARSAL") is not None
    rehearsal_nccl = os.environ.get("CIMG_BENCH_REHEARSAL") == "nccl"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    red_dev = "cpu" if rehearsal and not rehearsal_nccl else "cuda"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal and not rehearsal_nccl:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    if rehearsal_nccl:
        rehearsal = False                                      # (from here on the run IS an nccl run; its numbers still mean nothing: one card)

    if args.config == 5:
        return run_config5(args, rank, world, local_rank, dist, red_dev)
    if args.config == 1:
        return run_config1(args, rank, world, local_rank, dist, red_dev)
    if args.config == 3:
        return run_config3(args, rank, world, local_rank, dist, red_dev)
    args.codec = args.codec or "lz4"
    args.filter = args.filter or "shuffle"

    from cimg import hip, synth

    # ---- inputs: this rank's 4 channels, resident in HBM ------------------------------------------------
    gen = getattr(synth, args.family + "_channel")
    images = 8 if args.config == 4 else 1                      # configs[3]: 64 images over 8 ranks
    chans = []
    for img in range(images):
        for c in range(CHANNELS):
            if args.family == "zero":
                chans.append(gen(DTYPE, WIDTH, HEIGHT))
            elif args.config == 4:                              # seeds 1234 + 4 * image + channel
                chans.append(gen(DTYPE, WIDTH, HEIGHT, c=c, seed=1234 + 4 * (images * rank + img)))
            else:
                chans.append(gen(DTYPE, WIDTH, HEIGHT, c=CHANNELS * rank + c))
    host = np.concatenate([c.view(np.uint8).ravel() for c in chans])
    strong = args.scaling == "strong" and args.config == 2
    if args.scaling == "strong" and not strong:
        print("bench.py: --scaling strong is defined for --config 2 (one image's 32 chunks over the ranks)", file=sys.stderr)
        sys.exit(2)
    N_image = host.size
    if strong:
        # ONE image for the whole job (rank 0's channels, the N = 1 workload): SURVEY.md section 8e -- channels round-robin while there
        # are at least as many as ranks, chunk-granular below that (8 GPUs: chunks r, r + 8, r + 16, r + 24)
        from cimg import shard
        chans = [gen(DTYPE, WIDTH, HEIGHT) if args.family == "zero" else gen(DTYPE, WIDTH, HEIGHT, c=c) for c in range(CHANNELS)]
        whole = np.concatenate([c.view(np.uint8).ravel() for c in chans])
        mine_chunks = shard.partition(whole.size // CHUNK, world, rank, items_per_group=whole.size // CHUNK // CHANNELS)
        host = np.concatenate([whole[int(g) * CHUNK:(int(g) + 1) * CHUNK] for g in mine_chunks]) if len(mine_chunks) else np.zeros(0, np.uint8)
        N_image = whole.size
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
    p = hip.cparams(np.dtype(DTYPE).itemsize, clevel=9, blocksize=BLOCK, compcode=hip.LZ4 if args.codec == "lz4" else hip.BLOSCLZ, filters=(0, 0, 0, 0, 0, filt))

    # One step = one pass of the hot path over the batch: compress every chunk, decompress every chunk, results (sizes,
    # status words) on the host.  The two batches are enqueued back to back through the _begin / _fetch form of the
    # device-resident calls (include/cimg_hip.h): the decode kernels follow the encode kernels in stream order, and the host's
    # share -- planning, launches, the wait -- hides behind them (CIMG_BENCH_SYNC_CALLS=1: the plain calls, one wait each).
    sync_calls = bool(os.environ.get("CIMG_BENCH_SYNC_CALLS"))

    # (the arguments are marshalled once -- cimg/hip.py: roundtrip_calls -- so that a step is the four C calls and nothing else)
    fast_step = eng.roundtrip_calls(p, d_raw.data_ptr(), raw_off, nbytes, d_comp.data_ptr(), comp_off, destsize, blocksize, d_out.data_ptr())

    def step():
        if sync_calls:
            cb = eng.compress_device(p, d_raw.data_ptr(), raw_off, nbytes, d_comp.data_ptr(), comp_off, destsize)
            eng.decompress_device(d_comp.data_ptr(), comp_off, nbytes, blocksize, d_out.data_ptr(), raw_off)
            return cb
        return fast_step()

    torch.cuda.synchronize()
    t_first = time.perf_counter()
    cbytes = step()                                                # the COLD call: allocations, descriptor upload, LDS opt-in, first launches
    torch.cuda.synchronize()
    first_call_ms = (time.perf_counter() - t_first) * 1e3
    no_verify = bool(os.environ.get("CIMG_BENCH_NO_VERIFY"))       # kernel-timing experiments with deliberately broken builds only
    if not torch.equal(d_out, d_raw) and not no_verify:
        print("bench.py: decompressed pixels differ from the input -- refusing to report a number", file=sys.stderr)
        sys.exit(3)
    for _ in range(args.warmup):
        step()

    # kernel durations: HIP events on the engine's stream around every kernel of every TIMING_PERIOD-th batch call of
    # the timed region (an event record costs ~5 us of stream time; all eight per step were 37 us of an 800 us step)
    eng.enable_timing(0 if os.environ.get("CIMG_BENCH_NO_EVENTS") else TIMING_PERIOD)
    eng.reset_timing()
    # (measurement hygiene: with torch imported the interpreter holds ~10^6 objects, and ONE full pass of Python's cyclic garbage
    # collector over them -- which the per-step result arrays trigger every few hundred steps -- stalls the host for ~40 ms: a
    # 200-step run measured 0.65 ms per step against 0.46 for 30 or 1000 steps.  Nothing of the step is skipped.)
    import gc
    gc.collect()
    gc.freeze()
    gc.disable()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    per_step = []
    t0 = time.perf_counter()
    for k in range(args.steps):
        if k == args.steps - 1:
            # the LAST step's pixels are the ones compared below: d_out is cleared in front of it (inside the timed region: 25 us of
            # memset once), so a decode that wrote nothing could not pass on the pixels an earlier step left there
            d_out.zero_()
            torch.cuda.synchronize()
        ts = time.perf_counter()
        step()
        per_step.append(time.perf_counter() - ts)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        csum = torch.tensor([float(cbytes.sum())], dtype=torch.float64, device=red_dev)
        dist.all_reduce(csum, op=dist.ReduceOp.SUM)
        total_c = float(csum.item())
    else:
        total_c = float(cbytes.sum())

    cbytes = np.array(cbytes, copy=True)            # (the marshalled step hands back ONE array, rewritten by every later call: keep this batch's sizes)
    ktimes = [eng.kernel_time(k) for k in range(4)]
    ksamples = [eng.kernel_samples(k) for k in range(4)]
    dstats = eng.decode_stats()
    eng.enable_timing(False)
    # per-rank kernel time (strong scaling shows the latency floor: a rank with 4 chunks still pays a whole chain round per launch)
    per_rank_us = None
    mine_us = [float(np.median(ksamples[hip.K_ENCODE])) * 1e3 if len(ksamples[hip.K_ENCODE]) else 0.0,
               float(np.median(ksamples[hip.K_DECODE])) * 1e3 if len(ksamples[hip.K_DECODE]) else 0.0]
    if dist is not None:
        tk = torch.zeros(world, 2, dtype=torch.float64, device=red_dev)
        tk[rank, 0], tk[rank, 1] = mine_us[0], mine_us[1]
        dist.all_reduce(tk, op=dist.ReduceOp.SUM)
        per_rank_us = [[round(float(tk[r, 0]), 2), round(float(tk[r, 1]), 2)] for r in range(world)]

    # ---- the exchange step of SURVEY.md section 8e (N > 1): every rank's finished chunks travel to rank 0, packed,
    # point to point with exact sizes (cimg/shard.py: gather_chunks; "nccl" = RCCL send/recv over xGMI).  Timed on its own,
    # after the codec region: it belongs to a caller that wants the whole result on one device, not to the codec path.
    # It runs LAST, behind a watchdog (below): nothing it does -- a refusal, a hang of the communicator -- may cost the line its
    # codec measurement.
    def run_exchange():
        exchange = None
        # (test hook, tools/... rehearsals only: CIMG_BENCH_EXCHANGE_FAULT = "raise:<rank>" / "hang:<rank>" injects a failure of the exchange
        # on one rank, to show that the line survives it)
        fault = os.environ.get("CIMG_BENCH_EXCHANGE_FAULT", "")
        if fault == f"raise:{rank}":
            raise RuntimeError("injected exchange failure")
        if fault == f"hang:{rank}":
            time.sleep(10 ** 6)
        if dist is not None:
            from cimg import shard
            per_group = nchunks
            if strong:
                n_items = N_image // CHUNK
                per_group = n_items // CHANNELS
                mine = shard.partition(n_items, world, rank, items_per_group=per_group)
            else:
                n_items = world * nchunks
                mine = shard.partition(n_items, world, rank, items_per_group=nchunks)    # rank r owns its own images: items r*nchunks ...
            sizes_all = shard.gather_sizes(dist, mine, cbytes, n_items, device=red_dev)
            comp_view = d_comp if not rehearsal else d_comp.cpu()
            xdev = "cpu" if rehearsal else "cuda"
            for rep in range(3):                                    # first pass warms the communicator
                dist.barrier()
                torch.cuda.synchronize()
                tx = time.perf_counter()
                got = shard.gather_chunks(dist, world, rank, mine, comp_view, comp_off, sizes_all, n_items, dst=0,
                                          items_per_group=per_group, device=xdev, as_tensor=True)
                torch.cuda.synchronize()
                dist.barrier()
                tx = time.perf_counter() - tx
            tmax = torch.tensor([tx], dtype=torch.float64, device=red_dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            moved = int(sizes_all.sum() - sizes_all[shard.partition(n_items, world, 0, per_group)].sum())
            if rank == 0:
                whole, offs, szs = got
                probe = int(mine[0]) if len(mine) else 0
                ok = whole.numel() == int(sizes_all.sum())
                exchange = {"what": "finished chunks of every rank gathered to rank 0 (packed compressed bytes, exact sizes, "
                                    "batch_isend_irecv)", "bytes_moved": moved, "seconds": round(float(tmax.item()), 6),
                            "exchange_GBps": round(moved / float(tmax.item()) / 1e9, 3) if moved else None,
                            "backend": "gloo (rehearsal on one GPU: meaningless as a number)" if rehearsal else ("nccl (RCCL), REHEARSAL with every rank on one GPU: meaningless as a number" if os.environ.get("CIMG_BENCH_REHEARSAL") == "nccl" else "nccl (RCCL over xGMI)"),
                            "complete": bool(ok)}
        return exchange

    if not torch.equal(d_out, d_raw) and not no_verify:
        print("bench.py: pixels differ after the timed region", file=sys.stderr)
        sys.exit(3)

    # ---- a measured copy peak beside the 8 TB/s specification, and the OTHER face of the codec (natural family) in the same line ----
    copy_peak = natural = dtypes = None
    headline = args.family == "tiled" and args.filter == "shuffle" and args.codec == "lz4" and args.config == 2 and not strong
    if rank == 0:
        copy_peak = measure_copy_peak()
    if rank == 0 and world == 1 and headline and not os.environ.get("CIMG_BENCH_NO_NATURAL"):
        nat = np.concatenate([synth.natural_channel(DTYPE, WIDTH, HEIGHT, c=c).view(np.uint8).ravel() for c in range(CHANNELS)])
        d_raw.copy_(torch.from_numpy(nat).cuda())
        step(); step()
        d_out.zero_()
        torch.cuda.synchronize()
        eng.enable_timing(1); eng.reset_timing()
        tn = time.perf_counter()
        nsteps = 8
        for _ in range(nsteps):
            ncb = step()
        torch.cuda.synchronize()
        tn = time.perf_counter() - tn
        ne, nd = eng.kernel_samples(hip.K_ENCODE), eng.kernel_samples(hip.K_DECODE)
        eng.enable_timing(False)
        ok = torch.equal(d_out, d_raw)
        natural = {"what": "the same geometry on the 'natural' family (smooth gradients + noise in every byte plane: what photographs look like; cimg/synth.py), "
                           f"{nsteps} steps outside the timed region", "value": round(nsteps * 2 * N / tn / 1e9, 3), "unit": "GB/s",
                   "encode_us": round(float(np.median(ne)) * 1e3, 1) if len(ne) else None, "decode_us": round(float(np.median(nd)) * 1e3, 1) if len(nd) else None,
                   "compression_ratio": round(N / float(ncb.sum()), 4), "pixels_verified": bool(ok)}
        if not args.no_cpu_baseline:
            # the CPU beside the GPU on the HARD data too: the same 16-thread port over the same natural chunks
            natural["cpu_baseline"] = cpu_baseline(nat, budget_s=5.0, policy="share")
            natural["gpu_over_cpu"] = round(natural["value"] / natural["cpu_baseline"]["value"], 2)
        # configs[1]'s geometry on the pixel types north_star names beside float16 (two steps each, outside the timed region):
        # byte-wide photographs -- ONE 32 KiB stream a block, three chains a CU -- and tiled float32, configs[4]'s element type
        dtypes = {}
        for dt_name, fam in (("uint8", "natural"), ("float32", "tiled")):
            dt = np.dtype(dt_name)
            # (as channels of 4096 x 4096, like the headline's: one tall image would cost the generator gigabytes of float64 temporaries)
            per = WIDTH * HEIGHT * dt.itemsize
            pix = np.concatenate([np.ascontiguousarray(getattr(synth, fam + "_channel")(dt.type, WIDTH, HEIGHT, c=k)).view(np.uint8).ravel() for k in range(N // per)])
            d_raw.copy_(torch.from_numpy(pix).cuda())
            pd = hip.cparams(dt.itemsize, clevel=9, blocksize=BLOCK, compcode=hip.LZ4, filters=(0, 0, 0, 0, 0, hip.SHUFFLE))
            def dstep():
                cb_ = eng.compress_device(pd, d_raw.data_ptr(), raw_off, nbytes, d_comp.data_ptr(), comp_off, destsize)
                eng.decompress_device(d_comp.data_ptr(), comp_off, nbytes, blocksize, d_out.data_ptr(), raw_off, comp_size=cb_)
                return cb_
            dstep()
            d_out.zero_()
            torch.cuda.synchronize()
            eng.enable_timing(1); eng.reset_timing()
            td = time.perf_counter()
            for _ in range(2):
                dcb = dstep()
            torch.cuda.synchronize()
            td = time.perf_counter() - td
            de, dd = eng.kernel_samples(hip.K_ENCODE), eng.kernel_samples(hip.K_DECODE)
            eng.enable_timing(False)
            dtypes[f"{dt_name}_{fam}"] = {"value": round(2 * 2 * N / td / 1e9, 3), "unit": "GB/s (compress + decompress, uncompressed side, plain calls)",
                                          "encode_us": round(float(np.median(de)) * 1e3, 1) if len(de) else None,
                                          "decode_us": round(float(np.median(dd)) * 1e3, 1) if len(dd) else None,
                                          "compression_ratio": round(N / float(np.asarray(dcb).sum()), 4), "pixels_verified": bool(torch.equal(d_out, d_raw))}
        d_raw.copy_(torch.from_numpy(host).cuda())
        torch.cuda.synchronize()

    if rank == 0:
        Cb = float(cbytes.sum())
        algo = {hip.K_ENCODE: N + Cb, hip.K_LAYOUT: 0.0, hip.K_EMIT: 0.0, hip.K_DECODE: Cb + N}
        # the decode entry of the engine's timers covers cimg_decode_lean + whatever cimg_decode_blocks had to do behind it; it is
        # reported under the kernel that did the work (engine statistics: blocks the lean launch left over)
        lean_only = dstats["lean_batches"] > 0 and dstats["blocks_left_to_general"] == 0
        dec_name = "cimg_decode_lean" if lean_only else ("cimg_decode_lean + cimg_decode_blocks" if dstats["lean_batches"] > 0 else "cimg_decode_blocks")
        enc_name = "cimg_encode_streams" if args.codec == "lz4" else "cimg_encode_streams_blosclz"
        names = {hip.K_ENCODE: enc_name, hip.K_LAYOUT: "cimg_layout_chunks", hip.K_EMIT: "cimg_emit_blocks", hip.K_DECODE: dec_name}
        kernels = {}
        for k, (ms, n) in enumerate(ktimes):
            avg = ms / n if n else 0.0
            med = float(np.median(ksamples[k])) if len(ksamples[k]) else 0.0
            kernels[names[k]] = {
                "launches": n, "avg_us": round(avg * 1e3, 2), "median_us": round(med * 1e3, 2),
                "algorithmic_GBps": round(algo[k] / (avg * 1e-3) / 1e9, 1) if avg > 0 and algo[k] else None}
        dom = max(range(4), key=lambda k: ktimes[k][0])
        dom_avg_s = ktimes[dom][0] / max(ktimes[dom][1], 1) * 1e-3
        achieved = algo[dom] / dom_avg_s / 1e9 if dom_avg_s > 0 else 0.0
        dec_avg_s = ktimes[hip.K_DECODE][0] / max(ktimes[hip.K_DECODE][1], 1) * 1e-3
        dec_med_s = (float(np.median(ksamples[hip.K_DECODE])) if len(ksamples[hip.K_DECODE]) else 0.0) * 1e-3
        traffic, traffic_src = pmc_traffic(hip.KERNELS[dom], args.pmc_json) if headline else (None, None)
        dec_traffic, _ = pmc_traffic(hip.KERNELS[hip.K_DECODE], args.pmc_json) if headline else (None, None)
        N_job = N_image if strong else world * N                  # uncompressed bytes one step of the whole job processes
        out = {
            "metric": "compress+decompress GB/s (uncompressed side)" + (" -- INVALID: pixels not verified (CIMG_BENCH_NO_VERIFY)" if no_verify else ""),
            "value": round(args.steps * 2 * N_job / elapsed / 1e9, 3),
            "unit": "GB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "ms_per_step_median": round(_median(per_step) * 1e3, 4),
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{'BASELINE configs[3] share of one rank: 8 images x ' if args.config == 4 else ''}"
                                   f"{len(chans)}x{WIDTH}x{HEIGHT} float16 {'for the WHOLE job (strong scaling), this rank holds' if strong else 'per GPU'}, "
                                   f"{args.codec} clevel 9 + {FILTER_TEXT[args.filter]}, "
                                   f"32 KiB blocks, 4 MiB chunks ({nchunks} chunks, {N // BLOCK} blocks, {2 * N // BLOCK} streams), "
                                   f"device-resident, family={args.family}",
                       "element_dtype": "float16", "uncompressed_bytes_per_gpu": N, "compressed_bytes_per_gpu": int(Cb),
                       "compression_ratio": round(world * N / total_c, 4) if total_c else None,
                       "roundtrip_GBps": round(args.steps * N_job / elapsed / 1e9, 3),
                       "first_call_ms": round(first_call_ms, 3),       # the steady state rides the descriptor cache and warm buffers; this is the cold step
                       "parallelism": (f"strong scaling: ONE image's {N_image // CHUNK} chunks split over {world} rank(s) as cimg/shard.py partition does "
                                       f"(this rank: {nchunks}), no data-path collective" if strong else
                                       f"chunks sharded by rank x{world}, no data-path collective"),
                       "per_rank_kernel_median_us": ({"encode_decode": per_rank_us} if per_rank_us else None)},
            "roofline": {"kernel": names[dom], "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                         "measured_copy_peak": copy_peak, "frac_of_measured_peak": round(achieved / copy_peak, 4) if copy_peak else None,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": int(algo[dom]), "avg_launch_us": round(dom_avg_s * 1e6, 2),
                         "note": ("the launch contains chunk layout + emit (separate kernels until round 2: encode 349 + layout 7 + emit 29 us; "
                                  "CIMG_NO_ASSEMBLE_IN_LAUNCH=1 runs them separately again)") if kernels["cimg_emit_blocks"]["launches"] == 0 and dom == hip.K_ENCODE else None},
            "roofline_decode": {"kernel": dec_name, "bound": "hbm",
                                "achieved": round((Cb + N) / dec_avg_s / 1e9, 1) if dec_avg_s > 0 else None,
                                "output_side": round(N / dec_avg_s / 1e9, 1) if dec_avg_s > 0 else None,
                                "peak": HBM_PEAK_GBPS, "unit": "GB/s", "traffic": dec_traffic,
                                "frac": round((Cb + N) / dec_avg_s / 1e9 / HBM_PEAK_GBPS, 4) if dec_avg_s > 0 else None,
                                "avg_launch_us": round(dec_avg_s * 1e6, 2), "median_launch_us": round(dec_med_s * 1e6, 2),
                                "measured_copy_peak": copy_peak,
                                "frac_of_measured_peak": round((Cb + N) / dec_avg_s / 1e9 / copy_peak, 4) if dec_avg_s > 0 and copy_peak else None},
            "kernels": kernels,
            "natural_family": natural,
            "dtypes": dtypes,
            "last_step_verified": "d_out cleared in front of the last timed step, pixels compared with the input after it",
            "exchange": None,
            "kernel_timing": f"HIP events around every kernel of every {TIMING_PERIOD}th batch call inside the timed region",
            "step_calls": ("cimg_compress_batch_device + cimg_decompress_batch_device (one wait each)" if sync_calls else
                           "cimg_compress_batch_device_begin, cimg_decompress_batch_device_begin, then both _fetch (sizes and status on the host every step)"),
        }
        if not args.no_cpu_baseline and world == 1 and args.filter == "shuffle" and args.config == 2 and args.codec == "lz4":
            # three labelled figures, none capped silently: this GPU's share of the host (16 threads), every visible hardware
            # thread, and the reference's own call structure (serial chunks, hw/2 threads inside a chunk for encode, 1 thread decode)
            out["cpu_baseline"] = cpu_baseline(host, budget_s=10.0, policy="share")
            # "all cores": every thread count from 32 up to everything visible is tried (a container may see 256 hardware threads and
            # be scheduled on far fewer; more threads than that only thrash) and the BEST is reported, with the sweep beside it
            sweep = {}
            tries = sorted({t for t in (32, 64, 128, visible_cores()) if 16 < t <= visible_cores()}) or [visible_cores()]
            for t in tries:
                sweep[t] = cpu_baseline(host, budget_s=max(2.0, 8.0 / len(tries)), policy="all_cores", threads=t)
            best_t = max(sweep, key=lambda t: sweep[t]["value"])
            out["cpu_baseline_all_cores"] = dict(sweep[best_t], sweep_GBps={str(t): sweep[t]["value"] for t in tries})
            out["cpu_baseline_reference_policy"] = cpu_baseline(host, budget_s=8.0, policy="reference")
            best_cpu = max(out[k]["value"] for k in ("cpu_baseline", "cpu_baseline_all_cores", "cpu_baseline_reference_policy"))
            out["gpu_over_cpu"] = {"vs_share_16_threads": round(out["value"] / out["cpu_baseline"]["value"], 2),
                                   "vs_all_cores": round(out["value"] / out["cpu_baseline_all_cores"]["value"], 2),
                                   "vs_reference_policy": round(out["value"] / out["cpu_baseline_reference_policy"]["value"], 2),
                                   "least_flattering": round(out["value"] / best_cpu, 2)}
            real = cblosc2_baseline(host)                     # only where a c-blosc2 library is installed
            if real is not None:
                out["cpu_baseline_cblosc2"] = real
    if dist is not None:
        # the exchange, behind a watchdog: a rank whose exchange has not come back after EXCHANGE_TIMEOUT_S leaves -- rank 0 with the
        # line it has (exchange: the reason), the others silently -- instead of sitting in a collective until the job is killed
        import threading

        def give_up():
            if rank == 0:
                out["exchange"] = {"error": f"the exchange step did not complete within {EXCHANGE_TIMEOUT_S} s (the codec figures above are unaffected)"}
                print(json.dumps(out), flush=True)
            os._exit(0)
        timer = threading.Timer(EXCHANGE_TIMEOUT_S, give_up)
        timer.daemon = True
        timer.start()
        try:
            ex = run_exchange()
        except Exception as err:                                # noqa: BLE001 -- whatever the communicator says goes into the record
            ex = {"error": f"{type(err).__name__}: {str(err).splitlines()[0][:300] if str(err) else ''}"}
        timer.cancel()
        if rank == 0:
            out["exchange"] = ex
        if isinstance(ex, dict) and "error" in ex:
            # (the communicator is in an unknown state: no further collective -- the line, then out)
            if rank == 0:
                print(json.dumps(out), flush=True)
            os._exit(0)
    if rank == 0:
        print(json.dumps(out), flush=True)
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
