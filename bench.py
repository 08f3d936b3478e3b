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


def cpu_baseline(host, budget_s=10.0, policy="share", typesize=2, compcode=None, chunk=None, threads=None):
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
    else:
        cores = max(1, avail // 2)
        enc, dec = (1, cores), (1, 1)
    p = O.cparams(typesize, clevel=9, blocksize=BLOCK, compcode=O.LZ4 if compcode is None else compcode)
    stride = chunk + 64
    comp = np.zeros(nchunks * stride, np.uint8)
    out = np.zeros(host.size, np.uint8)
    cb = np.zeros(nchunks, np.int32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)

    def one_pass():
        t0 = time.perf_counter()
        r = L.orc_bench_compress(C.byref(p), vp(host), nchunks, chunk, vp(comp), stride, chunk + 32, vp(cb), enc[0], enc[1])
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
            "sample": f"oracle (oracle/ CPU restatement at -O3, not c-blosc2): the same {nchunks} chunks x {chunk >> 20} MiB, {reps} passes in "
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
PMC_KERNELS = {"cimg_decode_blocks": ("cimg_decode_lean", "cimg_decode_blocks")}


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
    path = explicit or os.path.join(ROOT, "profiles", "r03", "pmc_per_launch.json")
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
                        "cimg_decode_zstd": {"launches": zn, "avg_us": round(z_avg_s * 1e6, 1)}},
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
            "gpu_decode_avg_us": round(rms / max(rn, 1) * 1e3, 1), "gpu_decode_GBps": round(N / (rms / max(rn, 1) * 1e-3) / 1e9, 3) if rn else None,
            "host_libzstd_compress_GBps": round(N / t_make / 1e9, 4), "host_compress_threads": cores,
            "host_libzstd_decompress_GBps": round(reps * N / td / 1e9, 3), "host_decompress_threads": thr}
        out["cpu_baseline"] = {"value": round(2 * N / (t_make + td / reps) / 1e9, 4), "unit": "GB/s", "cores": cores, "kind": "port",
                               "compress_GBps": round(N / t_make / 1e9, 4), "decompress_GBps": round(reps * N / td / 1e9, 3),
                               "sample": f"libzstd {O.zstd_version()} under the oracle's chunk layer on the same channel: one compress pass at clevel {clevel} on {cores} threads "
                                         f"({t_make:.1f} s), {reps} decompress passes on {thr} threads ({td:.1f} s)"}
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
    ap.add_argument("--config", type=int, default=2, choices=[2, 4, 5],
                    help="BASELINE.json configuration, counted from 1: 2 = configs[1], 4 x 4096^2 float16 per rank (the headline); "
                         "4 = configs[3], 64 such images over 8 GPUs = 8 images per rank, with the gather of the finished chunks to rank 0; "
                         "5 = configs[4], 16384^2 float32 zstd: one channel per rank, decode of libzstd-made chunks + compression ratio")
    ap.add_argument("--codec", default="lz4", choices=["lz4", "blosclz"], help="not part of the headline (BASELINE configs[1] is lz4)")
    ap.add_argument("--pmc-json", default=None, help="per-launch PMC averages to take roofline.traffic from (profiles/tools/collect.sh)")
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

    if args.config == 5:
        return run_config5(args, rank, world, local_rank, dist, red_dev)

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
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
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

    ktimes = [eng.kernel_time(k) for k in range(4)]
    eng.enable_timing(False)

    # ---- the exchange step of SURVEY.md section 8e (N > 1): every rank's finished chunks travel to rank 0, packed,
    # point to point with exact sizes (cimg/shard.py: gather_chunks; "nccl" = RCCL send/recv over xGMI).  Timed on its own,
    # after the codec region: it belongs to a caller that wants the whole result on one device, not to the codec path.
    exchange = None
    if dist is not None:
        from cimg import shard
        n_items = world * nchunks
        mine = shard.partition(n_items, world, rank, items_per_group=nchunks)        # rank r owns its own images: items r*nchunks ...
        sizes_all = shard.gather_sizes(dist, mine, cbytes, n_items, device=red_dev)
        comp_view = d_comp if not rehearsal else d_comp.cpu()
        xdev = "cpu" if rehearsal else "cuda"
        for rep in range(3):                                    # first pass warms the communicator
            dist.barrier()
            torch.cuda.synchronize()
            tx = time.perf_counter()
            got = shard.gather_chunks(dist, world, rank, mine, comp_view, comp_off, sizes_all, n_items, dst=0,
                                      items_per_group=nchunks, device=xdev, as_tensor=True)
            torch.cuda.synchronize()
            dist.barrier()
            tx = time.perf_counter() - tx
        tmax = torch.tensor([tx], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        moved = int(sizes_all.sum() - sizes_all[shard.partition(n_items, world, 0, nchunks)].sum())
        if rank == 0:
            whole, offs, szs = got
            probe = int(mine[0]) if len(mine) else 0
            ok = whole.numel() == int(sizes_all.sum())
            exchange = {"what": "finished chunks of every rank gathered to rank 0 (packed compressed bytes, exact sizes, "
                                "batch_isend_irecv)", "bytes_moved": moved, "seconds": round(float(tmax.item()), 6),
                        "exchange_GBps": round(moved / float(tmax.item()) / 1e9, 3) if moved else None,
                        "backend": "gloo (rehearsal on one GPU: meaningless as a number)" if rehearsal else "nccl (RCCL over xGMI)",
                        "complete": bool(ok)}
    if not torch.equal(d_out, d_raw) and not no_verify:
        print("bench.py: pixels differ after the timed region", file=sys.stderr)
        sys.exit(3)

    if rank == 0:
        Cb = float(cbytes.sum())
        algo = {hip.K_ENCODE: N + Cb, hip.K_LAYOUT: 0.0, hip.K_EMIT: 0.0, hip.K_DECODE: Cb + N}
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
        headline = args.family == "tiled" and args.filter == "shuffle" and args.codec == "lz4" and args.config == 2
        traffic, traffic_src = pmc_traffic(hip.KERNELS[dom], args.pmc_json) if headline else (None, None)
        dec_traffic, _ = pmc_traffic(hip.KERNELS[hip.K_DECODE], args.pmc_json) if headline else (None, None)
        out = {
            "metric": "compress+decompress GB/s (uncompressed side)" + (" -- INVALID: pixels not verified (CIMG_BENCH_NO_VERIFY)" if no_verify else ""),
            "value": round(world * args.steps * 2 * N / elapsed / 1e9, 3),
            "unit": "GB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{'BASELINE configs[3] share of one rank: 8 images x ' if args.config == 4 else ''}"
                                   f"{len(chans)}x{WIDTH}x{HEIGHT} float16 per GPU, {args.codec} clevel 9 + {FILTER_TEXT[args.filter]}, "
                                   f"32 KiB blocks, 4 MiB chunks ({nchunks} chunks, {N // BLOCK} blocks, {2 * N // BLOCK} streams), "
                                   f"device-resident, family={args.family}",
                       "element_dtype": "float16", "uncompressed_bytes_per_gpu": N, "compressed_bytes_per_gpu": int(Cb),
                       "compression_ratio": round(world * N / total_c, 4) if total_c else None,
                       "roundtrip_GBps": round(world * args.steps * N / elapsed / 1e9, 3),
                       "first_call_ms": round(first_call_ms, 3),       # the steady state rides the descriptor cache and warm buffers; this is the cold step
                       "parallelism": f"chunks sharded by rank x{world}, no data-path collective"},
            "roofline": {"kernel": hip.KERNELS[dom], "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": int(algo[dom]), "avg_launch_us": round(dom_avg_s * 1e6, 2),
                         "note": ("the launch contains chunk layout + emit (separate kernels until round 2: encode 349 + layout 7 + emit 29 us; "
                                  "CIMG_NO_ASSEMBLE_IN_LAUNCH=1 runs them separately again)") if kernels["cimg_emit_blocks"]["launches"] == 0 and dom == hip.K_ENCODE else None},
            "roofline_decode": {"kernel": hip.KERNELS[hip.K_DECODE], "bound": "hbm",
                                "achieved": round((Cb + N) / dec_avg_s / 1e9, 1) if dec_avg_s > 0 else None,
                                "output_side": round(N / dec_avg_s / 1e9, 1) if dec_avg_s > 0 else None,
                                "peak": HBM_PEAK_GBPS, "unit": "GB/s", "traffic": dec_traffic,
                                "frac": round((Cb + N) / dec_avg_s / 1e9 / HBM_PEAK_GBPS, 4) if dec_avg_s > 0 else None},
            "kernels": kernels,
            "exchange": exchange,
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
        print(json.dumps(out))
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
