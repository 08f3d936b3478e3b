// decode_kernel.h -- blosc2 chunk decode for gfx950, one 256-thread workgroup per block.
//
// Replaces what the reference reaches through blosc2_decompress_ctx (blosc2/wrapper.h:246, called
// from schunk.h:164-180 per chunk, serially, on one CPU thread).  Here a whole batch of chunks is
// decoded by one launch: workgroup b handles block b of the batch-wide block numbering.
//
//   phase A  every wave reads the chunk header + bstarts[j] (wave-uniform loads), walks the
//            per-stream int32 csize words of its block and stages its streams in LDS:
//              run / zero stream  -> fill
//              raw stream         -> 16-byte global loads -> LDS
//              LZ4 stream         -> compressed bytes are parked at the END of the stream's LDS
//                                    region and decoded *in place* towards the front by one wave
//                                    (64-byte register window over the token stream, lane-parallel
//                                    literal and match copies, overlap-safe for offset < 64)
//   barrier
//   phase B  all four waves undo the byte shuffle straight out of LDS (v_perm byte transposes for
//            typesize 2 and 4) and store the pixels with 16-byte coalesced writes.
//
// HBM traffic per block: compressed bytes read once, pixels written once (the algorithmic bytes).
// Memcpyed and special-zero chunks skip LDS.  Written in the wave.h vocabulary; see wave.h for the
// host-emulation build used by tests/emu.
#pragma once
#include "codec_types.h"
#include "wave.h"

namespace cimg {

struct DecodeArgs {
    const ChunkDesc* descs;
    int32_t nchunks;
    const uint8_t* comp;      // compressed chunks live at comp + desc.comp_off
    uint8_t* raw;             // pixels go to raw + desc.raw_off
    int32_t* status;          // per chunk: 0 ok, <0 blosc2 error code
    int32_t lds_bytes;        // dynamic LDS size the launch provides
    uint64_t* dbg;            // diagnostics only: per-workgroup time stamps (nullptr in production)
};

CIMG_HD int round16(int x) { return (x + 15) & ~15; }
// room between the end of a stream's output and the end of its parked compressed bytes that keeps
// the in-place write pointer behind the read pointer for every valid LZ4 block
CIMG_HD int inplace_margin(int n) { return round16(n / 255) + 64; }
CIMG_HD int region_stride(int neblock) { return round16(neblock) + inplace_margin(neblock); }
inline int decode_lds_bytes(int blocksize, int typesize)
{
    int a = round16(blocksize) + inplace_margin(blocksize);
    if (typesize >= 1 && typesize <= MAX_STREAMS) {
        const int ne = blocksize / typesize;
        const int b = typesize * (round16(ne) + inplace_margin(ne));
        if (b > a) a = b;
    }
    return a + 32;
}

// find the chunk that owns batch-wide block index b (descs are ordered by blk0)
CIMG_DEV int find_chunk(const ChunkDesc* descs, int nchunks, int b)
{
    int lo = 0, hi = nchunks - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].blk0 <= b) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// ---- wave-cooperative copies ---------------------------------------------------------------------
// global -> LDS, any source alignment, LDS offset 16-byte aligned
CIMG_DEV void wave_copy_g2l(const uint8_t* g, uint8_t* lds, int off, int nbytes)
{
    const int units = nbytes >> 4;
    for (int u0 = 0; u0 < units; u0 += 256) {
        LV<u128> t0, t1, t2, t3;
        FOR_LANES(l) {
            const int u = u0 + l;
            if (u < units) t0[l] = ld128u(g + 16 * u);
            if (u + 64 < units) t1[l] = ld128u(g + 16 * (u + 64));
            if (u + 128 < units) t2[l] = ld128u(g + 16 * (u + 128));
            if (u + 192 < units) t3[l] = ld128u(g + 16 * (u + 192));
        }
        FOR_LANES(l) {
            const int u = u0 + l;
            if (u < units) st128a(lds + off + 16 * u, t0[l]);
            if (u + 64 < units) st128a(lds + off + 16 * (u + 64), t1[l]);
            if (u + 128 < units) st128a(lds + off + 16 * (u + 128), t2[l]);
            if (u + 192 < units) st128a(lds + off + 16 * (u + 192), t3[l]);
        }
    }
    const int done = units << 4;
    FOR_LANES(l) { if (done + l < nbytes) lds[off + done + l] = g[done + l]; }
}

CIMG_DEV void wave_fill_lds(uint8_t* lds, int off, int nbytes, uint32_t byte)
{
    const uint32_t w = byte * 0x01010101u;
    const u128 q = {w, w, w, w};
    const int units = round16(nbytes) >> 4;          // regions are padded to 16
    for (int u0 = 0; u0 < units; u0 += 64) {
        FOR_LANES(l) { if (u0 + l < units) st128a(lds + off + 16 * (u0 + l), q); }
    }
}

// global -> global, used for memcpyed chunks; 256 threads = 4 waves, wave w takes every 4th KiB
CIMG_DEV void wave_copy_g2g(const uint8_t* src, uint8_t* dst, int nbytes, int wave, int nwaves)
{
    const int units = nbytes >> 4;
    for (int u0 = wave * 64; u0 < units; u0 += nwaves * 64) {
        LV<u128> t;
        FOR_LANES(l) { if (u0 + l < units) t[l] = ld128u(src + 16 * (u0 + l)); }
        FOR_LANES(l) { if (u0 + l < units) st128u(dst + 16 * (u0 + l), t[l]); }
    }
    if (wave == 0) {
        const int done = units << 4;
        FOR_LANES(l) { if (done + l < nbytes) dst[done + l] = src[done + l]; }
    }
}

CIMG_DEV void wave_fill_global(uint8_t* dst, int nbytes, uint32_t byte, int wave, int nwaves)
{
    const uint32_t w = byte * 0x01010101u;
    const u128 q = {w, w, w, w};
    const int units = nbytes >> 4;
    for (int u0 = wave * 64; u0 < units; u0 += nwaves * 64) {
        FOR_LANES(l) { if (u0 + l < units) st128u(dst + 16 * (u0 + l), q); }
    }
    if (wave == 0) {
        const int done = units << 4;
        FOR_LANES(l) { if (done + l < nbytes) dst[done + l] = (uint8_t)byte; }
    }
}

// ---- LZ4 block decode by one wave, in place inside LDS ----------------------------------------------
// Compressed bytes occupy [cs, cs + csize); output is written to [base, base + n).  Every LDS index is
// clamped to lds_limit so a corrupt stream can produce garbage but never an out-of-range access.
CIMG_DEV int lz4_decode_wave(uint8_t* lds, int base, int n, int cs, int csize, int lds_limit)
{
    int ip = cs;
    const int iend = cs + csize;
    int op = base;
    const int oend = base + n;
    int wbase = -4096;
    LV<uint32_t> wb;
    const int clampmax = lds_limit - 1;

#define CIMG_PEEK(dst, at)                                                          \
    do {                                                                            \
        const int at_ = (at);                                                       \
        if (at_ - wbase >= 64) {                                                    \
            wbase = at_;                                                            \
            FOR_LANES(l) { wb[l] = lds[imin(at_ + l, clampmax)]; }                  \
        }                                                                           \
        dst = readlane(wb, at_ - wbase);                                            \
    } while (0)

    for (;;) {
        if (ip >= iend) return ERR_DATA;
        uint32_t token;
        CIMG_PEEK(token, ip);
        ip++;
        int lit = (int)(token >> 4);
        if (lit == 15) {
            uint32_t b;
            do {
                if (ip >= iend) return ERR_DATA;
                CIMG_PEEK(b, ip);
                ip++;
                lit += (int)b;
            } while (b == 255);
        }
        if (lit > iend - ip || lit > oend - op) return ERR_DATA;
        if (lit > 0) {
            const int rel = ip - wbase;
            if (rel >= 0 && rel + lit <= 64) {
                FOR_LANES_W(l) { if (l >= rel && l < rel + lit) lds[op + (l - rel)] = (uint8_t)wb[l]; }
            } else {
                for (int c = 0; c < lit; c += 64) {
                    LV<uint32_t> t;
                    FOR_LANES(l) { if (c + l < lit) t[l] = lds[ip + c + l]; }
                    FOR_LANES_W(l) { if (c + l < lit) lds[op + c + l] = (uint8_t)t[l]; }
                }
            }
            ip += lit;
            op += lit;
        }
        if (ip == iend) break;                              // a block ends with literals
        if (iend - ip < 2) return ERR_DATA;
        uint32_t o0, o1;
        CIMG_PEEK(o0, ip);
        CIMG_PEEK(o1, ip + 1);
        ip += 2;
        const int offset = (int)(o0 | (o1 << 8));
        if (offset == 0 || offset > op - base) return ERR_DATA;
        int ml = (int)(token & 15);
        if (ml == 15) {
            uint32_t b;
            do {
                if (ip >= iend) return ERR_DATA;
                CIMG_PEEK(b, ip);
                ip++;
                ml += (int)b;
            } while (b == 255);
        }
        ml += 4;
        if (ml > oend - op) return ERR_DATA;
        const int src = op - offset;
        if (offset >= 64) {
            for (int c = 0; c < ml; c += 64) {
                LV<uint32_t> t;
                FOR_LANES(l) { if (c + l < ml) t[l] = lds[src + c + l]; }
                FOR_LANES_W(l) { if (c + l < ml) lds[op + c + l] = (uint8_t)t[l]; }
            }
        } else {
            // overlapping match: byte t of the match equals pattern byte t mod offset
            const int period = offset * ((63 + offset) / offset);     // smallest multiple of offset >= 64
            for (int c = 0; c < ml; c += 64) {
                LV<uint32_t> t;
                FOR_LANES(l) {
                    if (c + l < ml) t[l] = (c == 0) ? lds[src + (l % offset)] : lds[op + c + l - period];
                }
                FOR_LANES_W(l) { if (c + l < ml) lds[op + c + l] = (uint8_t)t[l]; }
            }
        }
        op += ml;
    }
#undef CIMG_PEEK
    return op == oend ? 0 : ERR_DATA;
}

// ---- byte-plane helpers ----------------------------------------------------------------------------
#ifdef CIMG_EMULATE
inline uint32_t byte_perm(uint32_t hi, uint32_t lo, uint32_t sel)
{
    const uint64_t both = ((uint64_t)hi << 32) | lo;
    uint32_t r = 0;
    for (int i = 0; i < 4; i++) r |= (uint32_t)((both >> (8 * ((sel >> (8 * i)) & 7))) & 0xFF) << (8 * i);
    return r;
}
#else
CIMG_DEV uint32_t byte_perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
#endif

// ---- the kernel body ---------------------------------------------------------------------------------
struct DecodeBlock {
    const DecodeArgs& a;
    uint8_t* lds;
    int b;                 // batch-wide block index

    // results of the uniform header walk
    int chunk, j, bsize, ns, neblock, rs, ts, filter, mode;   // mode: 0 regular, 1 memcpyed, 2 zero, 3 skip
    const uint8_t* c;      // chunk base
    uint8_t* out;          // block output

    CIMG_DEV DecodeBlock(const DecodeArgs& a_, uint8_t* lds_, int b_) : a(a_), lds(lds_), b(b_) {}

    CIMG_DEV void fail(int code) { a.status[chunk] = code; }

    CIMG_DEV void phase_a(int wave)
    {
        chunk = find_chunk(a.descs, a.nchunks, b);
        const ChunkDesc& d = a.descs[chunk];
        j = b - d.blk0;
        c = a.comp + d.comp_off;
        out = a.raw + d.raw_off + (int64_t)j * d.blocksize;
        bsize = (j == d.nblocks - 1 && d.leftover) ? d.leftover : d.blocksize;
        mode = 3;
        const int flags = c[OFF_FLAGS];
        ts = c[OFF_TYPESIZE];
        const int nbytes = ld32s(c + OFF_NBYTES), blocksize = ld32s(c + OFF_BLOCKSIZE), cbytes = ld32s(c + OFF_CBYTES);
        if (c[0] > 5) { fail(ERR_VERSION_SUPPORT); return; }
        if (nbytes != d.nbytes || blocksize != d.blocksize || ts == 0 || cbytes < HEADER_LEN) { fail(ERR_INVALID_HEADER); return; }
        if ((flags & (FLAG_SHUFFLE | FLAG_BITSHUFFLE)) != (FLAG_SHUFFLE | FLAG_BITSHUFFLE)) { fail(ERR_VERSION_SUPPORT); return; }
        const int special = (c[OFF_BLOSC2_FLAGS] >> 4) & 7;
        if (special == SPECIAL_ZERO) { mode = 2; wave_fill_global(out, bsize, 0, wave, 4); return; }
        if (special != 0) { fail(ERR_DATA); return; }
        if (flags & FLAG_MEMCPYED) {
            if (cbytes != nbytes + HEADER_LEN) { fail(ERR_DATA); return; }
            mode = 1;
            wave_copy_g2g(c + HEADER_LEN + (int64_t)j * blocksize, out, bsize, wave, 4);
            return;
        }
        if ((flags >> 5) != 1) { fail(ERR_CODEC_SUPPORT); return; }
        // filter pipeline: exactly one of {none, shuffle, bitshuffle}, in the last slot
        filter = c[OFF_FILTERS + 5];
        for (int i = 0; i < 5; i++) if (c[OFF_FILTERS + i] != 0) { fail(ERR_CODEC_SUPPORT); return; }
        if (filter != FILTER_NONE && filter != FILTER_SHUFFLE) { fail(ERR_CODEC_SUPPORT); return; }
        const bool leftover_blk = bsize != blocksize;
        ns = (!(flags & FLAG_DONT_SPLIT) && !leftover_blk) ? ts : 1;
        neblock = bsize / ns;
        rs = region_stride(neblock);
        if (ns * rs + 16 > a.lds_bytes) { fail(ERR_FAILURE); return; }
        const int bstart = ld32s(c + HEADER_LEN + 4 * j);
        if (bstart < HEADER_LEN + 4 * d.nblocks || bstart > cbytes) { fail(ERR_DATA); return; }
        mode = 0;
        // walk the stream table; wave w stages streams w, w+4, ...
        int pos = bstart;
        for (int s = 0; s < ns; s++) {
            if (cbytes - pos < 4) { fail(ERR_READ_BUFFER); mode = 3; return; }
            const int cs = ld32s(c + pos);
            pos += 4;
            const int payload = cs > 0 ? cs : (cs < 0 ? 1 : 0);
            if (payload > cbytes - pos) { fail(ERR_READ_BUFFER); mode = 3; return; }
            if ((s & 3) == wave) {
                const int base = s * rs;
                if (cs == 0) {
                    wave_fill_lds(lds, base, neblock, 0);
                } else if (cs < 0) {
                    const int token = c[pos];
                    if (!(token & 1) || cs < -255) { fail(ERR_RUN_LENGTH); }
                    wave_fill_lds(lds, base, neblock, (uint32_t)(-cs) & 0xFF);
                } else if (cs == neblock) {
                    wave_copy_g2l(c + pos, lds, base, neblock);
                } else if (cs > neblock) {
                    fail(ERR_DATA);
                } else {
                    const int park = base + rs - round16(cs);
                    wave_copy_g2l(c + pos, lds, park, cs);
                    const int rc = lz4_decode_wave(lds, base, neblock, park, cs, a.lds_bytes);
                    if (rc < 0) fail(rc);
                }
            }
            pos += payload;
        }
    }

    // LDS offset of filtered byte k of the block (k in shuffled order)
    CIMG_DEV int plane_base(int p) const { return ns > 1 ? p * rs : p * (bsize / ts); }

    CIMG_DEV void phase_b(int wave)
    {
        if (mode != 0) return;
        const int tid0 = wave * 64;
        const int units = bsize >> 4;
        if (filter == FILTER_SHUFFLE && ts == 2 && !(bsize & 1)) {
            const int p0 = plane_base(0), p1 = plane_base(1);
            for (int u0 = tid0; u0 < units; u0 += 256) {
                FOR_LANES(l) {
                    const int u = u0 + l;
                    if (u < units) {
                        const uint32_t a0 = *reinterpret_cast<const uint32_t*>(lds + p0 + 8 * u);
                        const uint32_t a1 = *reinterpret_cast<const uint32_t*>(lds + p0 + 8 * u + 4);
                        const uint32_t b0 = *reinterpret_cast<const uint32_t*>(lds + p1 + 8 * u);
                        const uint32_t b1 = *reinterpret_cast<const uint32_t*>(lds + p1 + 8 * u + 4);
                        u128 o;
                        o.x = byte_perm(b0, a0, 0x05010400u);
                        o.y = byte_perm(b0, a0, 0x07030602u);
                        o.z = byte_perm(b1, a1, 0x05010400u);
                        o.w = byte_perm(b1, a1, 0x07030602u);
                        st128u(out + 16 * u, o);
                    }
                }
            }
        } else if (filter == FILTER_SHUFFLE && ts == 4 && !(bsize & 3)) {
            const int p0 = plane_base(0), p1 = plane_base(1), p2 = plane_base(2), p3 = plane_base(3);
            for (int u0 = tid0; u0 < units; u0 += 256) {
                FOR_LANES(l) {
                    const int u = u0 + l;
                    if (u < units) {
                        const uint32_t A = *reinterpret_cast<const uint32_t*>(lds + p0 + 4 * u);
                        const uint32_t B = *reinterpret_cast<const uint32_t*>(lds + p1 + 4 * u);
                        const uint32_t C = *reinterpret_cast<const uint32_t*>(lds + p2 + 4 * u);
                        const uint32_t D = *reinterpret_cast<const uint32_t*>(lds + p3 + 4 * u);
                        const uint32_t t0 = byte_perm(B, A, 0x05010400u), t1 = byte_perm(B, A, 0x07030602u);
                        const uint32_t v0 = byte_perm(D, C, 0x05010400u), v1 = byte_perm(D, C, 0x07030602u);
                        u128 o;
                        o.x = byte_perm(v0, t0, 0x05040100u);
                        o.y = byte_perm(v0, t0, 0x07060302u);
                        o.z = byte_perm(v1, t1, 0x05040100u);
                        o.w = byte_perm(v1, t1, 0x07060302u);
                        st128u(out + 16 * u, o);
                    }
                }
            }
        } else if (filter == FILTER_NONE || ts == 1) {
            // planes are consecutive slices of the block
            for (int u0 = tid0; u0 < units; u0 += 256) {
                FOR_LANES(l) {
                    const int u = u0 + l;
                    if (u < units) {
                        const int k = 16 * u;
                        if (ns > 1 && ((neblock & 15) != 0)) {
                            for (int i = 0; i < 16; i++) out[k + i] = lds[((k + i) / neblock) * rs + (k + i) % neblock];
                        } else {
                            const int off = ns > 1 ? (k / neblock) * rs + k % neblock : k;
                            st128u(out + k, ld128a(lds + off));
                        }
                    }
                }
            }
        } else {
            // generic typesize: byte gather
            const int ne = bsize / ts;
            for (int u0 = tid0; u0 < units; u0 += 256) {
                FOR_LANES(l) {
                    const int u = u0 + l;
                    if (u < units) {
                        for (int i = 0; i < 16; i++) {
                            const int k = 16 * u + i;
                            out[k] = (k < ne * ts) ? lds[plane_base(k % ts) + k / ts] : lds[k];
                        }
                    }
                }
            }
        }
        // tail: bsize % 16 bytes (and, for odd sizes, the verbatim bytes after ne*ts)
        if (wave == 0) {
            const int done = units << 4;
            const int ne = bsize / ts;
            FOR_LANES(l) {
                const int k = done + l;
                if (k < bsize) {
                    if (filter == FILTER_SHUFFLE && ts > 1)
                        out[k] = (k < ne * ts) ? lds[plane_base(k % ts) + k / ts] : lds[k];
                    else
                        out[k] = lds[ns > 1 ? (k / neblock) * rs + k % neblock : k];
                }
            }
        }
    }
};

}  // namespace cimg
