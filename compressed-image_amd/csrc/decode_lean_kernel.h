// decode_lean_kernel.h -- the decode kernel for the blocks that dominate image data: byte-shuffled, split into
// 2 or 4 byte planes, of which AT MOST ONE is coded (LZ4 or BloscLZ) (the others are stored raw -- noisy low mantissa bytes --
// or are run tokens).  Such a block needs LDS only for the one coded plane, so twice as many blocks are resident
// per CU as in cimg_decode_blocks (which keeps the whole block in LDS), and the serial LZ4 chains -- the thing
// decode time is made of -- run two per SIMD instead of one.  Raw planes never enter LDS: the un-shuffle merges
// the LDS plane with 8- or 4-byte global loads of the stored planes.
//
// Round 3: the launch is PERSISTENT and SOFTWARE-PIPELINED.  One wave owns an LDS plane and walks blocks b, b + G, b + 2G, ...
// (G = waves of the launch = what is resident at once).  A block used to be header walk (three dependent HBM reads, 4 us) ->
// coded bytes staged (4 us) -> chain (19 us) -> un-shuffle (3.5 us), everything in turn; now the loads of the NEXT blocks are
// in flight while the current chain runs:
//     stage A (block i + 2): chunk descriptor, then the 32-byte header and bstarts[j]
//     stage B (block i + 1): the size words and -- for the usual shape, low plane stored / high plane coded, whose positions
//                            follow from bstart alone -- the first 4 KiB of coded bytes, into registers
//     block i:               coded bytes registers -> LDS, stored plane requested into registers (64 VGPRs), chain, un-shuffle
// so that what a block costs its wave is the chain and the un-shuffle, not the walk.
//
// The kernel never reports an error and never guesses: any block outside its case (other filters / typesizes,
// unsplit or leftover blocks, two coded planes, anything unusual or damaged) is left alone -- done[b] is not
// set -- and cimg_decode_blocks, launched right behind it with the same done[] array, handles (and diagnoses) it.
#pragma once
#include "decode_kernel.h"

namespace cimg {

#ifdef CIMG_PROFILE
#define LEAN_STAMP(dbg, b, k) ((void)0)
#else
#define LEAN_STAMP(dbg, b, k) debug_stamp(dbg, b, k)
#endif

// what stage A knows about a block (wave-uniform) + its loads in flight
struct LeanHead {
    int b = -1;                        // batch-wide block index; -1: no such block
    int j = 0, nblocks = 0, nbytes = 0, blocksize = 0, held = 0;   // from the chunk descriptor
    bool leftover_blk = false;
    const uint8_t* c = nullptr;        // chunk base
    uint8_t* out = nullptr;            // the block's pixels
    LV<u128> h0, h1;                   // in flight: the chunk header
    LV<int32_t> bst;                   // in flight: bstarts[j]
    // after finish_head():
    int ok = 0, ts = 0, fmt = 0, cbytes = 0, bstart = 0, neblock = 0, rs = 0;
};

// stage B: the loads that follow from bstart alone (the usual block: plane 0 stored, plane 1 coded)
struct LeanBody {
    bool spec = false;
    LV<int32_t> cs0, cs1;              // size words at bstart and bstart + 4 + neblock
    LV<u128> st[4];                    // the first 4 KiB behind the second size word
};

struct DecodeLeanWave {
    const DecodeArgs& a;
    uint8_t* lds;

    // per-block results of the walk (wave-uniform)
    int ts = 0, bsize = 0, neblock = 0, ok = 0;
    int kind[4] = {0, 0, 0, 0};          // per plane: 0 = in LDS (decoded), 1 = stored raw at c + at[p], 2 = constant byte at[p]
    int at[4] = {0, 0, 0, 0};
    int lz_pos = 0, lz_cs = 0, lz_plane = -1;
    const uint8_t* c = nullptr;
    uint8_t* out = nullptr;
    // the stored plane of a two-plane block, requested from HBM BEFORE the serial LZ4 decode of the other plane and kept in
    // registers until the un-shuffle (16 KiB = 16 x 16 bytes per lane): its latency hides behind the decode
    LV<u128> pre[16];
    int prefetched = 0;
    int left = 0;                        // blocks this wave left to the general kernel (leave())

    CIMG_DEV DecodeLeanWave(const DecodeArgs& a_, uint8_t* lds_) : a(a_), lds(lds_) {}

    // ---- stage A: descriptor (every wave reads the same few descriptors: cache hits), header + bstarts[j] requested ------
    CIMG_DEV void issue_head(LeanHead& h, int b) const
    {
        h.b = -1; h.ok = 0;
        if (b >= a.total_blocks) return;
        const int chunk = find_chunk(a.descs, a.nchunks, b, a.uniform_nblocks);
        const ChunkDesc d = uniform_desc(a.descs + chunk);
        h.b = b;
        h.j = b - d.blk0;
        h.nblocks = d.nblocks; h.nbytes = d.nbytes; h.blocksize = d.blocksize; h.held = d.destsize;
        h.leftover_blk = h.j == d.nblocks - 1 && d.leftover != 0;
        h.c = a.comp + d.comp_off;
        h.out = a.raw + d.raw_off + (int64_t)h.j * d.blocksize;
        const bool table_held = (int64_t)HEADER_LEN + 4 * (int64_t)h.j + 4 <= (int64_t)h.held;
        FOR_LANES(l) {
            h.h0[l] = ld128u(h.c);
            h.h1[l] = ld128u(h.c + 16);
            // bstarts[j]: only read where the caller's buffer holds it (believed only after finish_head's checks)
            h.bst[l] = table_held ? ld32s(h.c + HEADER_LEN + 4 * h.j) : 0;
        }
    }

    // the header checks of the lean case; everything that fails leaves the block to the general kernel
    CIMG_DEV void finish_head(LeanHead& h) const
    {
        h.ok = 0;
        if (h.b < 0 || h.leftover_blk) return;                                   // leftover block: one unsplit stream
        const uint32_t w0 = uni(h.h0[0].x);
        const int flags = (int)((w0 >> 16) & 0xFF);
        h.ts = (int)(w0 >> 24);
        const int nbytes = (int)uni(h.h0[0].y), blocksize = (int)uni(h.h0[0].z);
        h.cbytes = (int)uni(h.h0[0].w);
        const uint32_t f0 = uni(h.h1[0].x), f1 = uni(h.h1[0].y), b2 = uni(h.h1[0].w);
        if ((w0 & 0xFF) > 5 || nbytes != h.nbytes || blocksize != h.blocksize || h.cbytes < HEADER_LEN || h.cbytes > h.held) return;
        if ((flags & (FLAG_SHUFFLE | FLAG_BITSHUFFLE)) != (FLAG_SHUFFLE | FLAG_BITSHUFFLE)) return;
        h.fmt = flags >> 5;                                                       // 0 blosclz, 1 lz4 / lz4hc
        if (((b2 >> 28) & 7) != 0 || (flags & (FLAG_MEMCPYED | FLAG_DONT_SPLIT)) || (h.fmt != 0 && h.fmt != 1)) return;
        if (f0 != 0 || (f1 & 0xFF) != 0 || (int)((f1 >> 8) & 0xFF) != FILTER_SHUFFLE) return;
        const int bsz = h.blocksize;
        if ((h.ts != 2 && h.ts != 4) || (bsz & 15) || bsz % h.ts) return;
        h.neblock = bsz / h.ts;
        h.rs = h.fmt == 0 ? blz_region_stride(h.neblock) : region_stride(h.neblock);
        if (h.rs + 16 > a.lds_bytes) return;
        if (h.cbytes < HEADER_LEN + 4 * h.nblocks) return;
        h.bstart = (int)uni((uint32_t)h.bst[0]);
        if (h.bstart < HEADER_LEN + 4 * h.nblocks || h.bstart > h.cbytes) return;
        h.ok = 1;
    }

    // ---- stage B: the usual image block stores its low byte plane raw and codes the high one.  Then the stored plane starts
    // at bstart + 4, the second size word sits at bstart + 4 + neblock and the coded bytes behind it: requested NOW, one
    // block ahead, instead of two dependent HBM round trips in front of the chain.  Everything requested lies inside the
    // chunk; a guess that does not hold costs the loads.
    CIMG_DEV void issue_body(LeanBody& q, const LeanHead& h) const
    {
        const int spec_lz = h.bstart + 8 + h.neblock;
        q.spec = h.ok && h.ts == 2 && h.neblock == 16384 && h.cbytes - spec_lz >= 4096 + 16;
        if (!q.spec) return;
        FOR_LANES(l) {
            q.cs0[l] = ld32s(h.c + h.bstart);
            q.cs1[l] = ld32s(h.c + spec_lz - 4);
        }
        CIMG_UNROLL
        for (int k = 0; k < 4; k++) { FOR_LANES(l) { q.st[k][l] = ld128u(h.c + spec_lz + 16 * (64 * k + l)); } }
    }

    // ---- block i: finish the walk, bring the coded bytes into LDS, request the stored plane -----------------------------------
    CIMG_DEV void begin_block(const LeanHead& h, const LeanBody& q)
    {
        ok = 0; prefetched = 0; lz_plane = -1; lz_cs = 0; lz_pos = 0;
        CIMG_UNROLL
        for (int s = 0; s < 4; s++) { kind[s] = 0; at[s] = 0; }
        if (!h.ok) return;
        c = h.c; out = h.out; ts = h.ts; bsize = h.blocksize; neblock = h.neblock;
        const int cbytes = h.cbytes, rs = h.rs;
        int coded = 0;
        bool walked = false;
        if (q.spec) {
            // the guess: plane 0 stored raw, plane 1 coded
            const int cs0 = (int)uni((uint32_t)q.cs0[0]), cs1 = (int)uni((uint32_t)q.cs1[0]);
            const int pos1 = h.bstart + 8 + neblock;
            if (cs0 == neblock && cs1 > 0 && cs1 < neblock && cs1 <= cbytes - pos1) {
                kind[0] = 1; at[0] = h.bstart + 4;
                kind[1] = 0; lz_plane = 1; lz_pos = pos1; lz_cs = cs1; coded = 1;
                walked = true;
            }
        }
        if (!walked) {
            int pos = h.bstart;
            CIMG_UNROLL
            for (int s = 0; s < 4; s++) {                                         // fixed trip count: kind[] / at[] stay in registers
                if (s >= ts) continue;
                if (cbytes - pos < 4) return;
                const int cs = ld32s(c + pos);
                pos += 4;
                const int payload = cs > 0 ? cs : (cs < 0 ? 1 : 0);
                if (payload > cbytes - pos) return;
                if (cs == 0) { kind[s] = 2; at[s] = 0; }
                else if (cs < 0) {
                    if (cs < -255 || !(c[pos] & 1)) return;
                    kind[s] = 2; at[s] = (-cs) & 0xFF;
                }
                else if (cs == neblock) { kind[s] = 1; at[s] = pos; }
                else if (cs > neblock) return;
                else { kind[s] = 0; lz_plane = s; lz_pos = pos; lz_cs = cs; coded++; }
                pos += payload;
            }
            if (coded > 1) return;
        }
        ok = 1;
        // the stored plane travels through the decode in registers (static indices only: a dynamically indexed member array
        // would push the whole object, pre[] included, to scratch)
        const bool raw0 = kind[0] == 1, raw1 = kind[1] == 1;
        const int raw_at = raw0 ? at[0] : at[1];
        // (only next to a CODED plane: the un-shuffle of the register form takes the other plane from LDS.  A block whose other
        // plane is a run token -- 16-bit pixels below 256 with a noisy low byte -- goes the plane_word way.)
        const bool want_pre = coded == 1 && ts == 2 && neblock == 16384 && (raw0 != raw1);
        if (coded == 1) {
            const int park = rs - round16(lz_cs);
            LEAN_STAMP(a.dbg, h.b, 1);                                            // header walk done
            if (walked && lz_cs <= 4096) {
                // the guess held and the stream is short: its bytes are here already; whole 16-byte units go to LDS, the last
                // one with up to 15 bytes the decoder never looks at (the parking area is round16(lz_cs) long)
                const int units16 = (lz_cs + 15) >> 4;
                CIMG_UNROLL
                for (int k = 0; k < 4; k++) { FOR_LANES(l) { if (64 * k + l < units16) st128a(lds + park + 16 * (64 * k + l), q.st[k][l]); } }
            } else {
                wave_copy_g2l(c + lz_pos, lds, park, lz_cs);
            }
        }
        if (want_pre) {
            // eight bytes per lane and load: piece j = stored-plane bytes [512 j + 8 l, + 8), pieces 2 k and 2 k + 1 in pre[k] --
            // the un-shuffle then writes 16 CONTIGUOUS bytes per lane (a whole KiB per store instruction)
            CIMG_UNROLL
            for (int k = 0; k < 16; k++) {
                FOR_LANES(l) {
                    uint64_t qa, qb;
                    memcpy(&qa, c + raw_at + 1024 * k + 8 * l, 8);
                    memcpy(&qb, c + raw_at + 1024 * k + 512 + 8 * l, 8);
                    pre[k][l].x = (uint32_t)qa; pre[k][l].y = (uint32_t)(qa >> 32); pre[k][l].z = (uint32_t)qb; pre[k][l].w = (uint32_t)(qb >> 32);
                }
            }
            prefetched = 2;
        }
        LEAN_STAMP(a.dbg, h.b, 2);                                                // coded bytes staged
    }

    // the serial part: the coded plane decoded in place in LDS.  false: damaged stream (the general kernel reports it)
    CIMG_DEV bool chain(const LeanHead& h)
    {
        if (lz_plane < 0) return true;
        const int park = h.rs - round16(lz_cs);
#ifdef CIMG_PROFILE
        // diagnostic builds: the 16 uint64 of the block carry the LZ4 decoder's cycle laps and counts instead of phase stamps
        const int rc = h.fmt == 0 ? blosclz_decode_wave(lds, 0, neblock, park, lz_cs, a.lds_bytes)
                                  : lz4_decode_wave(lds, 0, neblock, park, lz_cs, a.lds_bytes, a.dbg, h.b);
#else
        const int rc = h.fmt == 0 ? blosclz_decode_wave(lds, 0, neblock, park, lz_cs, a.lds_bytes)
                                  : CIMG_LZ4_DECODE(lds, 0, neblock, park, lz_cs, a.lds_bytes);
#endif
        return rc >= 0;
    }

    // 8 (ts = 2) or 4 (ts = 4) consecutive bytes of plane p, starting at plane offset `off`
    CIMG_DEV uint32_t plane_word(int p, int off) const
    {
        if (kind[p] == 0) return *reinterpret_cast<const uint32_t*>(lds + off);
        if (kind[p] == 1) return ld32u(c + at[p] + off);
        return (uint32_t)at[p] * 0x01010101u;
    }

    // a block left to the general kernel: counted in a register; the wave reports its count ONCE, with a plain store into its own
    // word of page-locked host memory (run()).  (Round 2 counted with a system-scope atomic per block on one host address:
    // a PCIe round trip each, one after the other -- a batch the lean kernel cannot take at all, 4096 blocks, waited 4 ms for them.)
    CIMG_DEV void leave() { ++left; }

    CIMG_DEV void unshuffle(int b)
    {
        if (prefetched == 2) {
            // piece j of both planes -> output bytes [1024 j + 16 l, + 16)
            CIMG_UNROLL
            for (int k = 0; k < 16; k++) {
                FOR_LANES(l) {
                    CIMG_UNROLL
                    for (int half = 0; half < 2; half++) {
                        const int off8 = 1024 * k + 512 * half + 8 * l;
                        const uint32_t x0 = *reinterpret_cast<const uint32_t*>(lds + off8), x1 = *reinterpret_cast<const uint32_t*>(lds + off8 + 4);
                        const uint32_t p0 = half ? pre[k][l].z : pre[k][l].x, p1 = half ? pre[k][l].w : pre[k][l].y;
                        const uint32_t lo0 = lz_plane == 0 ? x0 : p0, lo1 = lz_plane == 0 ? x1 : p1;      // plane 0 = low bytes
                        const uint32_t hi0 = lz_plane == 0 ? p0 : x0, hi1 = lz_plane == 0 ? p1 : x1;
                        u128 o;
                        o.x = byte_perm(hi0, lo0, 0x05010400u); o.y = byte_perm(hi0, lo0, 0x07030602u);
                        o.z = byte_perm(hi1, lo1, 0x05010400u); o.w = byte_perm(hi1, lo1, 0x07030602u);
#ifdef CIMG_ABL_NO_STORE      /* timing experiment only (tools/diag_dectime.py): the block's pixels are not written */
                        if (o.x == 0x12345678u && o.y == 0x9abcdef0u && o.z == 1u && o.w == 2u) st128u(out + 2 * off8, o);
#else
                        st128u(out + 2 * off8, o);
#endif
                    }
                }
            }
            FOR_LANES_W(l) { a.done[b] = a.gen; }
            return;
        }
        const int units = bsize >> 4;
        // The stored planes come from global memory: DEPTH units per lane are requested before the first is used
        // (one wave per block has nothing else to hide HBM latency with).
        constexpr int DEPTH = 8;
        if (ts == 2) {
            int u0 = 0;
            for (; u0 + (DEPTH - 1) * 64 + 64 <= units; u0 += DEPTH * 64) {
                LV<uint32_t> a0[DEPTH], a1[DEPTH], b0[DEPTH], b1[DEPTH];
                CIMG_UNROLL
                for (int k = 0; k < DEPTH; k++) {
                    FOR_LANES(l) {
                        const int u = u0 + k * 64 + l;
                        a0[k][l] = plane_word(0, 8 * u); a1[k][l] = plane_word(0, 8 * u + 4);
                        b0[k][l] = plane_word(1, 8 * u); b1[k][l] = plane_word(1, 8 * u + 4);
                    }
                }
                CIMG_UNROLL
                for (int k = 0; k < DEPTH; k++) {
                    FOR_LANES(l) {
                        const int u = u0 + k * 64 + l;
                        u128 o;
                        o.x = byte_perm(b0[k][l], a0[k][l], 0x05010400u);
                        o.y = byte_perm(b0[k][l], a0[k][l], 0x07030602u);
                        o.z = byte_perm(b1[k][l], a1[k][l], 0x05010400u);
                        o.w = byte_perm(b1[k][l], a1[k][l], 0x07030602u);
                        st128u(out + 16 * u, o);
                    }
                }
            }
            for (; u0 < units; u0 += 64) {
                FOR_LANES(l) {
                    const int u = u0 + l;
                    if (u < units) {
                        const uint32_t a0 = plane_word(0, 8 * u), a1 = plane_word(0, 8 * u + 4);
                        const uint32_t b0 = plane_word(1, 8 * u), b1 = plane_word(1, 8 * u + 4);
                        u128 o;
                        o.x = byte_perm(b0, a0, 0x05010400u);
                        o.y = byte_perm(b0, a0, 0x07030602u);
                        o.z = byte_perm(b1, a1, 0x05010400u);
                        o.w = byte_perm(b1, a1, 0x07030602u);
                        st128u(out + 16 * u, o);
                    }
                }
            }
        } else {
            int u0 = 0;
            for (; u0 + (DEPTH - 1) * 64 + 64 <= units; u0 += DEPTH * 64) {
                LV<uint32_t> A[DEPTH], B[DEPTH], C[DEPTH], D[DEPTH];
                CIMG_UNROLL
                for (int k = 0; k < DEPTH; k++) {
                    FOR_LANES(l) {
                        const int u = u0 + k * 64 + l;
                        A[k][l] = plane_word(0, 4 * u); B[k][l] = plane_word(1, 4 * u); C[k][l] = plane_word(2, 4 * u); D[k][l] = plane_word(3, 4 * u);
                    }
                }
                CIMG_UNROLL
                for (int k = 0; k < DEPTH; k++) {
                    FOR_LANES(l) {
                        const int u = u0 + k * 64 + l;
                        const uint32_t t0 = byte_perm(B[k][l], A[k][l], 0x05010400u), t1 = byte_perm(B[k][l], A[k][l], 0x07030602u);
                        const uint32_t v0 = byte_perm(D[k][l], C[k][l], 0x05010400u), v1 = byte_perm(D[k][l], C[k][l], 0x07030602u);
                        u128 o;
                        o.x = byte_perm(v0, t0, 0x05040100u);
                        o.y = byte_perm(v0, t0, 0x07060302u);
                        o.z = byte_perm(v1, t1, 0x05040100u);
                        o.w = byte_perm(v1, t1, 0x07060302u);
                        st128u(out + 16 * u, o);
                    }
                }
            }
            for (; u0 < units; u0 += 64) {
                FOR_LANES(l) {
                    const int u = u0 + l;
                    if (u < units) {
                        const uint32_t A = plane_word(0, 4 * u), B = plane_word(1, 4 * u), C = plane_word(2, 4 * u), D = plane_word(3, 4 * u);
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
        }
        FOR_LANES_W(l) { a.done[b] = a.gen; }
    }

    // ---- the persistent loop: wave w of G walks blocks w, w + G, w + 2G, ... ---------------------------------------------------
    CIMG_DEV void run(int w, int G)
    {
        LeanHead cur, nxt, far;
        LeanBody body, nbody;
        // (DecodeArgs::tune -- only where waves are persistent and start together: a launch of one workgroup per block has its
        // waves starting whenever a slot falls free)
        const int tune = G < a.total_blocks ? a.tune : 0;
        const int slot = tune ? wave_slot() : 0;
        if ((tune & 0xFF) && (tune & 0x10000)) wave_sleep64((tune & 0xFF) * (2 * wave_simd() + (slot & 1)));      // every wave of a CU its own delay
        else if ((tune & 0xFF) && (slot & 1)) wave_sleep64(tune & 0xFF);
        issue_head(cur, w);
        issue_head(nxt, w + G);
        finish_head(cur);
        issue_body(body, cur);
        // hard bound: a wave visits at most total_blocks / G + 1 blocks
        for (int b = w, guard = 0; b < a.total_blocks && guard <= a.total_blocks; b += G, ++guard) {
            LEAN_STAMP(a.dbg, b, 0);
            begin_block(cur, body);                     // block i: coded bytes -> LDS, stored plane requested
            finish_head(nxt);                           // block i + 1: its header arrived during the previous chain ...
            issue_body(nbody, nxt);                     // ... its size words and coded bytes are requested now
            issue_head(far, b + 2 * G);                 // block i + 2: descriptor, header, bstarts[j]
#ifdef CIMG_ABL_NO_CHAIN      /* timing experiment only: the coded plane is not decoded */
            if (ok) unshuffle(b);
#else
            if (ok && chain(cur)) unshuffle(b);
#endif
            else if (!(cur.leftover_blk && a.blk_first)) leave();      // (a.blk_first != 0: the leftover blocks are launched separately, nobody counts them)
            LEAN_STAMP(a.dbg, b, 3);
            if ((tune & 0x100) && (slot & 1) && guard == 0) wave_priority(tune >> 9 & 3);
            cur = nxt; body = nbody; nxt = far;
        }
        if (a.skipped && left) { const uint32_t v = (uint32_t)left; FOR_LANES_W(l) { if (l == 0) a.skipped[w] = v; } }
    }
};

}  // namespace cimg
