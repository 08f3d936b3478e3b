// decode_lean_kernel.h -- the decode kernel for the blocks that dominate image data: byte-shuffled, split into
// 2 or 4 byte planes, of which AT MOST ONE is coded (LZ4 or BloscLZ) (the others are stored raw -- noisy low mantissa bytes --
// or are run tokens).  Such a block needs LDS only for the one coded plane, so twice as many blocks are resident
// per CU as in cimg_decode_blocks (which keeps the whole block in LDS), and the serial LZ4 chains -- the thing
// decode time is made of -- run two per SIMD instead of one.  Raw planes never enter LDS: the un-shuffle merges
// the LDS plane with 8- or 4-byte global loads of the stored planes.
//
// The kernel never reports an error and never guesses: any block outside its case (other filters / typesizes,
// unsplit or leftover blocks, two coded planes, anything unusual or damaged) is left alone -- done[b] is not
// set -- and cimg_decode_blocks, launched right behind it with the same done[] array, handles (and diagnoses) it.
#pragma once
#include "decode_kernel.h"
#include "decode_pair.h"

namespace cimg {

struct DecodeLean {
    const DecodeArgs& a;
    uint8_t* lds;
    int b;
    // results of the uniform header walk (wave-uniform, identical in all four waves)
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
    // pair mode (cimg_decode_lean_pair: two waves per block): wave 0 finds the tokens of the LZ4 chain, wave 1 moves the
    // bytes (decode_pair.h); both carry half of the stored plane in registers
    int pair_mode = 0, pair_on = 0, mail = 0;
    Lz4PairProducer prod;

    CIMG_DEV DecodeLean(const DecodeArgs& a_, uint8_t* lds_, int b_) : a(a_), lds(lds_), b(b_) {}

    CIMG_DEV void phase_a(int wave, int nwaves)
    {
        const int chunk = find_chunk(a.descs, a.nchunks, b, a.uniform_nblocks);
        const ChunkDesc d = uniform_desc(a.descs + chunk);
        const int j = b - d.blk0;
        c = a.comp + d.comp_off;
        out = a.raw + d.raw_off + (int64_t)j * d.blocksize;
        bsize = d.blocksize;
        if (j == d.nblocks - 1 && d.leftover) return;                          // leftover block: one unsplit stream
        const u128 h0 = ld128u(c), h1 = ld128u(c + 16);
        const uint32_t w0 = uni(h0.x);
        const int flags = (int)((w0 >> 16) & 0xFF);
        ts = (int)(w0 >> 24);
        const int nbytes = (int)uni(h0.y), blocksize = (int)uni(h0.z), cbytes = (int)uni(h0.w);
        const uint32_t f0 = uni(h1.x), f1 = uni(h1.y), b2 = uni(h1.w);
        if ((w0 & 0xFF) > 5 || nbytes != d.nbytes || blocksize != d.blocksize || cbytes < HEADER_LEN || cbytes > d.destsize) return;
        if ((flags & (FLAG_SHUFFLE | FLAG_BITSHUFFLE)) != (FLAG_SHUFFLE | FLAG_BITSHUFFLE)) return;
        const int fmt = flags >> 5;                                           // 0 blosclz, 1 lz4 / lz4hc
        if (((b2 >> 28) & 7) != 0 || (flags & (FLAG_MEMCPYED | FLAG_DONT_SPLIT)) || (fmt != 0 && fmt != 1)) return;
        if (f0 != 0 || (f1 & 0xFF) != 0 || (int)((f1 >> 8) & 0xFF) != FILTER_SHUFFLE) return;
        if ((ts != 2 && ts != 4) || (bsize & 15) || bsize % ts) return;
        neblock = bsize / ts;
        const int rs = fmt == 0 ? blz_region_stride(neblock) : region_stride(neblock);
        if (rs + 16 > a.lds_bytes) return;
        if (cbytes < HEADER_LEN + 4 * d.nblocks) return;
        const int bstart = ld32s(c + HEADER_LEN + 4 * j);
        if (bstart < HEADER_LEN + 4 * d.nblocks || bstart > cbytes) return;
        int pos = bstart, coded = 0;
        // Speculation.  The usual image block stores its low byte plane raw and codes the high one: then the stored plane starts
        // at bstart + 4 and the coded bytes at bstart + 4 + neblock + 4, and both are requested NOW, together with the size
        // words, instead of two dependent HBM round trips later (the walk is five of them).  Everything requested lies inside
        // the chunk; a guess that does not hold costs the loads.
        const int spec_raw = bstart + 4, spec_lz = bstart + 8 + neblock;
        const bool spec = nwaves == 1 && !pair_mode && ts == 2 && neblock == 16384 && cbytes - spec_lz >= 4096 + 16;
        LV<u128> st[4];
        if (spec) {
            CIMG_UNROLL
            for (int k = 0; k < 4; k++) { FOR_LANES(l) { st[k][l] = ld128u(c + spec_lz + 16 * (64 * k + l)); } }
            CIMG_UNROLL
            for (int k = 0; k < 16; k++) {
                FOR_LANES(l) {
                    uint64_t qa, qb;
                    memcpy(&qa, c + spec_raw + 1024 * k + 8 * l, 8);
                    memcpy(&qb, c + spec_raw + 1024 * k + 512 + 8 * l, 8);
                    pre[k][l].x = (uint32_t)qa; pre[k][l].y = (uint32_t)(qa >> 32); pre[k][l].z = (uint32_t)qb; pre[k][l].w = (uint32_t)(qb >> 32);
                }
            }
        }
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
        ok = 1;
        // which wave runs the serial LZ4 chain (the launch uses ONE wave per block: 9 blocks per CU and the hardware
        // spreads them over the SIMDs; with 4 waves per block 3 of them only wait and the chains clump: 137 vs 125 us)
        const int lzwave = pair_mode ? 0 : (int)((((uint32_t)b * 2654435761u) >> 30) & (uint32_t)(nwaves - 1));   // nwaves is 1, 2 or 4
        // one wave per block (the launch shape): that wave can carry the stored plane through the decode in registers
        // (static indices only: a dynamically indexed member array would push the whole object, pre[] included, to scratch)
        const bool raw0 = kind[0] == 1, raw1 = kind[1] == 1;
        const int raw_at = raw0 ? at[0] : at[1];
        const bool want_pre = nwaves == 1 && ts == 2 && neblock == 16384 && (raw0 != raw1);
        if (pair_mode) {
            // two waves per block.  An LZ4 plane is decoded by the pair; anything else by wave 0 alone, as below.
            mail = (a.lds_bytes - 16 - PAIR_MAIL_BYTES) & ~15;
            pair_on = coded == 1 && fmt == 1 && rs <= mail;
            const bool pre2 = ts == 2 && neblock == 16384 && (raw0 != raw1);
            const int park = rs - round16(lz_cs);
            if (pair_on && wave == 0) {
                debug_stamp(a.dbg, b, 1);
                const int units = lz_cs >> 4;
                if (pre2 && lz_cs <= 4096) {
                    LV<u128> t[4];
                    CIMG_UNROLL
                    for (int k = 0; k < 4; k++) { FOR_LANES(l) { t[k][l] = ld128u(c + lz_pos + 16 * imin(64 * k + l, imax(units - 1, 0))); } }
                    LV<uint32_t> tailb;
                    FOR_LANES(l) { tailb[l] = c[lz_pos + imin((units << 4) + l, lz_cs - 1)]; }
                    CIMG_UNROLL
                    for (int k = 0; k < 8; k++) { FOR_LANES(l) { pre[k][l] = ld128u(c + raw_at + 1024 * k + 16 * l); } }
                    prefetched = 1;
                    CIMG_UNROLL
                    for (int k = 0; k < 4; k++) { FOR_LANES(l) { if (64 * k + l < units) st128a(lds + park + 16 * (64 * k + l), t[k][l]); } }
                    FOR_LANES(l) { if ((units << 4) + l < lz_cs) lds[park + (units << 4) + l] = (uint8_t)tailb[l]; }
                } else {
                    wave_copy_g2l(c + lz_pos, lds, park, lz_cs);
                }
                debug_stamp(a.dbg, b, 2);
                prod.init(lds, 0, neblock, park, lz_cs, mail, mail);
            }
            if (pair_on && wave == 1 && pre2 && lz_cs <= 4096) {
                CIMG_UNROLL
                for (int k = 0; k < 8; k++) { FOR_LANES(l) { pre[k][l] = ld128u(c + raw_at + 1024 * (8 + k) + 16 * l); } }
                prefetched = 1;
            }
            if (pair_on) return;
        }
        if (coded == 1 && wave == lzwave) {
            const int park = rs - round16(lz_cs);
            debug_stamp(a.dbg, b, 1);                                             // header walk done
            if (want_pre && lz_cs <= 4096 && spec && raw_at == spec_raw && lz_pos == spec_lz) {
                // the guess held: coded bytes and stored plane are on their way (or here) already; whole 16-byte units go to LDS,
                // the last one with up to 15 bytes the decoder never looks at (the parking area is round16(lz_cs) long)
                const int units16 = (lz_cs + 15) >> 4;
                prefetched = 2;
                CIMG_UNROLL
                for (int k = 0; k < 4; k++) { FOR_LANES(l) { if (64 * k + l < units16) st128a(lds + park + 16 * (64 * k + l), st[k][l]); } }
            } else if (want_pre && lz_cs <= 4096) {
                // coded bytes first, stored plane behind them: the wait for the coded bytes leaves the 16 plane loads in flight
                const int units = lz_cs >> 4;
                LV<u128> t[4];
                CIMG_UNROLL
                for (int k = 0; k < 4; k++) { FOR_LANES(l) { t[k][l] = ld128u(c + lz_pos + 16 * imin(64 * k + l, imax(units - 1, 0))); } }
                LV<uint32_t> tailb;
                FOR_LANES(l) { tailb[l] = c[lz_pos + imin((units << 4) + l, lz_cs - 1)]; }
                CIMG_UNROLL
                // eight bytes per lane and load: piece j = stored-plane bytes [512 j + 8 l, + 8), pieces 2 k and 2 k + 1 in pre[k] --
                // the un-shuffle then writes 16 CONTIGUOUS bytes per lane (a whole KiB per store instruction)
                for (int k = 0; k < 16; k++) {
                    FOR_LANES(l) {
                        uint64_t a, b;
                        memcpy(&a, c + raw_at + 1024 * k + 8 * l, 8);
                        memcpy(&b, c + raw_at + 1024 * k + 512 + 8 * l, 8);
                        pre[k][l].x = (uint32_t)a; pre[k][l].y = (uint32_t)(a >> 32); pre[k][l].z = (uint32_t)b; pre[k][l].w = (uint32_t)(b >> 32);
                    }
                }
                prefetched = 2;
                CIMG_UNROLL
                for (int k = 0; k < 4; k++) { FOR_LANES(l) { if (64 * k + l < units) st128a(lds + park + 16 * (64 * k + l), t[k][l]); } }
                FOR_LANES(l) { if ((units << 4) + l < lz_cs) lds[park + (units << 4) + l] = (uint8_t)tailb[l]; }
            } else {
                wave_copy_g2l(c + lz_pos, lds, park, lz_cs);
            }
            debug_stamp(a.dbg, b, 2);                                             // coded bytes staged
            const int rc = fmt == 0 ? blosclz_decode_wave(lds, 0, neblock, park, lz_cs, a.lds_bytes)
                                    : lz4_decode_wave(lds, 0, neblock, park, lz_cs, a.lds_bytes);
            // the verdict travels to the other waves through the last LDS word of the allocation
            FOR_LANES_W(l) { *reinterpret_cast<int32_t*>(lds + a.lds_bytes - 4) = rc; }
        }
    }

    // pair mode, wave 0, after the step loop: the chain's verdict goes where the single-wave path leaves it
    CIMG_DEV void pair_finish()
    {
        const int rc = prod.finished ? prod.rc : ERR_FAILURE;        // (the step loop is bounded: it cannot leave a stream half done silently)
        FOR_LANES_W(l) { *reinterpret_cast<int32_t*>(lds + a.lds_bytes - 4) = rc; }
    }

    // 8 (ts = 2) or 4 (ts = 4) consecutive bytes of plane p, starting at plane offset `off`
    CIMG_DEV uint32_t plane_word(int p, int off) const
    {
        if (kind[p] == 0) return *reinterpret_cast<const uint32_t*>(lds + off);
        if (kind[p] == 1) return ld32u(c + at[p] + off);
        return (uint32_t)at[p] * 0x01010101u;
    }

    CIMG_DEV void leave(int wave) const
    {
        if (wave == 0 && a.skipped) { FOR_LANES(l) { if (l == 0) atomic_count(a.skipped); } }
    }

    CIMG_DEV void phase_b(int wave, int nwaves)
    {
        if (!ok) { leave(wave); return; }
        if (lz_plane >= 0) {
            LV<int32_t> rc;
            FOR_LANES(l) { rc[l] = *reinterpret_cast<const int32_t*>(lds + a.lds_bytes - 4); }
            if (readlane(rc, 0) < 0) { leave(wave); return; }                 // damaged stream: the general kernel reports it
        }
        if (prefetched) {
            // lane l, piece k: plane bytes [1024 k + 16 l, + 16) of both planes -> 32 contiguous pixels bytes
            // (pair mode: wave w holds pieces 8 w .. 8 w + 7 in pre[0 .. 7])
            if (prefetched == 2) {
                // one-wave launch: piece j of both planes -> output bytes [1024 j + 16 l, + 16)
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
                            st128u(out + 2 * off8, o);
                        }
                    }
                }
                if (wave == 0) { FOR_LANES_W(l) { a.done[b] = a.gen; } }
                return;
            }
            CIMG_UNROLL
            for (int k = 0; k < 16; k++) {
                if (pair_mode && k >= 8) break;
                FOR_LANES(l) {
                    const int off = 1024 * (pair_mode ? 8 * wave + k : k) + 16 * l;
                    const u128 x = ld128a(lds + off);
                    const u128 lo = lz_plane == 0 ? x : pre[k][l], hi = lz_plane == 0 ? pre[k][l] : x;    // plane 0 = low bytes
                    u128 o0, o1;
                    o0.x = byte_perm(hi.x, lo.x, 0x05010400u); o0.y = byte_perm(hi.x, lo.x, 0x07030602u);
                    o0.z = byte_perm(hi.y, lo.y, 0x05010400u); o0.w = byte_perm(hi.y, lo.y, 0x07030602u);
                    o1.x = byte_perm(hi.z, lo.z, 0x05010400u); o1.y = byte_perm(hi.z, lo.z, 0x07030602u);
                    o1.z = byte_perm(hi.w, lo.w, 0x05010400u); o1.w = byte_perm(hi.w, lo.w, 0x07030602u);
                    st128u(out + 2 * off, o0);
                    st128u(out + 2 * off + 16, o1);
                }
            }
            if (wave == 0) { FOR_LANES_W(l) { a.done[b] = a.gen; } }
            return;
        }
        const int units = bsize >> 4;
        const int tid0 = wave * 64, step = nwaves * 64;
        // The stored planes come from global memory: DEPTH units per lane are requested before the first is used
        // (one wave per block has nothing else to hide HBM latency with).
        constexpr int DEPTH = 8;
        if (ts == 2) {
            int u0 = tid0;
            for (; u0 + (DEPTH - 1) * step + 64 <= units; u0 += DEPTH * step) {
                LV<uint32_t> a0[DEPTH], a1[DEPTH], b0[DEPTH], b1[DEPTH];
                CIMG_UNROLL
                for (int k = 0; k < DEPTH; k++) {
                    FOR_LANES(l) {
                        const int u = u0 + k * step + l;
                        a0[k][l] = plane_word(0, 8 * u); a1[k][l] = plane_word(0, 8 * u + 4);
                        b0[k][l] = plane_word(1, 8 * u); b1[k][l] = plane_word(1, 8 * u + 4);
                    }
                }
                CIMG_UNROLL
                for (int k = 0; k < DEPTH; k++) {
                    FOR_LANES(l) {
                        const int u = u0 + k * step + l;
                        u128 o;
                        o.x = byte_perm(b0[k][l], a0[k][l], 0x05010400u);
                        o.y = byte_perm(b0[k][l], a0[k][l], 0x07030602u);
                        o.z = byte_perm(b1[k][l], a1[k][l], 0x05010400u);
                        o.w = byte_perm(b1[k][l], a1[k][l], 0x07030602u);
                        st128u(out + 16 * u, o);
                    }
                }
            }
            for (; u0 < units; u0 += step) {
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
            int u0 = tid0;
            for (; u0 + (DEPTH - 1) * step + 64 <= units; u0 += DEPTH * step) {
                LV<uint32_t> A[DEPTH], B[DEPTH], C[DEPTH], D[DEPTH];
                CIMG_UNROLL
                for (int k = 0; k < DEPTH; k++) {
                    FOR_LANES(l) {
                        const int u = u0 + k * step + l;
                        A[k][l] = plane_word(0, 4 * u); B[k][l] = plane_word(1, 4 * u); C[k][l] = plane_word(2, 4 * u); D[k][l] = plane_word(3, 4 * u);
                    }
                }
                CIMG_UNROLL
                for (int k = 0; k < DEPTH; k++) {
                    FOR_LANES(l) {
                        const int u = u0 + k * step + l;
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
            for (; u0 < units; u0 += step) {
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
        if (wave == 0) { FOR_LANES_W(l) { a.done[b] = a.gen; } }
    }
};

}  // namespace cimg
