// decode_pair.h -- the LZ4 block decode of decode_kernel.h as a TWO-WAVE pipeline (lean decode launch, pair mode).
//
// One wave walking an LZ4 chain spends a parse window's time half on finding the tokens (per-lane header parse, chain
// walk, prefix sum of the output positions, classification of the matches) and half on moving bytes (literal scatter,
// lane-parallel matches, the in-order rest) -- two dependent latency chains of about 2000 cycles each, back to back.
// They do not depend on each other ACROSS windows: finding the tokens of window i + 1 reads only compressed bytes,
// moving the bytes of window i writes only output that lies in front of them (in-place decode keeps op <= ip).
// So wave A (producer) parses window i while wave B (consumer) executes window i - 1, ONE barrier per step:
//
//      step i:   A: [a sequence the window parse could not take, found in step i - 1: the scalar path, run now -- B is
//                    idle this step and every earlier window is in place]
//                   parse window i  ->  mailbox slot i & 1          B: move the bytes of mailbox slot (i - 1) & 1
//      ------------------------------------------- barrier -------------------------------------------
//
// A mailbox slot is 1088 bytes of LDS behind the plane: sixteen words of header, two packed words per token lane, and
// the dense descriptors of the lane-parallel matches (A writes them straight to their dense slot: no ds_permute).
// Results are those of lz4_decode_wave bit for bit -- the same parse, the same independence rule, the same copies --
// also on damaged input (tests/emu runs both on the same streams).  Measured with the mailbox written between two
// barriers of a step (first form): no gain, the write sat on the critical path; hence two slots and one barrier.
#pragma once
#include "decode_kernel.h"

namespace cimg {

enum : int { PAIR_HDR_BYTES = 64, PAIR_SLOT_BYTES = 64 + 4 * 256, PAIR_MAIL_BYTES = 2 * PAIR_SLOT_BYTES, PAIR_BATCH = 0, PAIR_NONE = 1, PAIR_END = 2 };

CIMG_DEV int32_t mail_hdr(const uint8_t* lds, int mail, int k)
{
    LV<int32_t> t;
    FOR_LANES(l) { t[l] = *reinterpret_cast<const int32_t*>(lds + mail + 4 * k); }
    return readlane(t, 0);
}

struct Lz4PairProducer {
    uint8_t* lds = nullptr;
    int base = 0, iend = 0, oend = 0, clampmax = 0, mail = 0;
    int ip = 0, op = 0;
    bool dense_tokens = false, need_scalar = false, finished = false;
    int slot = 0;                       // mailbox slot (byte offset) compute() writes this step
    int kind = PAIR_NONE, rc = 0;
    // the window compute() found, waiting for post()
    uint64_t tokens = 0, parmask = 0;
    int P = 0, s = 0, biglast = 0, win_ip = 0, win_op = 0;
    LV<int> w0, w1, rank, d0, d1, d2, cnt;
    LV<bool> par;
    // scalar path state (decode_kernel.h)
    int wbase = -4096;
    LV<uint32_t> win;

    CIMG_DEV void init(uint8_t* lds_, int base_, int n, int cs, int csize, int lds_limit, int mail_)
    {
        lds = lds_; base = base_; iend = cs + csize; oend = base_ + n; mail = mail_;
        clampmax = (lds_limit - 4) & ~3;
        ip = cs; op = base_;
        dense_tokens = need_scalar = finished = false;
        kind = PAIR_NONE; rc = 0; wbase = -4096;
        FOR_LANES(l) { w0[l] = 0; w1[l] = 0; rank[l] = 0; d0[l] = 0; d1[l] = 0; d2[l] = 0; cnt[l] = 0; par[l] = false; win[l] = 0; }
        // the consumer looks at the mailbox before the first post(): LDS still holds what the previous workgroup left there
        FOR_LANES_W(l) {
            if (l < 16) {
                *reinterpret_cast<int32_t*>(lds + mail + 4 * l) = l == 0 ? (int32_t)PAIR_NONE : 0;
                *reinterpret_cast<int32_t*>(lds + mail + PAIR_SLOT_BYTES + 4 * l) = l == 0 ? (int32_t)PAIR_NONE : 0;
            }
        }
    }

    CIMG_DEV void fail(int code) { finished = true; rc = code; kind = PAIR_END; }

    // ---- wave A, in front of barrier 1: find the tokens of the next window (reads compressed bytes only) -------------
    CIMG_DEV void compute(int step)
    {
        slot = mail + (step & 1) * PAIR_SLOT_BYTES;
#if defined(CIMG_PAIR_PROF) && !defined(CIMG_EMULATE)
        const unsigned long long c0 = cimg_cycles();
        const bool sc = need_scalar;
        parse();
        const unsigned long long c1 = cimg_cycles();
        post();
        const unsigned long long c2 = cimg_cycles();
        if (sc) t_scalar += c1 - c0; else t_parse += c1 - c0;
        t_post += c2 - c1;
#else
        parse();
        post();
#endif
    }
    unsigned long long t_parse = 0, t_post = 0, t_scalar = 0;      // -DCIMG_PAIR_PROF builds only

    CIMG_DEV void parse()
    {
        if (need_scalar && !finished) {
            // found in the previous step; the consumer has nothing to do in this one
            need_scalar = false;
            const int r = scalar_sequence();
            if (r < 0) { fail(r); return; }
            if (r == 1) { finished = true; rc = op == oend ? 0 : ERR_DATA; kind = PAIR_END; return; }
        }
        need_scalar = false;
        if (finished) { kind = PAIR_END; return; }
        kind = PAIR_NONE;
        if (ip >= iend) { fail(ERR_DATA); return; }
        if (iend - ip < 24) { need_scalar = true; return; }
        // (this block is the batch parse of lz4_decode_wave, decode_kernel.h -- kept identical on purpose)
        LV<uint32_t> tb, o0, o1, ex;
        LV<int> lit_l, lsrc_l, walk_l, len_l, off_l, ml_l;
        LV<bool> good, litok;
        FOR_LANES(l) {
            const int at = ip + l;
            tb[l] = lds[imin(at, clampmax)];
            const uint32_t lb = lds[imin(at + 1, clampmax)];
            const int litn = (int)(tb[l] >> 4);
            const bool lext = litn == 15;
            lit_l[l] = lext ? 15 + (int)lb : litn;
            litok[l] = !lext | (lb < 255);
            lsrc_l[l] = l + (lext ? 2 : 1);
            const int hp = imin(ip + lsrc_l[l] + lit_l[l], clampmax);
            o0[l] = lds[hp];
            o1[l] = lds[hp + 1];
            ex[l] = lds[hp + 2];
        }
        FOR_LANES(l) {
            const int mln = (int)(tb[l] & 15);
            const bool has_ext = mln == 15;
            off_l[l] = (int)(o0[l] | (o1[l] << 8));
            ml_l[l] = mln + 4 + (has_ext ? (int)ex[l] : 0);
            const int lend = lsrc_l[l] + lit_l[l];
            const int nxt = lend + 2 + (has_ext ? 1 : 0);
            walk_l[l] = nxt + ((lit_l[l] >= 15) | (lend > 64) ? 1024 : 0);
            len_l[l] = lit_l[l] + ml_l[l];
            good[l] = litok[l] & (!has_ext | (ex[l] < 255)) & (ip + nxt < iend) & (off_l[l] != 0);
        }
        const uint64_t goodmask = ballot(good);
        tokens = 0;
        s = 0;
        if (dense_tokens) {
            LV<int> J, p, onei;
            FOR_LANES(l) {
                const int nx = walk_l[l];
                J[l] = (good[l] & (nx < 64)) ? nx : l;
                p[l] = 0;
                onei[l] = 1;
            }
            CIMG_UNROLL
            for (int k = 0; k < 5; ++k) {
                LV<int> pj, J2;
                lane_gather(J, p, pj);
                if (k < 4) lane_gather(J, J, J2);                       // both gathers in flight before either is used
                FOR_LANES(l) { p[l] = ((l >> k) & 1) ? pj[l] : p[l]; }
                if (k < 4) { FOR_LANES(l) { J[l] = J2[l]; } }
            }
            LV<int> mark;
            lane_scatter(onei, p, mark);
            LV<bool> on;
            FOR_LANES(l) { on[l] = mark[l] != 0; }
            tokens = ballot(on) & goodmask;
            const int last = readlane(p, 31);
            s = ((goodmask >> last) & 1) ? readlane(walk_l, last) : last;
        } else {
            while (s < 64 && ((goodmask >> s) & 1)) { tokens |= 1ull << s; s = readlane(walk_l, s); }
        }
        dense_tokens = popc64(tokens) >= 8;
        biglast = 0;
        if (s >= 1024) { s -= 1024; biglast = 1; }
        LV<int> tlen, opos;
        LV<bool> istok;
        FOR_LANES(l) {
            istok[l] = (tokens >> l) & 1;
            tlen[l] = istok[l] ? len_l[l] : 0;
        }
        int acc;
        wave_exscan(tlen, opos, acc);
        if (acc > oend - op) {
            const int room = oend - op;
            LV<bool> over;
            FOR_LANES(l) { over[l] = istok[l] & (opos[l] + tlen[l] > room); }
            const int f = ctz64(ballot(over));
            tokens &= (1ull << f) - 1;
            s = f;
            biglast = 0;
            acc = readlane(opos, f);
            FOR_LANES(l) { istok[l] = (tokens >> l) & 1; }
        }
        if (!tokens) { need_scalar = true; return; }
        LV<bool> badv;
        FOR_LANES(l) {
            const int dst = op + opos[l] + lit_l[l];
            const int src = dst - off_l[l];
            badv[l] = istok[l] & (src < base);
            const int off = off_l[l], ml = ml_l[l];
            // (up to 128 bytes: such a match goes in as two halves of at most 64, decode_kernel.h)
            par[l] = istok[l] & (ml <= 128) & ((off >= ml) | (off == 1)) & ((src + (off == 1 ? 1 : ml) <= op) | (off <= lit_l[l]));
            cnt[l] = par[l] ? (ml > 64 ? 2 : 1) : 0;
            d0[l] = dst | ((ml > 64 ? 64 : ml) << 18);
            d1[l] = src | (off == 1 ? 1 << 18 : 0);
            d2[l] = (dst + 64) | ((ml > 64 ? ml - 64 : 0) << 18);
            w0[l] = lit_l[l] | (ml << 9);
            w1[l] = opos[l] | (off << 16);
        }
        if (ballot(badv)) { fail(ERR_DATA); return; }
        parmask = ballot(par);
        if (popc64(parmask) < 3) { parmask = 0; FOR_LANES(l) { cnt[l] = 0; } }
        wave_exscan(cnt, rank, P);
        kind = PAIR_BATCH;
        win_ip = ip; win_op = op;
        ip += s;
        op += acc;
    }

    // ---- one sequence through the scalar path of lz4_decode_wave (runs at the start of the step after the one that
    // found it: wave B is idle in that step and every earlier window is in place).  Returns 0 = go on, 1 = the stream ended with its last literals, < 0 = error.
    CIMG_DEV int scalar_sequence()
    {
#define CIMG_P_FETCH8(dst, at)                                                                   \
        do {                                                                                     \
            const int at_ = (at);                                                                \
            if (at_ - wbase > 256 - 12 || at_ < wbase) {                                         \
                wbase = at_ & ~3;                                                                \
                FOR_LANES(l) { win[l] = *reinterpret_cast<const uint32_t*>(lds + imin(wbase + 4 * l, clampmax)); } \
            }                                                                                    \
            const int i_ = (at_ - wbase) >> 2;                                                   \
            const uint64_t d0_ = readlane(win, i_), d1_ = readlane(win, i_ + 1), d2_ = readlane(win, i_ + 2); \
            const int sh_ = (at_ & 3) * 8;                                                       \
            dst = ((d0_ | (d1_ << 32)) >> sh_) | (sh_ ? (d2_ << (64 - sh_)) : 0);                \
        } while (0)
#define CIMG_P_LENEXT(acc)                                                                       \
        do {                                                                                     \
            for (;;) {                                                                           \
                if (ip >= iend) return ERR_DATA;                                                 \
                LV<uint32_t> eb_;                                                                \
                LV<bool> stop_;                                                                  \
                FOR_LANES(l) {                                                                   \
                    eb_[l] = lds[imin(ip + l, clampmax)];                                        \
                    stop_[l] = (eb_[l] != 255) | (ip + l >= iend);                               \
                }                                                                                \
                const int f_ = ctz64(ballot(stop_));                                             \
                if (f_ < 64) {                                                                   \
                    if (ip + f_ >= iend) return ERR_DATA;                                        \
                    acc += 255 * f_ + (int)readlane(eb_, f_);                                    \
                    ip += f_ + 1;                                                                \
                    break;                                                                       \
                }                                                                                \
                acc += 255 * 64;                                                                 \
                ip += 64;                                                                        \
            }                                                                                    \
        } while (0)
        wbase = -4096;                                       // the register window may be stale: output moved since
        if (ip >= iend) return ERR_DATA;
        uint64_t q;
        CIMG_P_FETCH8(q, ip);
        const uint32_t token = (uint32_t)(q & 0xFF);
        int lit = (int)(token >> 4);
        int ml = (int)(token & 15);
        int offset;
        const bool ext = ml == 15;
        if ((lit <= 4 || (lit == 5 && !ext)) && iend - ip >= 8) {
            const int hdr = 1 + lit + 2;
            offset = (int)((q >> (8 * (1 + lit))) & 0xFFFF);
            int extra = 0;
            if (ext) {
                extra = (int)((q >> (8 * hdr)) & 0xFF);
                if (extra == 255) {
                    ip += hdr + 1;
                    ml += 255;
                    CIMG_P_LENEXT(ml);
                    extra = -1;
                } else {
                    ml += extra;
                }
            }
            if (extra >= 0) ip += hdr + (ext ? 1 : 0);
            if (lit > oend - op) return ERR_DATA;
            if (lit) {
                const uint64_t lits = q >> 8;
                FOR_LANES_W(l) { if (l < lit) lds[op + l] = (uint8_t)(lits >> (8 * (l & 7))); }
                op += lit;
            }
        } else {
            ip++;
            if (lit == 15) {
                CIMG_P_LENEXT(lit);
            }
            if (lit > iend - ip || lit > oend - op) return ERR_DATA;
            if (lit > 0) {
                if (lit >= 512) lds_copy_wide(lds, op, ip, lit); else lds_copy_bytes(lds, op, ip, lit);
                ip += lit;
                op += lit;
            }
            if (ip == iend) return 1;                        // a block ends with literals
            if (iend - ip < 2) return ERR_DATA;
            uint64_t e;
            CIMG_P_FETCH8(e, ip);
            offset = (int)(e & 0xFFFF);
            ip += 2;
            if (ext) {
                CIMG_P_LENEXT(ml);
            }
        }
        if (ip > iend) return ERR_DATA;
        if (offset == 0 || offset > op - base) return ERR_DATA;
        ml += 4;
        if (ml > oend - op) return ERR_DATA;
        const int src = op - offset;
        if (ml <= 64) {
            LV<uint32_t> mv;
            if (offset >= ml) {
                FOR_LANES(l) { mv[l] = lds[src + (l < ml ? l : 0)]; }
            } else if (offset == 1) {
                FOR_LANES(l) { mv[l] = lds[src]; }
            } else {
                const float inv = fast_rcp((float)offset);
                FOR_LANES(l) { mv[l] = lds[src + small_mod(l < ml ? l : 0, offset, inv)]; }
            }
            FOR_LANES_W(l) { if (l < ml) lds[op + l] = (uint8_t)mv[l]; }
        } else {
            lds_copy_match(lds, op, src, ml);
        }
        op += ml;
        return 0;
#undef CIMG_P_FETCH8
#undef CIMG_P_LENEXT
    }

    // ---- the window (or the end of the stream) -> this step's mailbox slot -----------------------------------------------
    CIMG_DEV void post()
    {
        // eleven wave-uniform words, stored by one lane (a per-lane select chain compiles to a nest of exec-mask branches:
        // measured 1100 cycles per post)
        FOR_LANES_W(l) {
            if (l == 0) {
                int32_t* h = reinterpret_cast<int32_t*>(lds + slot);
                h[0] = kind; h[1] = rc; h[2] = win_op; h[3] = win_ip;
                h[4] = (int32_t)(uint32_t)tokens; h[5] = (int32_t)(uint32_t)(tokens >> 32);
                h[6] = (int32_t)(uint32_t)parmask; h[7] = (int32_t)(uint32_t)(parmask >> 32);
                h[8] = P; h[9] = s; h[10] = biglast; h[11] = 0;
            }
        }
        if (kind == PAIR_BATCH) {
            FOR_LANES_W(l) {
                *reinterpret_cast<int32_t*>(lds + slot + PAIR_HDR_BYTES + 4 * l) = w0[l];
                *reinterpret_cast<int32_t*>(lds + slot + PAIR_HDR_BYTES + 256 + 4 * l) = w1[l];
                if (cnt[l] > 0) {
                    *reinterpret_cast<int32_t*>(lds + slot + PAIR_HDR_BYTES + 512 + 4 * rank[l]) = d0[l];
                    *reinterpret_cast<int32_t*>(lds + slot + PAIR_HDR_BYTES + 768 + 4 * rank[l]) = d1[l];
                }
                if (cnt[l] > 1) {                                   // second half: 64 bytes further on both sides (a fill keeps its byte)
                    *reinterpret_cast<int32_t*>(lds + slot + PAIR_HDR_BYTES + 512 + 4 * (rank[l] + 1)) = d2[l];
                    *reinterpret_cast<int32_t*>(lds + slot + PAIR_HDR_BYTES + 768 + 4 * (rank[l] + 1)) = ((d1[l] >> 18) & 1) ? d1[l] : d1[l] + 64;
                }
            }
        }
    }
};

// ---- wave B, in front of barrier 1: move the bytes of the window in the mailbox -----------------------------------------
CIMG_DEV void lz4_pair_consume(uint8_t* lds, int base, int mail, int lds_limit)
{
    LV<int32_t> hd;
    FOR_LANES(l) { hd[l] = *reinterpret_cast<const int32_t*>(lds + mail + 4 * (l & 15)); }
    if (readlane(hd, 0) != PAIR_BATCH) return;
    const int clampmax = (lds_limit - 4) & ~3;
    const int op = readlane(hd, 2), ip = readlane(hd, 3);
    const uint64_t tokens = (uint64_t)(uint32_t)readlane(hd, 4) | ((uint64_t)(uint32_t)readlane(hd, 5) << 32);
    const uint64_t parmask = (uint64_t)(uint32_t)readlane(hd, 6) | ((uint64_t)(uint32_t)readlane(hd, 7) << 32);
    const int P = imin(readlane(hd, 8), 64), s = readlane(hd, 9), biglast = readlane(hd, 10);
    (void)base;
    LV<uint32_t> tb;
    LV<int> lit_l, ml_l, opos, off_l, lsrc_l;
    LV<bool> istok;
    FOR_LANES(l) {
        tb[l] = lds[imin(ip + l, clampmax)];
        const int a = *reinterpret_cast<const int32_t*>(lds + mail + PAIR_HDR_BYTES + 4 * l);
        const int b = *reinterpret_cast<const int32_t*>(lds + mail + PAIR_HDR_BYTES + 256 + 4 * l);
        lit_l[l] = a & 0x1FF;
        ml_l[l] = (a >> 9) & 0x1FF;
        opos[l] = b & 0xFFFF;
        off_l[l] = (b >> 16) & 0xFFFF;
        lsrc_l[l] = l + (lit_l[l] >= 15 ? 2 : 1);
        istok[l] = (tokens >> l) & 1;
    }
    // literals: lane j belongs to the last token at or before j - 1 (decode_kernel.h)
    const int tlast = 63 - (int)__builtin_clzll(tokens);
    const int litlim = biglast ? tlast : imin(s, 64);
    LV<int> owner;
    LV<bool> is_lit;
    FOR_LANES(l) {
        const uint64_t below = tokens & ((1ull << l) - 1);
        owner[l] = below ? 63 - (int)__builtin_clzll(below) : 0;
        is_lit[l] = below != 0;
    }
    LV<int> own_lit, own_pos;
    lane_gather(lit_l, owner, own_lit);
    lane_gather(opos, owner, own_pos);
    FOR_LANES_W(l) {
        const int k = l - owner[l] - 1;
        if (is_lit[l] & (k < own_lit[l]) & (l < litlim)) lds[op + own_pos[l] + k] = (uint8_t)tb[l];
    }
    if (biglast) lds_copy_bytes(lds, op + readlane(opos, tlast), ip + readlane(lsrc_l, tlast), readlane(lit_l, tlast));
    // the lane-parallel matches: sixteen per step, descriptors straight from the mailbox
    for (int g = 0; g < P; g += 16) {
        LV<int> who, e0, e1;
        FOR_LANES(l) {
            who[l] = g + (l >> 2);
            e0[l] = *reinterpret_cast<const int32_t*>(lds + mail + PAIR_HDR_BYTES + 512 + 4 * (who[l] & 63));
            e1[l] = *reinterpret_cast<const int32_t*>(lds + mail + PAIR_HDR_BYTES + 768 + 4 * (who[l] & 63));
        }
        LV<u128> w;
        FOR_LANES(l) {
            const bool act = who[l] < P;
            const bool f = (e1[l] >> 18) & 1;
            const int sa = act ? (e1[l] & 0x3FFFF) + (f ? 0 : (l & 3) * 16) : base;
            const int a = sa & ~3;
            const uint32_t sh = (uint32_t)sa & 3u;
            const uint32_t q0 = *reinterpret_cast<const uint32_t*>(lds + a);
            const uint32_t q1 = *reinterpret_cast<const uint32_t*>(lds + a + 4);
            const uint32_t q2 = *reinterpret_cast<const uint32_t*>(lds + a + 8);
            const uint32_t q3 = *reinterpret_cast<const uint32_t*>(lds + a + 12);
            const uint32_t q4 = *reinterpret_cast<const uint32_t*>(lds + a + 16);
            const uint32_t x0 = alignbyte(q1, q0, sh);
            const uint32_t fb = (x0 & 0xFF) * 0x01010101u;
            w[l].x = f ? fb : x0;
            w[l].y = f ? fb : alignbyte(q2, q1, sh);
            w[l].z = f ? fb : alignbyte(q3, q2, sh);
            w[l].w = f ? fb : alignbyte(q4, q3, sh);
        }
        FOR_LANES_W(l) {
            const int rem = who[l] < P ? ((e0[l] >> 18) & 0x7F) - (l & 3) * 16 : 0;
            uint8_t* d = lds + (e0[l] & 0x3FFFF) + (l & 3) * 16;
            const uint32_t v[4] = {w[l].x, w[l].y, w[l].z, w[l].w};
            CIMG_UNROLL
            for (int k = 0; k < 16; k++) { if (rem > k) d[k] = (uint8_t)(v[k >> 2] >> (8 * (k & 3))); }
        }
    }
    // what is left runs in order
    uint64_t todo = tokens & ~parmask;
    while (todo) {
        const int t = ctz64(todo);
        todo &= todo - 1;
        const int dst = op + readlane(opos, t) + readlane(lit_l, t);
        const int offset = readlane(off_l, t);
        const int src = dst - offset;
        const int ml = readlane(ml_l, t);
        if (ml <= 64) {
            LV<uint32_t> mv;
            if (offset >= ml) {
                FOR_LANES(l) { mv[l] = lds[src + (l < ml ? l : 0)]; }
            } else if (offset == 1) {
                FOR_LANES(l) { mv[l] = lds[src]; }
            } else {
                const float inv = fast_rcp((float)offset);
                FOR_LANES(l) { mv[l] = lds[src + small_mod(l < ml ? l : 0, offset, inv)]; }
            }
            FOR_LANES_W(l) { if (l < ml) lds[dst + l] = (uint8_t)mv[l]; }
        } else {
            lds_copy_match(lds, dst, src, ml);
        }
    }
}

}  // namespace cimg
