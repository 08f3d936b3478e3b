// encode_rt_kernel.h -- bit-exact LZ4_compress_fast with the byU16 hash table in REGISTERS (round 5).
//
// Replaces, for LZ4 / LZ4HC streams, the match search of lz4_encode_body (encode_kernel.h) behind blosc2_compress_ctx
// (blosc2/wrapper.h:139,172).  Why a second form: the encode launch is chains x latency, the number of chains a CU holds was set
// by LDS (plane + 16 KiB table = 32 KiB: five), and on sequence-dense data -- what photographs are -- every sequence paid four or
// five dependent LDS round trips of ~130 cycles around a table that is read and written ONE slot at a time by wave-uniform code.
// A table that is only ever touched one slot at a time does not need a memory: the 8192 x u16 entries are 64 dwords per lane,
// i.e. 64 VGPRs (two tuples of 32), entry h = half (h & 1) of lane ((h >> 1) & 63) of register (h >> 7).  A slot is read with a
// register-indexed move (s_set_gpr_idx_on) + v_readlane and written back with v_writelane: about a dozen scalar-issue
// instructions and NO memory round trip.  LDS then holds the plane and nothing else (16.6 KiB a chain instead of 32.8): EIGHT
// chains a CU at two waves a SIMD, 256 registers each.
//
// The search itself is the sequential algorithm, probe by probe in the order LZ4 takes them -- nothing speculative about the
// table, so there is no collision machinery and no roll-back rule to prove: a window of 64 upcoming probe positions is laid out
// (their four bytes and hashes in one LDS round trip), the probes are entered into the table a BATCH at a time (1, 1, 2, 4, 8, 16
// ...: the post-match probe and the first probe alone, because on dense data they are the hit 60 - 95 % of the time), the
// candidates of a batch are compared in ONE round trip, and the probes of a batch behind the first hit are taken out of the table
// again in reverse order (each put back what it displaced -- exactly the state sequential LZ4 has at the hit).  A single probe's
// candidate comparison already IS its match extension (256 bytes forwards, 64 backwards in the same round trip).
// Sequences are parked in lanes and written 64 at a time by emit_pending, as in the first form; the limited-output checks,
// `need` and the last literals are the same code.
#pragma once

namespace cimg {

// LDS of one chain: the plane, and room behind it for the unguarded reads of the match extension (up to 260 bytes past the end)
enum : int { RT_LDS_MARGIN = 272 };
CIMG_HD int encode_lds_bytes_rt(int stream_bytes) { return ((stream_bytes + 15) & ~15) + RT_LDS_MARGIN; }

#ifdef CIMG_EMULATE
struct RegTab { uint16_t e[8192]; };
#define CIMG_RT_DECL RegTab rt_; memset(rt_.e, 0, sizeof(rt_.e))
#define CIMG_RT_PARAMS RegTab& rt_
#define CIMG_RT_ARGS rt_
inline uint32_t rt_xchg(RegTab& t, uint32_t h, uint32_t pos) { const uint32_t o = t.e[h & 8191]; t.e[h & 8191] = (uint16_t)pos; return o; }
inline void rt_put(RegTab& t, uint32_t h, uint32_t pos) { t.e[h & 8191] = (uint16_t)pos; }
#else
// (two tuples, passed as two references: a struct of both is kept in scratch memory by the compiler)
typedef uint32_t rt_v32 __attribute__((ext_vector_type(32)));
#define CIMG_RT_DECL rt_v32 rt_a_ = 0, rt_b_ = 0
#define CIMG_RT_PARAMS rt_v32& rt_a_, rt_v32& rt_b_
#define CIMG_RT_ARGS rt_a_, rt_b_
// table[h] = pos, returns what was there.  h and pos are wave-uniform.
CIMG_DEV uint32_t rt_xchg(rt_v32& ta, rt_v32& tb, uint32_t h, uint32_t pos)
{
    const uint32_t r = (h >> 7) & 31, lane = (h >> 1) & 63, sh = (h & 1) << 4;
    uint32_t old;
    if (h & 0x1000) {
        const uint32_t w = tb[r];
        const uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)w, (int)lane);
        old = (d >> sh) & 0xFFFFu;
        const uint32_t nd = (d & ~(0xFFFFu << sh)) | (pos << sh);
        tb[r] = (uint32_t)cimg_writelane_i32((int)nd, (int)lane, (int)w);
    } else {
        const uint32_t w = ta[r];
        const uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)w, (int)lane);
        old = (d >> sh) & 0xFFFFu;
        const uint32_t nd = (d & ~(0xFFFFu << sh)) | (pos << sh);
        ta[r] = (uint32_t)cimg_writelane_i32((int)nd, (int)lane, (int)w);
    }
    return old;
}
CIMG_DEV void rt_put(rt_v32& ta, rt_v32& tb, uint32_t h, uint32_t pos) { (void)rt_xchg(ta, tb, h, pos); }
#endif

// how many probes the walk enters before it looks at their candidates: `done` probes of this search were looked at already
// (the first probe goes alone; the post-match probe in front of it travels with the refill)
#ifndef CIMG_RT_BATCH_SINGLES
#define CIMG_RT_BATCH_SINGLES 1
#endif
CIMG_DEV int rt_batch_size(int done)
{
    if (done < CIMG_RT_BATCH_SINGLES) return 1;
    if (done < CIMG_RT_BATCH_SINGLES + 2) return 2;
    if (done < CIMG_RT_BATCH_SINGLES + 6) return 4;
    if (done < CIMG_RT_BATCH_SINGLES + 14) return 8;
    return 16;
}

// Bit-exact LZ4_compress_fast(in, out, n, cap, accel) in limited-output mode (byU16 regime: n < 65547), by one wave, the table in
// registers.  in: LDS plane with RT_LDS_MARGIN readable bytes behind it.  Returns bytes written, 0 if the result does not fit cap,
// < 0 if a loop guard tripped; need_out = smallest cap that still succeeds.
//
// The search is ONE flat loop of steps with ONE place that touches the table (the loop marked "the table" below): the table is 64
// registers, and every further place that modified it -- or a loop nest that carried it -- made the compiler keep second copies of
// both tuples and move 64 registers at the joins.  A step = [lay a window out] -> enter a batch of probes into the table (or take the
// probes behind a hit out again) -> compare candidates / extend -> sequence.
CIMG_DEV int lz4_encode_rt_body(const uint8_t* in, int n, uint8_t* out_generic, int cap, int accel, int& need_out, uint64_t* dbg = nullptr, int item = 0)
{
    cimg_global_u8p out = CIMG_AS_GLOBAL(out_generic);
    CIMG_PROF_DECL;
    (void)dbg; (void)item;
    CIMG_RT_DECL;
    const int mflimit_p1 = n - 11, matchlimit = n - 5;
    const int s64 = accel << 6;
    const int f64 = skip_prefix(s64);
    int anchor = 0, op = 0, need = 0, np = 0;
    LV<int> P_anchor, P_lit, P_off, P_mcode;
    FOR_LANES(l) { P_anchor[l] = 0; P_lit[l] = 0; P_off[l] = 0; P_mcode[l] = 0; }
    int ending = 1;     // 1: the search reached the end of the plane; 0: the output does not fit; < 0: a loop guard tripped
    if (n >= 13) {
        // (position 0 is entered first: the table is all zero, so it already says so)
        int s = 1;      // position of probe 0 of the current search
        int sp = 0;     // 2: this window opens with the refill at s - 3 (lane 0) and the post-match probe at s - 1 (lane 1)
        int k0 = 0;     // probes of this search that earlier windows walked
        int j = 0, nv = 0;          // next lane of the window to walk, valid lanes (a prefix)
        bool lay_out = true;        // the next step starts with a new window
        int undo = 0;               // 1: this step takes the lanes ta, ta - 1, ... tb + 1 out of the table again, then extends the hit at lane tb
        int ta = 0, tb = 0, tdir = 1;
        int hit_p = 0, hit_old = 0;
        LV<int> pos;
        LV<uint32_t> W, H, OLD;
        FOR_LANES(l) { pos[l] = 0; W[l] = 0; H[l] = 0; OLD[l] = 0; }
        // every step walks at least one lane, lays a window out, or takes probes out that an earlier step entered
        for (int guard = 0; ; ++guard) {
            if (guard > 4 * n + 16) { ending = -1; break; }
            if (lay_out) {
                // ---- the window: lane l = probe k0 + l - sp of the skip schedule (the two special lanes in front) ------------
                lay_out = false;
                LV<bool> valid;
                if (s64 == 64 && k0 == 0) {
                    // acceleration 1, first window of a search: probes 0 .. 64 sit at s, s + 1, ... with gap 1
                    FOR_LANES(l) {
                        pos[l] = s - sp + l - (sp && l == 0 ? 1 : 0);       // (with the special lanes: s - 3, s - 1, s, s + 1, ...)
                        valid[l] = (l < sp) | (pos[l] + 1 <= mflimit_p1);
                    }
                } else {
                    FOR_LANES(l) {
                        const int k = k0 + l - sp;
                        int p = s + k, gap = 1;
                        if (k > 0) { p = s + 1 + skip_prefix(s64 + k - 1) - f64; gap = (s64 + k - 1) >> 6; }
                        if (l < sp) p = l == 0 ? s - 3 : s - 1;
                        pos[l] = p;
                        valid[l] = (l < sp) | (p + gap <= mflimit_p1);
                    }
                }
                nv = popc64(ballot(valid));
                FOR_LANES(l) { W[l] = lds_ld32u(in, valid[l] ? pos[l] : 0); H[l] = lz4_hash<13>(W[l]); }
                j = 0;
                CIMG_PROF_LAP(0); CIMG_PROF_COUNT(0);              // window laid out
            }
            int bs = 1;
            if (!undo) {
                if (j >= nv) {
                    if (nv < 64) break;                                 // the next probe would pass mflimit: last literals
                    k0 += 64 - sp;
                    sp = 0;
                    lay_out = true;
                    continue;
                }
                // the refill and the post-match probe go together; then batches that grow with the length of the search
                bs = j < sp ? sp - j : imin(rt_batch_size(k0 + j - sp), nv - j);
                ta = j; tb = j + bs; tdir = 1;
            }
            // ---- the table: the ONE place that reads and writes it ------------------------------------------------------------
            // entering: slot of lane t <- its position, what was there -> OLD[t]; taking out: slot of lane t <- OLD[t]
            for (int t = ta; t != tb; t += tdir) {
                const uint32_t v = undo ? readlane(OLD, t) : (uint32_t)readlane(pos, t);
                const uint32_t o = rt_xchg(CIMG_RT_ARGS, readlane(H, t), v);
                setlane(OLD, t, o);
            }
            CIMG_PROF_LAP(1); CIMG_PROF_COUNT(1);                  // table phase
            int p, old;
            if (undo) {
                undo = 0;
                p = hit_p; old = hit_old;
            } else if (bs == 1 || j < sp) {
                j = tb - 1;
                p = readlane(pos, j);
                old = (int)readlane(OLD, j);
            } else {
                LV<bool> hit;
                FOR_LANES(l) {
                    const bool mine = (l >= ta) & (l < tb);
                    hit[l] = mine & (lds_ld32u(in, mine ? (int)OLD[l] : 0) == W[l]);
                }
                const uint64_t hm = ballot(hit);
                CIMG_PROF_LAP(2);                                       // candidates of a batch compared
                if (!hm) { j = tb; continue; }
                j = ctz64(hm);
                p = readlane(pos, j);
                old = (int)readlane(OLD, j);
                if (j < tb - 1) {
                    // the probes behind the hit never happened: each gives its slot back, last first
                    undo = 1; ta = tb - 1; tb = j; tdir = -1;
                    hit_p = p; hit_old = old;
                    continue;
                }
            }
            // ---- candidate `old` for the probe at p: the comparison and, for a hit, the whole extension, in ONE round trip --------
            const int room = imin(p - anchor, old);                     // bytes the match may grow backwards (0 for the post-match probe)
            const int maxc = matchlimit - (p + 4);
            LV<int> len;
            LV<bool> stop, eq;
            FOR_LANES(l) {
                const uint32_t x = lds_ld32u(in, old + 4 * l) ^ lds_ld32u(in, p + 4 * l);
                const int kb = l < room ? l + 1 : 0;
                const uint32_t pa = in[p - kb], pb = in[old - kb];
                int ln = x ? (int)(__builtin_ctz(x) >> 3) : 4;
                if (l > 0) ln = imin(ln, imax(maxc - 4 * (l - 1), 0));
                len[l] = ln;
                stop[l] = ln < 4;
                eq[l] = (l < room) & (pa == pb);
            }
            const uint64_t sm = ballot(stop);
            CIMG_PROF_LAP(3); CIMG_PROF_COUNT(3);                  // candidate + extension round trip
            if (sm & 1) { j += 1; continue; }                           // the four bytes differ: no match at this probe
            int mcode;
            if (sm) { const int f = ctz64(sm); mcode = 4 * (f - 1) + readlane(len, f); }
            else mcode = match_more(in, p, old, maxc, 252, n);
            int backrun = ctz64(~ballot(eq));
            if (backrun == 64) {                                        // rare: more than 64 bytes backwards
                int left = room - 64;
                while (left > 0) {
                    FOR_LANES(l) { eq[l] = l < left && in[p - 1 - backrun - l] == in[old - 1 - backrun - l]; }
                    const int r = ctz64(~ballot(eq));
                    backrun += r; left -= r;
                    if (r < 64) break;
                }
            }
            CIMG_STAT(g_emu_matches);
            // ---- the sequence ----------------------------------------------------------------------------------------------------
            const int ip = p - backrun, mp = old - backrun;
            mcode += backrun;
            setlane(P_anchor, np, anchor);
            setlane(P_lit, np, ip - anchor);
            setlane(P_off, np, ip - mp);
            setlane(P_mcode, np, mcode);
            if (++np == 64) {
                if (!emit_pending(in, out, cap, op, need, np, P_anchor, P_lit, P_off, P_mcode)) { ending = 0; break; }
                np = 0;
            }
            const int ipe = ip + mcode + 4;
            CIMG_PROF_LAP(4); CIMG_PROF_COUNT(2);                  // sequence parked
            anchor = ipe;
            if (ipe >= mflimit_p1) break;
            s = ipe + 1;
            sp = 2;
            k0 = 0;
            lay_out = true;
        }
        if (ending <= 0) return ending;
    }
    if (np && !emit_pending(in, out, cap, op, need, np, P_anchor, P_lit, P_off, P_mcode)) return 0;
    // ---- last literals ----------------------------------------------------------------------------------------------------------
    {
        const int run = n - anchor;
        const int lhs = op + run + 1 + (run + 240) / 255;
        if (lhs > cap) return 0;
        need = imax(need, lhs);
        const uint32_t token = (uint32_t)((run >= 15 ? 15 : run) << 4);
        FOR_LANES(l) { if (l == 0) out[op] = (uint8_t)token; }
        op++;
        if (run >= 15) { emit_len_ext(out, op, run - 15); op += (run - 15) / 255 + 1; }
        emit_literals(in, anchor, out, op, run);
        op += run;
    }
    CIMG_PROF_LAP(6);
    CIMG_PROF_STORE(dbg, item);
    need_out = need;
    return op;
}

}  // namespace cimg
