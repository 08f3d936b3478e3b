// tests/emu/fuzz_main.cpp -- TEST INFRASTRUCTURE: the emulated kernels under AddressSanitizer + UBSan.
// GPU sanitizers are not available on the pool, so memory safety of the decode paths against damaged input is
// checked here: chunks produced by the emulated encoder are corrupted (bit flips, length fields, truncation,
// random garbage) and decoded with the workgroup's LDS modelled as a host buffer of EXACTLY the launch size.
// Any out-of-range LDS / global access the kernels would make shows up as a sanitizer report (exit code != 0).
// A clean run also shows that every loop terminates on garbage (the process would hang otherwise; the test
// runs it under a timeout).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

extern "C" {
struct EmuCParams { int32_t typesize, clevel, blocksize, compcode, splitmode; uint8_t filters[6], filters_meta[6]; };
int emu_compress_batch(const EmuCParams* p, int nchunks, const uint8_t* raw, const int64_t* raw_off, const int32_t* nbytes,
                       uint8_t* comp, const int64_t* comp_off, const int32_t* destsize, int32_t* cbytes);
int emu_decompress_batch(int nchunks, const uint8_t* comp, const int64_t* comp_off, const int32_t* nbytes, const int32_t* blocksize,
                         uint8_t* raw, const int64_t* raw_off, int32_t* status);
int emu_lz4_encode(const uint8_t* src, int n, uint8_t* dst, int cap, int accel, int* need);
int emu_lz4_decode(const uint8_t* src, int csize, uint8_t* dst, int n);
void emu_set_write_order(int o);
void emu_set_lean(int on);
}

static std::mt19937 rng(12345);
static uint32_t rnd(uint32_t n) { return n ? rng() % n : 0; }

static std::vector<uint8_t> make_input(int kind, int n, int ts)
{
    std::vector<uint8_t> v((size_t)n);
    for (int i = 0; i < n; i++) {
        const int e = i / ts, b = i % ts;
        switch (kind) {
        case 0: v[(size_t)i] = (uint8_t)rng(); break;                                         // noise
        case 1: v[(size_t)i] = (uint8_t)(b == ts - 1 ? (e / 64) * 37 : rng());  break;          // smooth top byte, noisy rest
        case 2: v[(size_t)i] = (uint8_t)((e / 50) & 0xFF); break;                               // runs
        case 3: v[(size_t)i] = (uint8_t)((e * 7 + b) % 23); break;                              // short period
        default: v[(size_t)i] = 0; break;                                                       // zeros
        }
    }
    return v;
}

int main(int argc, char** argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 300;
    long decoded = 0, rejected = 0;
    for (int codec : {1, 0})                   // lz4, blosclz
    for (int ts : {1, 2, 4, 8}) {
        for (int filter : {0, 1, 2}) {
            for (int kind = 0; kind < 5; kind++) {
                const int blocksize = (int)(1024 * (1 + rnd(4))) / ts * ts;
                const int n = blocksize * (int)(1 + rnd(3)) + (int)rnd((uint32_t)blocksize) / ts * ts;
                std::vector<uint8_t> raw = make_input(kind, n, ts);
                EmuCParams p{};
                p.typesize = ts; p.clevel = 9; p.blocksize = blocksize; p.compcode = codec; p.splitmode = 3; p.filters[5] = (uint8_t)filter;
                std::vector<uint8_t> comp((size_t)n + 64 + 4096);
                const int64_t zero = 0;
                const int32_t nb = n, dest = n + 32;
                int32_t cb = 0;
                emu_set_write_order((int)rnd(3));
                if (emu_compress_batch(&p, 1, raw.data(), &zero, &nb, comp.data(), &zero, &dest, &cb) < 0 || cb <= 0) { fprintf(stderr, "encode failed ts %d filter %d kind %d\n", ts, filter, kind); return 2; }
                // the clean chunk must round-trip
                std::vector<uint8_t> back((size_t)n);
                int32_t st = 0, bs = 0;
                memcpy(&bs, comp.data() + 8, 4);
                if (emu_decompress_batch(1, comp.data(), &zero, &nb, &bs, back.data(), &zero, &st) < 0 || st != 0 || memcmp(back.data(), raw.data(), (size_t)n)) { fprintf(stderr, "round trip failed ts %d filter %d kind %d\n", ts, filter, kind); return 3; }
                for (int r = 0; r < rounds / 20 + 1 && cb > 48; r++) {      // (a 32-byte special chunk has no body to damage)
                    // exact-size copy of the chunk: reads past cbytes are reads past the allocation
                    std::vector<uint8_t> bad(comp.begin(), comp.begin() + cb);
                    const int how = (int)rnd(5);
                    if (how == 0) { for (int k = 0; k < 1 + (int)rnd(6); k++) bad[32 + rnd((uint32_t)cb - 32)] ^= (uint8_t)(1u << rnd(8)); }            // bit flips in the body
                    else if (how == 1) { const uint32_t at = 32 + rnd((uint32_t)cb - 36); const int32_t v = (int32_t)rng(); memcpy(&bad[at], &v, 4); }         // a wild 32-bit field
                    else if (how == 2) { for (uint32_t k = 32 + rnd((uint32_t)cb - 32); k < (uint32_t)cb; k++) bad[k] = (uint8_t)rng(); }                   // garbage tail
                    else if (how == 3) { bad[16 + rnd(16)] ^= (uint8_t)(1u << rnd(8)); bad[2] ^= (uint8_t)(rnd(2) << 4); }                                   // filter bytes / flags
                    else { for (int k = 0; k < 3; k++) bad[32 + rnd((uint32_t)cb - 32)] = 255; }                                                              // length-extension bytes
                    // nbytes / blocksize as the host reads them from the (possibly damaged) header are kept from the clean chunk:
                    // the C ABI validates the header fields it plans with before launching
                    emu_set_lean((int)rnd(2));
                    emu_decompress_batch(1, bad.data(), &zero, &nb, &bs, back.data(), &zero, &st);
                    (st < 0 ? rejected : decoded)++;
                }
            }
        }
    }
    emu_set_lean(1);
    // bare LZ4 streams: random bytes and damaged real streams, exact-size buffers
    for (int r = 0; r < rounds; r++) {
        const int n = 16 + (int)rnd(8000);
        std::vector<uint8_t> src = make_input((int)rnd(5), n, 1 + (int)rnd(2));
        std::vector<uint8_t> c((size_t)n + n / 255 + 64);
        int need = 0;
        int cs = emu_lz4_encode(src.data(), n, c.data(), (int)c.size(), 1, &need);
        if (cs <= 0 || cs >= n) continue;                           // the codec only ever decodes streams shorter than their plane
        std::vector<uint8_t> bad(c.begin(), c.begin() + cs), out((size_t)n);
        if (emu_lz4_decode(bad.data(), cs, out.data(), n) != 0 || memcmp(out.data(), src.data(), (size_t)n)) { fprintf(stderr, "lz4 round trip failed\n"); return 4; }
        for (int k = 0; k < 1 + (int)rnd(4); k++) bad[rnd((uint32_t)cs)] = (uint8_t)rng();
        emu_lz4_decode(bad.data(), cs, out.data(), n);
        std::vector<uint8_t> junk((size_t)(1 + rnd((uint32_t)n - 1)));
        for (auto& x : junk) x = (uint8_t)rng();
        emu_lz4_decode(junk.data(), (int)junk.size(), out.data(), n);
    }
    printf("fuzz ok: %ld damaged chunks decoded to something, %ld rejected, no sanitizer report\n", decoded, rejected);
    return 0;
}
