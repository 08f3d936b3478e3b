// tests/cpp/ref_headers_test.cpp -- the REFERENCE'S OWN host headers over this repository's C ABI.
//
// Compiled only in the build container, by tests/test_reference_headers.py, with
//     -I /root/reference/compressed_image/include      the reference's headers, read where they lie (nothing is copied)
//     -I include                                       THIS repository's blosc2.h: the seam
//     -I tests/cpp/compat                              <format> served by {fmt} (this image's libstdc++ has none)
// and linked against tests/emu/libcimg_hip_mock.so (the C ABI served by the host lane emulator: no GPU here).  It replays the
// checks of the reference's test/src/test_schunk.cpp:19-75 -- an empty schunk; iota(4096) of u8 / u16 / u32 / f32 with block 64 and
// chunk 256: element count, equality, nchunks, nbytes, chunk(0).size() -- plus the lazy_schunk cases of test_lazyschunk.cpp, through
// blosc2::compress / decompress / schunk / lazy_schunk exactly as the reference compiled them.  Test infrastructure: not an oracle.
// (the reference's headers lean on what their own translation units include in front of them -- test_schunk.cpp:3-9 -- and on
// MSVC's transitive includes: <cassert>, <ranges>, <execution>, <algorithm> come first here for the same reason)
#include <algorithm>
#include <cassert>
#include <cstdint>
#include <cstdio>
#include <execution>
#include <numeric>
#include <ranges>
#include <span>
#include <string>
#include <thread>
#include <vector>

#include <compressed/blosc2/lazyschunk.h>
#include <compressed/blosc2/schunk.h>
#include <compressed/blosc2/wrapper.h>
#ifdef REF_WITH_ITERATOR
#include <compressed/blosc2/typedefs.h>          // (iterator.h names schunk_var_ptr and leaves the include to channel.h:18)
#include <compressed/iterators/iterator.h>
#endif

static int g_failed = 0;
#define CHECK(x) do { if (!(x)) { std::printf("FAILED %s:%d  %s\n", __FILE__, __LINE__, #x); ++g_failed; } } while (0)

template <typename T>
static void schunk_cases(const char* name)
{
    namespace b2 = compressed::blosc2;
    {   // "Schunk: initialize with chunk size" (test_schunk.cpp:19-35)
        b2::schunk<T> super_chunk(128, 4096);
        auto ctx = b2::create_decompression_context(std::thread::hardware_concurrency());
        auto decompressed = super_chunk.to_uncompressed(ctx);
        CHECK(decompressed.size() == 0);
        auto raw_schunk = super_chunk.to_schunk();
        CHECK(raw_schunk != nullptr);
    }
    {   // "Schunk: initialize with data" (test_schunk.cpp:40-75)
        std::vector<T> data(4096);
        std::iota(data.begin(), data.end(), T{0});
        auto ctx = b2::create_compression_context<T>(std::thread::hardware_concurrency(), compressed::enums::codec::lz4, 9, 128);
        b2::schunk<T> super_chunk(std::span<const T>(data), 64, 256, ctx);
        auto decomp_ctx = b2::create_decompression_context(std::thread::hardware_concurrency());
        auto decompressed = super_chunk.to_uncompressed(decomp_ctx);
        CHECK(decompressed.size() == 4096);
        CHECK(decompressed == data);
        auto raw_schunk = super_chunk.to_schunk();
        CHECK(raw_schunk->nchunks == (int64_t)(4096 * sizeof(T) / 256));
        CHECK(raw_schunk->nbytes / (int64_t)sizeof(T) == 4096);
        auto chunk = super_chunk.chunk(decomp_ctx, 0);
        CHECK(chunk.size() == 256 / sizeof(T));
        // set_chunk / append_chunk round trip (schunk.h:189-248)
        std::vector<T> other(256 / sizeof(T), T{7});
        super_chunk.set_chunk(ctx, std::span<T>(other), 3);
        auto back = super_chunk.chunk(decomp_ctx, 3);
        CHECK(back == other);
    }
    {   // lazy_schunk: a constant channel stays lazy until a chunk is set (lazyschunk.h:176-281)
        b2::lazy_schunk<T> lazy(T{25}, 4096, 64, 256);
        auto ctx = b2::create_compression_context<T>(1, compressed::enums::codec::lz4, 9, 64);
        auto dctx = b2::create_decompression_context(1);
        auto all = lazy.to_uncompressed(dctx);
        CHECK(all.size() == 4096);
        bool same = true;
        for (const T& v : all) same = same && v == T{25};
        CHECK(same);
        std::vector<T> other(256 / sizeof(T), T{90});
        lazy.set_chunk(ctx, std::span<T>(other), 1);
        auto c1 = lazy.chunk(dctx, 1);
        CHECK(c1 == other);
        auto raw = lazy.to_schunk();
        CHECK(raw->nbytes / (int64_t)sizeof(T) == 4096);
    }
    std::printf("%-8s ok so far (failed checks: %d)\n", name, g_failed);
}

#ifdef REF_WITH_ITERATOR
// channel.h itself stops at nlohmann/json.hpp (absent from this image: a library the reference needs, not stubbed), so the channel's
// own loop -- `for (auto chunk : channel)`, channel.h:311-340 -- is replayed one layer down, on the iterator it is made of
// (iterators/iterator.h:30-141): every element of every chunk += 1 through channel_iterator, then the whole schunk is read back.
template <typename T>
static void iterator_case()
{
    namespace b2 = compressed::blosc2;
    const size_t width = 64, height = 48, chunk_bytes = 512;
    std::vector<T> data(width * height);
    std::iota(data.begin(), data.end(), T{0});
    auto cctx = b2::create_compression_context<T>(1, compressed::enums::codec::lz4, 9, 128);
    auto dctx = b2::create_decompression_context(1);
    auto var = std::make_shared<b2::schunk_var<T>>(b2::schunk<T>(std::span<const T>(data), 128, chunk_bytes, cctx));
    const size_t nchunks = std::get<b2::schunk<T>>(*var).num_chunks();
    CHECK(nchunks == width * height * sizeof(T) / chunk_bytes);
    {
        compressed::channel_iterator<T> it(var, cctx.get(), dctx.get(), 0, width, height);
        compressed::channel_iterator<T> end(var, cctx.get(), dctx.get(), nchunks, width, height);
        size_t visited = 0;
        for (; it != end; ++it) {
            auto chunk = *it;
            for (auto& v : chunk) v = (T)(v + T{1});
            ++visited;
        }
        CHECK(visited == nchunks);
    }   // (the last chunk is recompressed when the iterator goes out of scope, iterator.h:72-93)
    auto back = std::get<b2::schunk<T>>(*var).to_uncompressed(dctx);
    CHECK(back.size() == data.size());
    bool same = back.size() == data.size();
    for (size_t i = 0; same && i < data.size(); ++i) same = back[i] == (T)(data[i] + T{1});
    CHECK(same);
}
#endif

int main()
{
#ifdef REF_WITH_ITERATOR
    iterator_case<uint8_t>();
    iterator_case<uint16_t>();
    iterator_case<float>();
    std::printf("iterator ok so far (failed checks: %d)\n", g_failed);
#endif
    schunk_cases<uint8_t>("uint8");
    schunk_cases<uint16_t>("uint16");
    schunk_cases<uint32_t>("uint32");
    schunk_cases<float>("float");
    std::printf(g_failed ? "REFERENCE HEADERS: %d check(s) failed\n" : "REFERENCE HEADERS: all checks passed%.0d\n", g_failed);
    return g_failed ? 1 : 0;
}
