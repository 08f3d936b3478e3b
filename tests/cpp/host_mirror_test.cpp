// tests/cpp/host_mirror_test.cpp -- the reference's OIIO-free C++ test cases, restated against the host
// mirror (compressed-image_amd/include/compressed).  Each block names the reference test it follows:
//   test/src/test_schunk.cpp:19-75, test_channel.cpp:20-126, test_image.cpp:552-859 (constructors,
//   iterator + zip read / modify), test_chunk_span.cpp:15-56, test_zip.cpp (lock-step, shortest range).
// Built twice by tests/test_host_mirror.py: against tests/emu/libcimg_hip_mock.so (CPU, `not gpu`) and
// against compressed-image_amd/libcimg_hip.so (`gpu`).
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <numeric>
#include <string>
#include <atomic>
#include <thread>
#include <vector>

#include "compressed/channel.h"
#include "compressed/image.h"
#include "compressed/image_algo.h"
#include "compressed/ranges.h"
#include "compressed/containers/chunk_span.h"

static int g_failures = 0, g_checks = 0;
#define CHECK(cond) do { ++g_checks; if (!(cond)) { ++g_failures; std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); } } while (0)
template <typename E, typename F> static bool throws(F&& f) { try { f(); } catch (const E&) { return true; } catch (...) { return false; } return false; }
template <typename F> static bool throws_any(F&& f) { try { f(); } catch (...) { return true; } return false; }

using namespace compressed;

template <typename T> static void schunk_cases()
{
	// test_schunk.cpp:19-34 -- an empty table
	{
		blosc2::schunk<T> s(128, 4096);
		auto dctx = blosc2::create_decompression_context(1);
		CHECK(s.to_uncompressed(dctx).size() == 0);
		auto raw = s.to_schunk();
		CHECK(raw != nullptr && raw->nchunks == 0);
	}
	// test_schunk.cpp:39-75 -- iota(4096), block 64, chunk 256, lz4 level 9
	{
		std::vector<T> data(4096);
		std::iota(data.begin(), data.end(), T{0});
		auto cctx = blosc2::create_compression_context<T>(1, enums::codec::lz4, 9, 128);
		auto dctx = blosc2::create_decompression_context(1);
		blosc2::schunk<T> s(std::span<const T>(data), 64, 256, cctx);
		CHECK(s.to_uncompressed(dctx) == data);
		auto raw = s.to_schunk();
		CHECK(static_cast<size_t>(raw->nchunks) == 4096 * sizeof(T) / 256);
		CHECK(static_cast<size_t>(raw->nbytes) / sizeof(T) == 4096);
		CHECK(s.chunk(dctx, 0).size() == 256 / sizeof(T));
		CHECK(s.num_chunks() == 4096 * sizeof(T) / 256 && s.size() == 4096);
		CHECK(throws<std::out_of_range>([&] { s.chunk(dctx, s.num_chunks()); }));
		std::vector<T> small(3);
		CHECK(throws<std::invalid_argument>([&] { s.chunk(dctx, std::span<T>(small), 0); }));
		if (sizeof(T) > 1) CHECK(throws<std::invalid_argument>([&] { blosc2::schunk<T> bad(64, 255); (void)bad; }));   // chunk size must divide by sizeof(T)
	}
}

static void channel_cases()
{
	// test_channel.cpp:20-41 -- should_fail
	CHECK(throws_any([] { channel<uint8_t> c(blosc2::schunk_var<uint8_t>(blosc2::schunk<uint8_t>(128, 4096)), 1, 1); }));
	{
		std::vector<uint8_t> d(50);
		CHECK(throws_any([&] { channel<uint8_t> c(std::span<const uint8_t>(d), 1, 1); }));
	}
	// test_channel.cpp:46-55 -- 50-byte buffer, defaults
	{
		std::vector<uint8_t> d(50);
		std::iota(d.begin(), d.end(), uint8_t{0});
		channel<uint8_t> c(std::span<const uint8_t>(d), 10, 5);
		CHECK(c.get_decompressed() == d);
		CHECK(c.compression() == enums::codec::lz4 && c.compression_level() == 9);
	}
	// test_channel.cpp:74-87 -- attributes of a blosclz channel
	{
		std::vector<uint8_t> d(50);
		channel<uint8_t> c(std::span<const uint8_t>(d), 10, 5, enums::codec::blosclz, 9);
		CHECK(c.width() == 10 && c.height() == 5);
		CHECK(c.compression() == enums::codec::blosclz);
		CHECK(c.compression_context() != nullptr && c.decompression_context() != nullptr);
		CHECK(c.uncompressed_size() == 50 && c.num_chunks() == 1);
		CHECK(c.get_decompressed() == d);
	}
	// test_channel.cpp:60-69 -- 8192 bytes, 128x64, lz4/9, block 128, chunk 4096 -> 2 chunks
	{
		std::vector<uint8_t> d(8192);
		std::iota(d.begin(), d.end(), uint8_t{0});
		channel<uint8_t> c(std::span<const uint8_t>(d), 128, 64, enums::codec::lz4, 9, 128, 4096);
		CHECK(c.num_chunks() == 2 && c.chunk_size() == 4096 && c.block_size() == 128);
		CHECK(c.get_decompressed() == d);
		std::vector<uint8_t> one(4096);
		c.get_chunk(std::span<uint8_t>(one), 1);
		CHECK(std::equal(one.begin(), one.end(), d.begin() + 4096));
		CHECK(throws<std::out_of_range>([&] { c.get_chunk(std::span<uint8_t>(one), 2); }));
	}
	// all four codecs of the reference construct, compress and round-trip (enums.h:18-24; lz4hc / zstd chunks are format-valid, not
	// the CPU libraries' bytes); a chunk table written elsewhere under such a label can be ADOPTED, read and rewritten
	{
		std::vector<uint8_t> d(8192);
		std::iota(d.begin(), d.end(), uint8_t{0});
		channel<uint8_t> src(std::span<const uint8_t>(d), 128, 64, enums::codec::lz4, 9, 128, 4096);
		for (auto codec : {enums::codec::zstd, enums::codec::lz4hc}) {
			channel<uint8_t> c(blosc2::schunk_var<uint8_t>(src.chunks()), 128, 64, codec, 5);
			CHECK(c.compression() == codec && c.compression_context() != nullptr && c.decompression_context() != nullptr);
			CHECK(c.get_decompressed() == d);
			std::vector<uint8_t> one(4096);
			c.get_chunk(std::span<uint8_t>(one), 1);
			CHECK(std::equal(one.begin(), one.end(), d.begin() + 4096));
			c.set_chunk(std::span<uint8_t>(one), 0);                      // chunk 0 := the pixels of chunk 1, re-encoded with `codec`
			std::vector<uint8_t> back(4096);
			c.get_chunk(std::span<uint8_t>(back), 0);
			CHECK(back == one);
			c.update_nthreads(2, 128);
			channel<uint8_t> fresh(std::span<const uint8_t>(d), 128, 64, codec, 5);
			CHECK(fresh.compression() == codec && fresh.get_decompressed() == d);
			for (auto chunk : fresh) for (auto& px : chunk) px = uint8_t(px + 1);
			auto plus = fresh.get_decompressed();
			bool all = true;
			for (size_t i = 0; i < d.size(); ++i) all = all && plus[i] == uint8_t(d[i] + 1);
			CHECK(all);
		}
	}
	// test_channel.cpp:92-126 -- iterator read, then modify to 128 and re-read (u16 16x8)
	{
		std::vector<uint16_t> d(16 * 8, 255);
		channel<uint16_t> c(std::span<const uint16_t>(d), 16, 8);
		size_t seen = 0;
		for (auto chunk : c) for (auto& px : chunk) { CHECK(px == 255); ++seen; }
		CHECK(seen == 128);
		for (auto chunk : c) for (auto& px : chunk) px = 128;
		for (auto v : c.get_decompressed()) CHECK(v == 128);
	}
	// the iterator works on windows of 8 chunks: 21 chunks = two full windows and a short one; every chunk gets its
	// index, a second pass reads them back and rewrites only the odd ones, a partial pass stops inside a window
	{
		const size_t w = 32, h = 84, chunk = 32 * 4 * sizeof(uint16_t);           // 4 scanlines per chunk -> 21 chunks
		std::vector<uint16_t> d(w * h, 7);
		channel<uint16_t> c(std::span<const uint16_t>(d), w, h, enums::codec::lz4, 9, 128, chunk);
		CHECK(c.num_chunks() == 21);
		for (auto it = c.begin(); it != c.end(); ++it) { auto ch = *it; for (auto& px : ch) px = static_cast<uint16_t>(it.chunk_index()); }
		auto all = c.get_decompressed();
		for (size_t i = 0; i < all.size(); ++i) CHECK(all[i] == i / (w * 4));
		for (auto it = c.begin(); it != c.end(); ++it)
		{
			auto ch = *it;
			for (auto& px : ch) { CHECK(px == it.chunk_index()); if (it.chunk_index() & 1) px = 1000; }
		}
		all = c.get_decompressed();
		for (size_t i = 0; i < all.size(); ++i) CHECK(all[i] == (((i / (w * 4)) & 1) ? 1000 : i / (w * 4)));
		{
			auto it = c.begin();
			for (int k = 0; k < 10; ++k, ++it) { auto ch = *it; for (auto& px : ch) px = 5; }     // stops in the second window
		}                                                                                     // destructor flushes it
		all = c.get_decompressed();
		for (size_t i = 0; i < all.size(); ++i) CHECK(all[i] == (i / (w * 4) < 10 ? 5 : (((i / (w * 4)) & 1) ? 1000 : i / (w * 4))));
		// lazy channel: windows mix fill-value slots and real chunks
		auto z = channel<uint16_t>::full(w, h, 9, enums::codec::lz4, 9, 128, chunk);
		for (auto it = z.begin(); it != z.end(); ++it) { auto ch = *it; if (it.chunk_index() == 3 || it.chunk_index() == 12) for (auto& px : ch) px = 77; }
		all = z.get_decompressed();
		for (size_t i = 0; i < all.size(); ++i) CHECK(all[i] == ((i / (w * 4) == 3 || i / (w * 4) == 12) ? 77 : 9));
	}
	// double-buffered windows (iterator.h: write-back of window k-1 and read-ahead of window k+1 run behind the loop body):
	// (a) dereferences that skip windows -- the read-ahead is not the window wanted -- and a pass that touches one chunk in
	// three; (b) three channels in lock step through ranges::zip, 21 chunks each; (c) the chunks an iterator pass leaves
	// behind are byte-identical to compressing the same pixels directly, and (CIMG_TEST_DUMP) are written out so that
	// tests/test_host_mirror.py can hold them against the oracle
	{
		const size_t w = 32, h = 84, chunk = 32 * 4 * sizeof(uint16_t), per = w * 4;    // 21 chunks of 128 pixels
		std::vector<uint16_t> d(w * h);
		for (size_t i = 0; i < d.size(); ++i) d[i] = static_cast<uint16_t>((i / 5) * 3);
		channel<uint16_t> c(std::span<const uint16_t>(d), w, h, enums::codec::lz4, 9, 128, chunk);
		std::vector<uint16_t> want = d;
		{
			auto it = c.begin();
			{ auto ch = *it; for (auto& px : ch) px = 11; }                                  // chunk 0 (window 0..7, read-ahead 8..15)
			for (int k = 0; k < 17; ++k) ++it;
			{ auto ch = *it; CHECK(it.chunk_index() == 17); for (auto& px : ch) px = 17; }   // window 17..20: not the read-ahead
			++it; ++it;
			{ auto ch = *it; CHECK(ch.size() == per); for (auto& px : ch) px = static_cast<uint16_t>(px + 1); }   // chunk 19, same window
		}
		for (size_t i = 0; i < per; ++i) { want[i] = 11; want[17 * per + i] = 17; want[19 * per + i] = static_cast<uint16_t>(want[19 * per + i] + 1); }
		CHECK(c.get_decompressed() == want);
		for (auto it = c.begin(); it != c.end(); ++it)
			if (it.chunk_index() % 3 == 1) { auto ch = *it; for (auto& px : ch) px = static_cast<uint16_t>(px ^ 0x5555); }
		for (size_t i = 0; i < want.size(); ++i) if ((i / per) % 3 == 1) want[i] = static_cast<uint16_t>(want[i] ^ 0x5555);
		CHECK(c.get_decompressed() == want);

		channel<uint16_t> r(std::span<const uint16_t>(d), w, h, enums::codec::lz4, 9, 128, chunk), g(std::span<const uint16_t>(d), w, h, enums::codec::blosclz, 9, 128, chunk),
			b(std::span<const uint16_t>(d), w, h, enums::codec::lz4, 5, 128, chunk);
		size_t visited = 0;
		for (auto [cr, cg, cb] : ranges::zip(r, g, b))
		{
			for (auto [pr, pg, pb] : ranges::zip(cr, cg, cb)) { CHECK(pr == pg && pg == pb); pr = static_cast<uint16_t>(pr + 1); pg = static_cast<uint16_t>(pg + 2); pb = static_cast<uint16_t>(pr + pg); }
			++visited;
		}
		CHECK(visited == 21);
		auto dr = r.get_decompressed(), dg = g.get_decompressed(), db = b.get_decompressed();
		bool ok = true;
		for (size_t i = 0; i < d.size(); ++i) ok = ok && dr[i] == static_cast<uint16_t>(d[i] + 1) && dg[i] == static_cast<uint16_t>(d[i] + 2) && db[i] == static_cast<uint16_t>(2 * d[i] + 3);
		CHECK(ok);

		// (c) bytes: iterator pass vs direct compression of the same pixels
		channel<uint16_t> direct(std::span<const uint16_t>(dr), w, h, enums::codec::lz4, 9, 128, chunk);
		auto raw_it = std::visit([](auto& s) { return s.to_schunk(); }, r.chunks());
		auto raw_direct = std::visit([](auto& s) { return s.to_schunk(); }, direct.chunks());
		CHECK(raw_it->nchunks == 21 && raw_direct->nchunks == 21);
		auto csize = [](const uint8_t* ch) { return static_cast<size_t>(ch[12]) | (static_cast<size_t>(ch[13]) << 8) | (static_cast<size_t>(ch[14]) << 16) | (static_cast<size_t>(ch[15]) << 24); };
		bool same = true;
		for (int64_t k = 0; k < 21 && raw_it->nchunks == 21 && raw_direct->nchunks == 21; ++k)
			same = same && csize(raw_it->data[k]) == csize(raw_direct->data[k]) && std::equal(raw_it->data[k], raw_it->data[k] + csize(raw_it->data[k]), raw_direct->data[k]);
		CHECK(same);
		if (const char* path = std::getenv("CIMG_TEST_DUMP"))
			if (FILE* f = std::fopen(path, "wb"))
			{
				const uint32_t hdr[6] = { 2, 128, static_cast<uint32_t>(chunk), 21, static_cast<uint32_t>(w), static_cast<uint32_t>(h) };   // typesize, blocksize, chunk bytes, chunks
				std::fwrite(hdr, sizeof(hdr), 1, f);
				std::fwrite(dr.data(), sizeof(uint16_t), dr.size(), f);
				for (int64_t k = 0; k < raw_it->nchunks; ++k)
				{
					const uint32_t n = static_cast<uint32_t>(csize(raw_it->data[k]));
					std::fwrite(&n, 4, 1, f);
					std::fwrite(raw_it->data[k], 1, n, f);
				}
				std::fclose(f);
			}
	}
	// lazy factories (python test_channel.py:71-190 through the C++ surface)
	{
		auto z = channel<float>::zeros(123, 456, enums::codec::lz4, 9, s_default_blocksize, 123 * sizeof(float) * 10);
		CHECK(z.num_chunks() == 46 && z.uncompressed_size() == 123 * 456);
		CHECK(z.compressed_bytes() == 46 * sizeof(float));              // lazy chunks cost one T each
		std::vector<float> buf(z.chunk_elems(0));
		z.get_chunk(std::span<float>(buf), 0);
		CHECK(std::all_of(buf.begin(), buf.end(), [](float v) { return v == 0.f; }));
		std::fill(buf.begin(), buf.end(), 100.f);
		z.set_chunk(std::span<float>(buf), 0);
		auto all = z.get_decompressed();
		CHECK(std::all_of(all.begin(), all.begin() + buf.size(), [](float v) { return v == 100.f; }));
		CHECK(std::all_of(all.begin() + buf.size(), all.end(), [](float v) { return v == 0.f; }));
		auto f = channel<float>::full_like(z, 7.5f);
		CHECK(f.width() == 123 && f.chunk_size() == z.chunk_size());
		auto fv = f.get_decompressed();
		CHECK(std::all_of(fv.begin(), fv.end(), [](float v) { return v == 7.5f; }));
	}
}

template <typename T> static void image_cases()
{
	// test_image.cpp:552-859 -- constant planes 255 / 0 / 199, 64x16, block 256, chunk 1024 and chunk 768 (short last chunk)
	for (size_t chunk : { size_t{1024}, size_t{768} })
	{
		const size_t w = 64, h = 16;
		std::vector<std::vector<T>> planes = { std::vector<T>(w * h, T(255)), std::vector<T>(w * h, T(0)), std::vector<T>(w * h, T(199)) };
		image<T> img(planes, w, h, { "R", "G", "B" }, enums::codec::lz4, 9, 256, chunk);
		CHECK(img.num_channels() == 3 && img.width() == w && img.height() == h);
		CHECK(img.channelnames() == (std::vector<std::string>{ "R", "G", "B" }));
		const size_t aligned = chunk / sizeof(T) / w * w * sizeof(T);
		CHECK(img.chunk_size() == aligned);
		CHECK(img.channel(0).num_chunks() == (w * h * sizeof(T) + aligned - 1) / aligned);
		CHECK(img.get_decompressed() == planes);
		// iterator + zip: read
		size_t n = 0;
		for (auto [r, g, b] : ranges::zip(img.channel("R"), img.channel("G"), img.channel("B")))
			for (auto [pr, pg, pb] : ranges::zip(r, g, b)) { CHECK(pr == T(255) && pg == T(0) && pb == T(199)); ++n; }
		CHECK(n == w * h);
		// iterator + zip: modify to 12 / 13 / 14 and re-read
		for (auto [r, g, b] : ranges::zip(img.channel(0), img.channel(1), img.channel(2)))
			for (auto [pr, pg, pb] : ranges::zip(r, g, b)) { pr = T(12); pg = T(13); pb = T(14); }
		auto back = img.get_decompressed();
		CHECK(std::all_of(back[0].begin(), back[0].end(), [](T v) { return v == T(12); }));
		CHECK(std::all_of(back[1].begin(), back[1].end(), [](T v) { return v == T(13); }));
		CHECK(std::all_of(back[2].begin(), back[2].end(), [](T v) { return v == T(14); }));
		CHECK(img.compression_ratio() > 1.0);
		// add / extract / remove
		std::vector<T> extra(w * h, T(90));
		img.add_channel(std::span<const T>(extra), w, h, "A");
		CHECK(img.num_channels() == 4 && img.get_channel_offset("A") == 3);
		CHECK(throws<std::invalid_argument>([&] { img.add_channel(std::span<const T>(extra.data(), w * (h - 1)), w, h - 1, "bad"); }));
		CHECK(throws<std::invalid_argument>([&] { (void)img.channel("nope"); }));
		CHECK(throws<std::out_of_range>([&] { (void)img.channel(9); }));
		auto a = img.extract_channel("A");
		CHECK(a.get_decompressed() == extra && img.num_channels() == 3);
		img.remove_channel(size_t{0});
		CHECK(img.channelnames() == (std::vector<std::string>{ "G", "B" }));
	}
}

static void chunk_span_cases()
{
	// test_chunk_span.cpp:15-56 known answers (width 128, height 128, chunk index 1, chunk size 128)
	std::vector<uint8_t> data(128);
	container::chunk_span<uint8_t> base(std::span<uint8_t>(data), 128, 128, 0, 128);
	CHECK(base.x(9) == 9 && base.y(5) == 0 && base.chunk_index() == 0);
	container::chunk_span<uint8_t> next(std::span<uint8_t>(data), 128, 128, 1, 128);
	CHECK(next.x(9) == 9 && next.y(5) == 1 && next.x(135) == 7 && next.y(129) == 2);
	CHECK(next.size() == 128);
}

static void zip_cases()
{
	std::vector<int> a{ 1, 2, 3, 4 }, b{ 10, 20, 30 };
	int sum = 0, n = 0;
	for (auto [x, y] : ranges::zip(a, b)) { sum += x * y; x = 0; ++n; }
	CHECK(n == 3 && sum == 10 + 40 + 90);                       // stops at the shortest range
	CHECK(a[0] == 0 && a[2] == 0 && a[3] == 4);                   // elements are references
}

// interleave / deinterleave on the host (image_algo.h) and the producer path image::from_interleaved: interleaved pixels
// are uploaded once, split into planes on the device, compressed from there -- pixels AND chunk bytes equal those of the
// planar constructor
template <typename T> static void interleaved_cases()
{
	const size_t w = 67, h = 31, n = 3;                                   // odd sizes: ragged tiles, a short last chunk
	std::vector<std::vector<T>> planes(n, std::vector<T>(w * h));
	for (size_t c = 0; c < n; ++c)
		for (size_t i = 0; i < w * h; ++i) planes[c][i] = static_cast<T>((i / 9) * (c + 2) + c * 40 + (i % 3));
	std::vector<T> mixed(w * h * n);
	{
		std::vector<std::span<const T>> in;
		for (auto& p : planes) in.push_back(std::span<const T>(p));
		image_algo::interleave<T>(std::span<T>(mixed), in);
		for (size_t i = 0; i < w * h; i += 17) for (size_t c = 0; c < n; ++c) CHECK(mixed[i * n + c] == planes[c][i]);
		std::vector<std::vector<T>> back(n, std::vector<T>(w * h));
		image_algo::deinterleave<T>(std::span<const T>(mixed), back);
		CHECK(back == planes);
		std::vector<T> small(5);
		CHECK(throws<std::invalid_argument>([&] { image_algo::interleave<T>(std::span<T>(small), in); }));
		std::vector<std::vector<T>> uneven{ std::vector<T>(4), std::vector<T>(5) };
		CHECK(throws<std::invalid_argument>([&] { image_algo::deinterleave<T>(std::span<const T>(mixed), uneven); }));
	}
	const size_t chunk = w * 4 * sizeof(T);                               // 4 scanlines per chunk -> 8 chunks per channel, the last one short
	auto a = image<T>::from_interleaved(std::span<const T>(mixed), w, h, n, { "R", "G", "B" }, enums::codec::lz4, 9, 256, chunk);
	image<T> b(planes, w, h, { "R", "G", "B" }, enums::codec::lz4, 9, 256, chunk);
	CHECK(a.num_channels() == n && a.width() == w && a.height() == h);
	CHECK(a.get_decompressed() == planes);
	for (size_t c = 0; c < n; ++c)
	{
		auto ra = std::visit([](auto& s) { return s.to_schunk(); }, a.channel(c).chunks());
		auto rb = std::visit([](auto& s) { return s.to_schunk(); }, b.channel(c).chunks());
		CHECK(ra->nchunks == rb->nchunks && ra->nchunks == 8);
		auto csize = [](const uint8_t* ch) { return static_cast<size_t>(ch[12]) | (static_cast<size_t>(ch[13]) << 8) | (static_cast<size_t>(ch[14]) << 16) | (static_cast<size_t>(ch[15]) << 24); };
		bool same = ra->nchunks == rb->nchunks;
		for (int64_t k = 0; same && k < ra->nchunks; ++k)
			same = csize(ra->data[k]) == csize(rb->data[k]) && std::equal(ra->data[k], ra->data[k] + csize(ra->data[k]), rb->data[k]);
		CHECK(same);
	}
	CHECK(throws_any([&] { (void)image<T>::from_interleaved(std::span<const T>(mixed.data(), mixed.size() - 1), w, h, n); }));
}

// several host threads on the one shared engine: batches are serialised inside the library, results stay correct
static void thread_cases()
{
	std::atomic<int> bad{ 0 };
	auto work = [&](int seed) {
		for (int rep = 0; rep < 4; ++rep)
		{
			const size_t w = 64 + 8 * static_cast<size_t>(seed), h = 48;
			std::vector<uint16_t> d(w * h);
			for (size_t i = 0; i < d.size(); ++i) d[i] = static_cast<uint16_t>((i / 7) * (seed + 1) + rep);
			channel<uint16_t> c(std::span<const uint16_t>(d), w, h, enums::codec::lz4, 9, 512, w * 8 * sizeof(uint16_t));
			if (c.get_decompressed() != d) ++bad;
			for (auto chunk : c) for (auto& px : chunk) px = static_cast<uint16_t>(px + 3);
			auto back = c.get_decompressed();
			for (size_t i = 0; i < d.size(); ++i) if (back[i] != static_cast<uint16_t>(d[i] + 3)) { ++bad; break; }
		}
	};
	std::vector<std::thread> pool;
	for (int t = 0; t < 4; ++t) pool.emplace_back(work, t);
	for (auto& t : pool) t.join();
	CHECK(bad.load() == 0);
	// one thread only compresses (the two-step _begin / _fetch batch), another only decompresses an unrelated channel
	// and a third uses the single-chunk blosc2_*_ctx calls: none may invalidate another's pending fetch
	std::vector<uint8_t> base(40 * 50);
	for (size_t i = 0; i < base.size(); ++i) base[i] = static_cast<uint8_t>(i / 9);
	channel<uint8_t> shared(std::span<const uint8_t>(base), 40, 50, enums::codec::blosclz, 9, 256, 400);
	std::atomic<int> errors{ 0 };
	auto guarded = [&](auto&& fn) { return [&errors, fn] { try { for (int rep = 0; rep < 24; ++rep) fn(rep); } catch (const std::exception&) { ++errors; } }; };
	std::thread tc(guarded([&](int rep) {
		std::vector<uint16_t> d(64 * 32, static_cast<uint16_t>(rep));
		channel<uint16_t> c(std::span<const uint16_t>(d), 64, 32, rep % 2 ? enums::codec::lz4 : enums::codec::blosclz, 9, 512, 1024);
		if (c.num_chunks() != 4) ++errors;
	}));
	std::thread td(guarded([&](int) { if (shared.get_decompressed() != base) ++errors; }));
	std::thread ts(guarded([&](int rep) {
		std::vector<uint8_t> one(400);
		shared.get_chunk(std::span<uint8_t>(one), static_cast<size_t>(rep % 5));
		if (!std::equal(one.begin(), one.end(), base.begin() + 400 * (rep % 5))) ++errors;
	}));
	tc.join(); td.join(); ts.join();
	CHECK(errors.load() == 0);
}

int main()
{
	schunk_cases<uint8_t>(); schunk_cases<uint16_t>(); schunk_cases<uint32_t>(); schunk_cases<float>();
	channel_cases();
	image_cases<uint8_t>(); image_cases<uint16_t>(); image_cases<uint32_t>(); image_cases<float>();
	chunk_span_cases();
	zip_cases();
	interleaved_cases<uint8_t>();
	interleaved_cases<uint16_t>();
	interleaved_cases<float>();
	thread_cases();
	std::printf("%d checks, %d failures\n", g_checks, g_failures);
	return g_failures ? 1 : 0;
}
