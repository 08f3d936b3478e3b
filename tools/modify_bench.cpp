// tools/modify_bench.cpp -- the reference's "modify an image while it stays compressed" loop (benchmark/main.cpp:166-177,
// SURVEY section 3.4) on the host mirror: for every chunk of a 4096 x 4096 uint16 channel, pixel += 1 through the channel
// iterator; then three channels in lock step through ranges::zip.  Prints milliseconds per pass.  Run twice by
// tools/modify_bench.sh: double-buffered windows (default) and CIMG_ITERATOR_SERIAL=1 (decode -> modify -> encode in turn).
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <vector>
#include "compressed/channel.h"
#include "compressed/ranges.h"

using namespace compressed;
using clk = std::chrono::steady_clock;

int main()
{
	const size_t w = 4096, h = 4096;
	std::vector<uint16_t> d(w * h);
	uint32_t x = 12345;
	for (size_t i = 0; i < d.size(); ++i) { x = x * 1664525u + 1013904223u; d[i] = static_cast<uint16_t>((((i % w) / 64) * 37 + ((i / w) / 64) * 101) * 64 + (x >> 26)); }
	channel<uint16_t> a(std::span<const uint16_t>(d), w, h), b(std::span<const uint16_t>(d), w, h), c(std::span<const uint16_t>(d), w, h);
	auto ms = [](clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); };
	double best1 = 1e30, best3 = 1e30;
	for (int rep = 0; rep < 5; ++rep)
	{
		auto t0 = clk::now();
		for (auto chunk : a) for (auto& px : chunk) px = static_cast<uint16_t>(px + 1);
		best1 = std::min(best1, ms(t0));
		t0 = clk::now();
		for (auto [ca, cb, cc] : ranges::zip(a, b, c))
			for (auto [pa, pb, pc] : ranges::zip(ca, cb, cc)) { pa = static_cast<uint16_t>(pa + 1); pb = static_cast<uint16_t>(pb + 2); pc = static_cast<uint16_t>(pa + pb); }
		best3 = std::min(best3, ms(t0));
	}
	bool ok = true;
	auto back = a.get_decompressed();
	for (size_t i = 0; i < d.size(); i += 4097) ok = ok && back[i] == static_cast<uint16_t>(d[i] + 10);
	std::printf("modify pass, one 4096x4096 u16 channel (%zu chunks, ratio %.2f): %.2f ms = %.2f GB/s of pixels;  zip of three channels: %.2f ms = %.2f GB/s;  check %s\n",
		a.num_chunks(), double(a.uncompressed_size() * 2) / double(a.compressed_bytes()), best1, double(w * h * 2) / best1 / 1e6, best3, double(3 * w * h * 2) / best3 / 1e6, ok ? "ok" : "WRONG");
	return ok ? 0 : 1;
}
