// tests/cpp/iter_bench.cpp -- timing of the reference's "modify image compressed" pattern (iterate the chunks of
// every channel, touch every pixel, chunks are written back) on the host mirror.  Diagnostic, run on the GPU box:
//   g++ -std=c++20 -O2 -I include -I compressed-image_amd/include tests/cpp/iter_bench.cpp -o /tmp/iter_bench \
//       -L compressed-image_amd -lcimg_hip -Wl,-rpath,$PWD/compressed-image_amd && /tmp/iter_bench
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "compressed/image.h"
#include "compressed/ranges.h"

using namespace compressed;

int main()
{
	const size_t w = 4096, h = 4096;
	std::vector<std::vector<uint16_t>> planes(4, std::vector<uint16_t>(w * h));
	for (size_t c = 0; c < 4; ++c)
		for (size_t y = 0; y < h; ++y)
			for (size_t x = 0; x < w; ++x)
				planes[c][y * w + x] = static_cast<uint16_t>(0x0400 + ((((x / 64) * 37 + (y / 64) * 101 + c * 17) & 255) << 6) + ((x * 2654435761u + y * 40503u) >> 26));
	image<uint16_t> img(planes, w, h, { "R", "G", "B", "A" });
	const double bytes = 4.0 * w * h * sizeof(uint16_t);
	auto now = [] { return std::chrono::steady_clock::now(); };
	auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
	for (int rep = 0; rep < 3; ++rep)
	{
		auto t0 = now();
		for (size_t c = 0; c < 4; ++c)
			for (auto chunk : img.channel(c))
				for (auto& px : chunk) px = static_cast<uint16_t>(px + 1);
		auto t1 = now();
		for (auto [r, g, b, a] : ranges::zip(img.channel(0), img.channel(1), img.channel(2), img.channel(3)))
			for (auto [pr, pg, pb, pa] : ranges::zip(r, g, b, a)) { pr = static_cast<uint16_t>(pr + 1); pg = pr; pb = pr; pa = pr; }
		auto t2 = now();
		auto t3 = now();
		auto all = img.get_decompressed();
		auto t4 = now();
		std::printf("modify loop per channel: %.1f ms (%.2f GB/s)   zip over 4 channels: %.1f ms (%.2f GB/s)   get_decompressed: %.1f ms\n",
			ms(t0, t1), bytes / ms(t0, t1) / 1e6, ms(t1, t2), bytes / ms(t1, t2) / 1e6, ms(t3, t4));
		if (all[0][0] == 0) std::printf("?\n");
	}
	return 0;
}
