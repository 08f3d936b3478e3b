// image_algo.h -- interleave / deinterleave between one buffer of R G B A R G B A ... elements and one buffer per channel
// (reference: compressed/image_algo.h:33-64 interleave, :84-111 and :130-166 deinterleave; same names, arguments and
// std::invalid_argument conditions).  These are the HOST forms, for callers that hold the pixels in host memory and want
// them there.  The producer path of an image does not go through them: image<T>::from_interleaved (image.h) uploads the
// interleaved scanlines once and splits them on the device (csrc/deinterleave_kernel.h), where the codec reads them.
#pragma once
#include <algorithm>
#include <cstddef>
#include <span>
#include <stdexcept>
#include <vector>
#include "macros.h"

namespace NAMESPACE_COMPRESSED_IMAGE
{
	namespace image_algo
	{
		/// buffer[idx * spans.size() + i] = spans[i][idx].  buffer must hold exactly spans[0].size() * spans.size() elements.
		template <typename T>
		void interleave(std::span<T> buffer, const std::vector<std::span<const T>>& spans)
		{
			if (spans.empty())
				throw std::invalid_argument("Interleave: No spans provided for interleaving.");
			const std::size_t count = spans.front().size(), n = spans.size();
			for (const auto& s : spans)
				if (s.size() != count)
					throw std::invalid_argument("Interleave: All input spans must have the same size.");
			if (buffer.size() != count * n)
				throw std::invalid_argument("Interleave: Provided buffer is not large enough to hold all the elements to interleave.");
			// blocks of pixels keep every source span and the destination inside the cache while n streams are merged
			constexpr std::size_t block = 4096;
			for (std::size_t b = 0; b < count; b += block)
			{
				const std::size_t e = std::min(count, b + block);
				for (std::size_t i = 0; i < n; ++i)
				{
					const T* src = spans[i].data();
					T* dst = buffer.data() + i;
					for (std::size_t idx = b; idx < e; ++idx) dst[idx * n] = src[idx];
				}
			}
		}

		/// channel_spans[i][idx] = interleaved[idx * channel_spans.size() + i].
		template <typename T>
		void deinterleave(std::span<const T> interleaved, std::vector<std::span<T>>& channel_spans)
		{
			if (channel_spans.empty())
				throw std::invalid_argument("Deinterleave: No output spans provided.");
			const std::size_t count = channel_spans.front().size(), n = channel_spans.size();
			if (!std::all_of(channel_spans.begin(), channel_spans.end(), [count](const std::span<T>& s) { return s.size() == count; }))
				throw std::invalid_argument("Deinterleave: All output spans must have the same size.");
			if (interleaved.size() != count * n)
				throw std::invalid_argument("Deinterleave: Input buffer size does not match the expected size for deinterleaving.");
			constexpr std::size_t block = 4096;
			for (std::size_t b = 0; b < count; b += block)
			{
				const std::size_t e = std::min(count, b + block);
				for (std::size_t i = 0; i < n; ++i)
				{
					const T* src = interleaved.data() + i;
					T* dst = channel_spans[i].data();
					for (std::size_t idx = b; idx < e; ++idx) dst[idx] = src[idx * n];
				}
			}
		}

		template <typename T>
		void deinterleave(std::span<const T> interleaved, std::vector<std::vector<T>>& channel_vecs)
		{
			std::vector<std::span<T>> spans;
			for (auto& v : channel_vecs) spans.push_back(std::span<T>(v.data(), v.size()));
			deinterleave<T>(interleaved, spans);
		}
	}
}
