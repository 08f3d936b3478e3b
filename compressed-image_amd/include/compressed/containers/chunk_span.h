// containers/chunk_span.h -- a decompressed chunk as a span that knows where it sits in the image
// (reference compressed/containers/chunk_span.h:45-100).
#pragma once
#include <cstddef>
#include <ranges>
#include <span>
#include "../constants.h"
#include "../macros.h"

namespace NAMESPACE_COMPRESSED_IMAGE
{
	namespace container
	{
		template <typename T>
		struct chunk_span : public std::ranges::view_interface<chunk_span<T>>
		{
			using iterator = typename std::span<T>::iterator;

			chunk_span() = default;
			/// chunk_size is the nominal chunk size the index arithmetic uses.  The reference feeds BYTES here
			/// (iterator.h:140) although the arithmetic is in elements, so its x()/y() are only right for
			/// 1-byte pixels; the same convention is kept so results match.
			chunk_span(std::span<T> data, size_t width, size_t height, size_t chunk_index, size_t chunk_size)
				: m_Data(data), m_ChunkSize(chunk_size), m_Width(width), m_Height(height), m_ChunkIndex(chunk_index) {}

			size_t x(size_t index) const noexcept { return global_index(index) % m_Width; }
			size_t y(size_t index) const noexcept { return global_index(index) / m_Width; }
			size_t chunk_index() const noexcept { return m_ChunkIndex; }

			auto begin() const noexcept { return m_Data.begin(); }
			auto end() const noexcept { return m_Data.end(); }
			auto size() const noexcept { return m_Data.size(); }
			T* data() const noexcept { return m_Data.data(); }

		private:
			std::span<T> m_Data{};
			size_t m_ChunkSize = s_default_chunksize;
			size_t m_Width = 1;
			size_t m_Height = 1;
			size_t m_ChunkIndex = 0;

			size_t global_index(size_t index) const noexcept { return m_ChunkIndex * m_ChunkSize + index; }
		};
	}
}
