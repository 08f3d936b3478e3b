// iterators/iterator.h -- forward iterator over a channel's chunks: dereferencing decompresses the
// current chunk into an internal buffer and hands out a chunk_span; moving on (or destruction)
// recompresses it and stores it back.  Same contract as the reference (compressed/iterators/iterator.h:
// operator* :97-141, destructor :72-93, ++ :143-156), including that a visited chunk is always written
// back.  One decode + one encode per chunk, each a single-chunk engine call; a whole-channel modify is
// cheaper through get_decompressed() + a fresh channel (two batched calls).
#pragma once
#include <cstddef>
#include <iterator>
#include <span>
#include <stdexcept>
#include <variant>
#include <vector>
#include "../blosc2/schunk.h"
#include "../containers/chunk_span.h"
#include "../macros.h"

namespace NAMESPACE_COMPRESSED_IMAGE
{
	template <typename T>
	struct channel_iterator
	{
		using iterator_category = std::forward_iterator_tag;
		using difference_type = std::ptrdiff_t;
		using value_type = container::chunk_span<T>;
		using pointer = value_type*;
		using reference = value_type&;

		channel_iterator() = default;
		channel_iterator(blosc2::schunk_var_ptr<T> schunk, blosc2::context_raw_ptr compression_context,
			blosc2::context_raw_ptr decompression_context, size_t chunk_index, size_t width, size_t height)
			: m_Schunk(std::move(schunk)), m_Cctx(compression_context), m_Dctx(decompression_context),
			m_Index(chunk_index), m_Width(width), m_Height(height)
		{
			if (m_Index > num_chunks())
				throw std::out_of_range(detail::text("chunk_index is out of range for total number of chunks in blosc2_schunk."
					" Max chunk number is ", num_chunks(), " but received ", m_Index));
			if (m_Width == 0 || m_Height == 0)
				throw std::runtime_error(detail::text("passed zero width or height to iterator which is not valid, expected at least 1 pixel in either dimensions."
					" Got [width: ", m_Width, " px, height: ", m_Height, " px]"));
		}

		channel_iterator(const channel_iterator& other)
			: m_Schunk(other.m_Schunk), m_Cctx(other.m_Cctx), m_Dctx(other.m_Dctx), m_Index(other.m_Index),
			m_Width(other.m_Width), m_Height(other.m_Height) {}              // the live chunk stays with the original
		channel_iterator& operator=(const channel_iterator& other)
		{
			if (this != &other)
			{
				flush();
				m_Schunk = other.m_Schunk; m_Cctx = other.m_Cctx; m_Dctx = other.m_Dctx;
				m_Index = other.m_Index; m_Width = other.m_Width; m_Height = other.m_Height;
				m_Live = npos;
			}
			return *this;
		}

		~channel_iterator()
		{
			try { flush(); } catch (...) {}
		}

		value_type operator*()
		{
			if (!m_Schunk || !m_Cctx || !m_Dctx || m_Index >= num_chunks())
				throw std::runtime_error("Invalid Iterator struct encountered, cannot dereference item");
			if (m_Live != m_Index)
			{
				flush();                                   // recompress the chunk handed out before
				const size_t elems = std::visit([&](auto& s) { return s.chunk_elements(m_Index); }, *m_Schunk);
				m_Pixels.resize(elems);
				std::visit([&](auto& s) { s.chunk(m_Dctx, std::span<T>(m_Pixels), m_Index); }, *m_Schunk);
				m_Live = m_Index;
			}
			return value_type(std::span<T>(m_Pixels), m_Width, m_Height, m_Index, chunk_bytes());
		}

		channel_iterator& operator++()
		{
			++m_Index;
			if (m_Index > num_chunks())
				throw std::out_of_range("Iterator: count exceeds number of chunks");
			return *this;
		}
		channel_iterator operator++(int)
		{
			channel_iterator before(*this);
			++(*this);
			return before;
		}

		bool operator==(const channel_iterator& other) const noexcept { return m_Index == other.m_Index && m_Schunk == other.m_Schunk; }
		bool operator!=(const channel_iterator& other) const noexcept { return !(*this == other); }

		size_t chunk_index() const noexcept { return m_Index; }
		size_t chunk_elements() const { return std::visit([](auto& s) { return s.chunk_elements(); }, *m_Schunk); }
		size_t chunk_bytes() const { return std::visit([](auto& s) { return s.chunk_bytes(); }, *m_Schunk); }

	private:
		static constexpr size_t npos = static_cast<size_t>(-1);
		blosc2::schunk_var_ptr<T> m_Schunk;
		blosc2::context_raw_ptr m_Cctx = nullptr;
		blosc2::context_raw_ptr m_Dctx = nullptr;
		size_t m_Index = 0;
		size_t m_Width = 0;
		size_t m_Height = 0;
		size_t m_Live = npos;                     // index of the chunk currently decompressed in m_Pixels
		std::vector<T> m_Pixels;

		size_t num_chunks() const { return m_Schunk ? std::visit([](auto& s) { return s.num_chunks(); }, *m_Schunk) : 0; }

		// write the live chunk back, compressed
		void flush()
		{
			if (m_Live == npos || !m_Schunk) return;
			std::vector<std::byte> scratch(blosc2::min_compressed_size(chunk_bytes()));
			const size_t n = blosc2::compress<T>(m_Cctx, std::span<const T>(m_Pixels), std::span<std::byte>(scratch));
			scratch.resize(n);
			const size_t at = m_Live;
			m_Live = npos;
			std::visit([&](auto& s) { s.set_chunk(std::move(scratch), at); }, *m_Schunk);
		}
	};
}
