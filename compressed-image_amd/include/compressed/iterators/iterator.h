// iterators/iterator.h -- forward iterator over a channel's chunks: dereferencing decompresses the
// current chunk into an internal buffer and hands out a chunk_span; moving on (or destruction)
// recompresses it and stores it back.  Same contract as the reference (compressed/iterators/iterator.h:
// operator* :97-141, destructor :72-93, ++ :143-156), including that a visited chunk is always written
// back.  What differs is the granularity of the engine calls: the reference decodes and re-encodes one chunk at
// a time; here the iterator works on a WINDOW of up to 8 chunks -- one batched decode when the window is entered
// (into a recycled page-locked buffer), one batched encode of the visited chunks when it is left or the iterator
// dies.  A visited chunk therefore reaches the channel when its window is flushed, not at ++; code that reads a
// chunk back through the channel while still iterating over the same window sees the old bytes.
#pragma once
#include <cstddef>
#include <algorithm>
#include <iterator>
#include <memory>
#include <span>
#include <stdexcept>
#include <variant>
#include <vector>
#include "../blosc2/schunk.h"
#include "../containers/chunk_span.h"
#include "../detail/pinned_pool.h"
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
				m_WinFirst = npos;
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
			if (m_WinFirst == npos || m_Index < m_WinFirst || m_Index >= m_WinFirst + m_WinElems.size())
			{
				flush();                                   // recompress what the previous window handed out
				load_window(m_Index);
			}
			const size_t k = m_Index - m_WinFirst;
			m_Visited[k] = true;
			T* px = reinterpret_cast<T*>(m_Buffer.get()) + m_WinOffset[k];
			return value_type(std::span<T>(px, m_WinElems[k]), m_Width, m_Height, m_Index, chunk_bytes());
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
		static constexpr size_t s_window_chunks = 8;
		size_t m_WinFirst = npos;                 // first chunk of the decoded window (npos: none)
		std::vector<size_t> m_WinOffset, m_WinElems;   // element offset / count of each window chunk inside m_Buffer
		std::vector<bool> m_Visited;              // handed out since the window was loaded -> written back on flush
		std::shared_ptr<std::byte> m_Buffer;      // the window's pixels (recycled page-locked memory)

		size_t num_chunks() const { return m_Schunk ? std::visit([](auto& s) { return s.num_chunks(); }, *m_Schunk) : 0; }

		// decode chunks [first, first + s_window_chunks) with ONE engine call
		void load_window(size_t first)
		{
			const size_t count = std::min(s_window_chunks, num_chunks() - first);
			m_WinOffset.assign(count, 0);
			m_WinElems.assign(count, 0);
			m_Visited.assign(count, false);
			size_t total = 0;
			for (size_t k = 0; k < count; ++k)
			{
				m_WinOffset[k] = total;
				m_WinElems[k] = std::visit([&](auto& s) { return s.chunk_elements(first + k); }, *m_Schunk);
				total += m_WinElems[k];
			}
			m_Buffer = detail::pinned_pool::get().arena(std::max<size_t>(total * sizeof(T), 1));
			std::vector<blosc2::batch::target> work;
			std::visit([&](auto& s) { s.plan_decode_range(reinterpret_cast<T*>(m_Buffer.get()), first, count, work); }, *m_Schunk);
			blosc2::batch::decompress(work);
			m_WinFirst = first;
		}

		// write the visited chunks of the window back, compressed with ONE engine call
		void flush()
		{
			if (m_WinFirst == npos || !m_Schunk) return;
			std::vector<blosc2::batch::piece> pieces;
			std::vector<size_t> where;
			for (size_t k = 0; k < m_Visited.size(); ++k)
				if (m_Visited[k])
				{
					pieces.push_back({ m_Buffer.get() + m_WinOffset[k] * sizeof(T), m_WinElems[k] * sizeof(T) });
					where.push_back(m_WinFirst + k);
				}
			const size_t first = m_WinFirst;
			m_WinFirst = npos;
			(void)first;
			if (pieces.empty()) return;
			auto chunks = blosc2::batch::compress(m_Cctx, pieces, chunk_bytes());
			for (size_t i = 0; i < chunks.size(); ++i)
				std::visit([&](auto& s) { s.set_chunk(std::move(chunks[i]), where[i]); }, *m_Schunk);
		}
	};
}
