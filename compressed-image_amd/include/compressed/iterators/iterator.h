// iterators/iterator.h -- forward iterator over a channel's chunks: dereferencing decompresses the
// current chunk into an internal buffer and hands out a chunk_span; moving on (or destruction)
// recompresses it and stores it back.  Same contract as the reference (compressed/iterators/iterator.h:
// operator* :97-141, destructor :72-93, ++ :143-156), including that a visited chunk is always written
// back.  What differs is the granularity of the engine calls: the reference decodes and re-encodes one chunk at
// a time; here the iterator works on a WINDOW of up to 8 chunks (a quarter of the channel if that is less) -- one batched decode when the window is entered
// (into a recycled page-locked buffer), one batched encode of the visited chunks when it is left or the iterator
// dies.  A visited chunk therefore reaches the channel when its window is flushed, not at ++; code that reads a
// chunk back through the channel while still iterating over the same window sees the old bytes.
//
// The windows are DOUBLE-BUFFERED (the reference's decode -> modify -> encode ping-pong is strictly
// serial per channel, iterator.h:118-126).  Entering window k hands the window just left to a helper task that
// (1) recompresses it and stores its chunks, then (2) decodes window k + 1 into a second page-locked buffer -- both
// while the caller's loop body works on window k.  The next window switch only waits for that task.  The engine calls
// are the same batched ones (one encode, one decode per window); they are serialised by the engine's own lock, so
// several iterators (ranges::zip over three channels) interleave their windows on one device.  The helper task never
// WRITES the channel's chunk table: it hands the recompressed chunks back and the iterator's own thread stores them at
// the next window switch (wait()), so reading the channel's sizes inside the loop body does not race with it.
// Consequences, beyond the window rule above: chunks of window k + 1 are read up to one window early, so writing them
// through the channel (set_chunk) while an iterator is about to enter them is undefined (the reference is strictly serial
// here; CIMG_ITERATOR_SERIAL=1 gives its order); an exception of the helper task (a codec error) surfaces at the next
// window switch or is swallowed by the destructor, like the reference's destructor does (iterator.h:72-93).
#pragma once
#include <cstddef>
#include <cstdlib>
#include <algorithm>
#include <future>
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
				finish();
				m_Schunk = other.m_Schunk; m_Cctx = other.m_Cctx; m_Dctx = other.m_Dctx;
				m_Index = other.m_Index; m_Width = other.m_Width; m_Height = other.m_Height;
			}
			return *this;
		}

		~channel_iterator()
		{
			try { finish(); } catch (...) {}
		}

		value_type operator*()
		{
			if (!m_Schunk || !m_Cctx || !m_Dctx || m_Index >= num_chunks())
				throw std::runtime_error("Invalid Iterator struct encountered, cannot dereference item");
			if (!m_Cur.holds(m_Index)) enter(m_Index);
			const size_t k = m_Index - m_Cur.first;
			m_Cur.visited[k] = true;
			T* px = reinterpret_cast<T*>(m_Cur.buffer.get()) + m_Cur.offset[k];
			return value_type(std::span<T>(px, m_Cur.elems[k]), m_Width, m_Height, m_Index, chunk_bytes());
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
		// chunks per window: up to 8 (32 MiB of pixels at the default chunk size -- enough to fill the device), but at least four
		// windows per channel so that the helper task has something to overlap with
		static constexpr size_t s_max_window_chunks = 8;
		static size_t window_chunks(size_t total_chunks) noexcept { return std::clamp<size_t>((total_chunks + 3) / 4, 1, s_max_window_chunks); }

		// a decoded run of chunks [first, first + elems.size()) in one page-locked buffer
		struct window
		{
			size_t first = npos;
			std::vector<size_t> offset, elems;        // element offset / count of each chunk inside buffer
			std::vector<bool> visited;                // handed out since the window was loaded -> written back on flush
			std::shared_ptr<std::byte> buffer;        // recycled page-locked memory
			bool holds(size_t index) const noexcept { return first != npos && index >= first && index < first + elems.size(); }
			bool overlaps(size_t lo, size_t hi) const noexcept { return first != npos && lo < first + elems.size() && first < hi; }
			void drop() { first = npos; offset.clear(); elems.clear(); visited.clear(); buffer.reset(); }
		};
		// the visited chunks of a window, recompressed with ONE engine call, and where they belong
		struct recoded
		{
			std::vector<blosc2::byte_buffer> chunks;
			std::vector<size_t> where;
		};
		window m_Cur;                                  // the window the caller is working on
		std::shared_ptr<window> m_Ahead;               // filled by the helper task: the window after m_Cur
		std::future<recoded> m_Task;                   // the previous window recompressed (stored by wait()), then the read-ahead

		size_t num_chunks() const { return m_Schunk ? std::visit([](auto& s) { return s.num_chunks(); }, *m_Schunk) : 0; }

		// decode chunks [first, first + window_chunks) with ONE engine call
		static void load_window(blosc2::schunk_var<T>& schunk, window& w, size_t first)
		{
			const size_t total_chunks = std::visit([](auto& s) { return s.num_chunks(); }, schunk);
			const size_t count = std::min(window_chunks(total_chunks), total_chunks - first);
			w.offset.assign(count, 0);
			w.elems.assign(count, 0);
			w.visited.assign(count, false);
			size_t total = 0;
			for (size_t k = 0; k < count; ++k)
			{
				w.offset[k] = total;
				w.elems[k] = std::visit([&](auto& s) { return s.chunk_elements(first + k); }, schunk);
				total += w.elems[k];
			}
			w.buffer = detail::pinned_pool::get().arena(std::max<size_t>(total * sizeof(T), 1));
			std::vector<blosc2::batch::target> work;
			std::visit([&](auto& s) { s.plan_decode_range(reinterpret_cast<T*>(w.buffer.get()), first, count, work); }, schunk);
			blosc2::batch::decompress(work);
			w.first = first;
		}

		static recoded recode_window(size_t nominal_chunk_bytes, blosc2::context_raw_ptr cctx, window& w)
		{
			recoded r;
			if (w.first == npos) return r;
			std::vector<blosc2::batch::piece> pieces;
			for (size_t k = 0; k < w.visited.size(); ++k)
				if (w.visited[k])
				{
					pieces.push_back({ w.buffer.get() + w.offset[k] * sizeof(T), w.elems[k] * sizeof(T) });
					r.where.push_back(w.first + k);
				}
			w.first = npos;
			if (pieces.empty()) return r;
			r.chunks = blosc2::batch::compress(cctx, pieces, nominal_chunk_bytes);
			return r;
		}
		// the only place an iterator WRITES the chunk table -- always on the thread that owns the iterator
		static void apply(blosc2::schunk_var<T>& schunk, recoded&& r)
		{
			for (size_t i = 0; i < r.chunks.size(); ++i)
				std::visit([&](auto& s) { s.set_chunk(std::move(r.chunks[i]), r.where[i]); }, schunk);
			r.chunks.clear(); r.where.clear();
		}
		// write the visited chunks of a window back (caller's thread)
		static void store_window(blosc2::schunk_var<T>& schunk, blosc2::context_raw_ptr cctx, window& w)
		{
			const size_t nominal = std::visit([](auto& s) { return s.chunk_bytes(); }, schunk);
			apply(schunk, recode_window(nominal, cctx, w));
		}

		static bool serial()
		{
			static const bool on = [] { const char* v = std::getenv("CIMG_ITERATOR_SERIAL"); return v && *v && *v != '0'; }();
			return on;
		}

		// wait for the helper task and put the chunks it recompressed into the table; its exception (if any) is rethrown here
		void wait()
		{
			if (m_Task.valid()) apply(*m_Schunk, m_Task.get());
		}

		// The caller moves on to the window that starts at (or holds) `index`: take the read-ahead if it is the right one,
		// else decode now; then hand the window just left to the helper task, which also reads the next one ahead.
		void enter(size_t index)
		{
			wait();
			window left = std::move(m_Cur);
			m_Cur.drop();
			if (m_Ahead && m_Ahead->holds(index)) m_Cur = std::move(*m_Ahead);
			else
			{
				// not the sequential case (first dereference, or the iterator was moved by hand): a window that shares
				// chunks with the one just left must see them written back first
				if (left.overlaps(index, index + s_max_window_chunks)) store_window(*m_Schunk, m_Cctx, left);
				load_window(*m_Schunk, m_Cur, index);
			}
			m_Ahead.reset();
			if (serial())                                  // diagnostics (CIMG_ITERATOR_SERIAL=1): no helper task, the reference's order
			{
				store_window(*m_Schunk, m_Cctx, left);
				return;
			}
			const size_t next = m_Cur.first + m_Cur.elems.size();
			const bool ahead = next < num_chunks();
			if (left.first == npos && !ahead) return;
			auto target = ahead ? std::make_shared<window>() : std::shared_ptr<window>();
			m_Ahead = target;
			// The task only READS the chunk table (the compressed bytes of window k + 1, which this iterator does not write while
			// the task runs) and never writes it: what it recompressed comes back through the future and is stored by wait() on
			// this thread.  The caller's own reads during the loop body (num_chunks, chunk_bytes, ...) are therefore reads
			// beside reads.
			const size_t nominal = chunk_bytes();
			m_Task = std::async(std::launch::async,
				[schunk = m_Schunk, cctx = m_Cctx, left = std::move(left), target, next, nominal]() mutable
				{
					recoded r = recode_window(nominal, cctx, left);
					left.drop();
					if (target) load_window(*schunk, *target, next);
					return r;
				});
		}

		// everything handed out so far reaches the channel; nothing stays in flight
		void finish()
		{
			if (!m_Schunk) { m_Cur.drop(); m_Ahead.reset(); return; }
			std::exception_ptr err;
			try { wait(); } catch (...) { err = std::current_exception(); }
			m_Ahead.reset();
			try { store_window(*m_Schunk, m_Cctx, m_Cur); } catch (...) { if (!err) err = std::current_exception(); }
			m_Cur.drop();
			if (err) std::rethrow_exception(err);
		}
	};
}
