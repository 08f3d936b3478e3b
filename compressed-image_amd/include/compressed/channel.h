// channel.h -- compressed::channel<T>: one planar image channel stored as blosc2 chunks, with the
// reference's public surface (compressed/channel.h: constructors :97-201, zeros/full factories
// :219-304, iteration :309-327, accessors :353-491, get_chunk :502-516, set_chunk :527-538,
// get_decompressed :545-558).  Compression of the whole channel and get_decompressed are single
// batched calls into the MI355X engine.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <span>
#include <stdexcept>
#include <thread>
#include <variant>
#include <vector>

#include "blosc2/schunk.h"
#include "constants.h"
#include "enums.h"
#include "iterators/iterator.h"
#include "macros.h"
#include "util.h"

namespace NAMESPACE_COMPRESSED_IMAGE
{
	template <typename T>
	struct channel
	{
		using iterator = channel_iterator<T>;

		channel(channel&&) noexcept = default;
		channel& operator=(channel&&) noexcept = default;
		channel(const channel&) = delete;
		channel& operator=(const channel&) = delete;

		/// A valid-but-empty channel (one lazy element), as the reference's default constructor gives.
		channel()
		{
			m_Schunk = std::make_shared<blosc2::schunk_var<T>>(blosc2::lazy_schunk<T>(T{}, 1, s_default_blocksize, s_default_chunksize));
			make_contexts(s_default_blocksize);
		}

		/// Compress `data` (width * height elements).  Chunks are aligned to whole scanlines.
		channel(const std::span<const T> data, size_t width, size_t height,
			enums::codec compression_codec = enums::codec::lz4, uint8_t compression_level = 9,
			size_t block_size = s_default_blocksize, size_t chunk_size = s_default_chunksize)
			: m_Codec(compression_codec), m_CompressionLevel(util::ensure_compression_level(compression_level)), m_Width(width), m_Height(height)
		{
			if (data.size() != width * height)
				throw std::runtime_error(detail::text("Invalid channel data passed. Expected its size to match up to width * height (", width, " * ", height,
					") which would be ", width * height, ". Instead received ", data.size()));
			make_contexts(block_size);
			const size_t aligned = util::align_chunk_to_scanlines_bytes<T>(m_Width, chunk_size);
			m_Schunk = std::make_shared<blosc2::schunk_var<T>>(blosc2::schunk<T>(data, block_size, aligned, m_CompressionContext));
		}

		/// Adopt an existing chunk table.
		channel(blosc2::schunk_var<T> schunk, size_t width, size_t height,
			enums::codec compression_codec = enums::codec::lz4, uint8_t compression_level = 9)
			: m_Codec(compression_codec), m_CompressionLevel(util::ensure_compression_level(compression_level)), m_Width(width), m_Height(height)
		{
			const size_t have = std::visit([](auto& s) { return s.size(); }, schunk);
			if (have != width * height)
				throw std::invalid_argument(detail::text("Invalid schunk passed to compressed::channel constructor. Expected a size of ", width * height, " but instead got ", have));
			m_Schunk = std::make_shared<blosc2::schunk_var<T>>(std::move(schunk));
			m_Adopted = true;
			make_contexts(block_size());
		}

		static channel full(size_t width, size_t height, T fill_value, enums::codec compression_codec = enums::codec::lz4,
			uint8_t compression_level = 9, size_t block_size = s_default_blocksize, size_t chunk_size = s_default_chunksize)
		{
			const size_t aligned = util::align_chunk_to_scanlines_bytes<T>(width, chunk_size);
			return channel(blosc2::schunk_var<T>(blosc2::lazy_schunk<T>(fill_value, width * height, block_size, aligned)), width, height, compression_codec, compression_level);
		}
		static channel zeros(size_t width, size_t height, enums::codec compression_codec = enums::codec::lz4,
			uint8_t compression_level = 9, size_t block_size = s_default_blocksize, size_t chunk_size = s_default_chunksize)
		{
			return full(width, height, T{}, compression_codec, compression_level, block_size, chunk_size);
		}
		static channel full_like(const channel& other, T fill_value)
		{
			return full(other.width(), other.height(), fill_value, other.compression(), other.compression_level(), other.block_size(), other.chunk_size());
		}
		static channel zeros_like(const channel& other) { return full_like(other, T{}); }

		iterator begin() { require_encoder(); return iterator(m_Schunk, m_CompressionContext.get(), m_DecompressionContext.get(), 0, m_Width, m_Height); }
		iterator end() { require_encoder(); return iterator(m_Schunk, m_CompressionContext.get(), m_DecompressionContext.get(), num_chunks(), m_Width, m_Height); }

		blosc2::context_raw_ptr compression_context() { return m_CompressionContext.get(); }
		blosc2::context_raw_ptr decompression_context() { return m_DecompressionContext.get(); }

		/// Rebuilds the contexts (possibly with another block size), as the reference does.  The thread count
		/// itself is meaningless here -- the GPU is the parallelism -- and is only remembered.
		void update_nthreads(size_t nthreads, size_t block_size = s_default_blocksize)
		{
			m_Nthreads = nthreads;
			make_contexts(block_size);
		}

		size_t width() const noexcept { return m_Width; }
		size_t height() const noexcept { return m_Height; }
		enums::codec compression() const noexcept { return m_Codec; }
		uint8_t compression_level() const noexcept { return m_CompressionLevel; }

		size_t compressed_bytes() const { return visit([](auto& s) { return s.csize(); }); }
		size_t uncompressed_size() const { return visit([](auto& s) { return s.size(); }); }
		size_t num_chunks() const { return visit([](auto& s) { return s.num_chunks(); }); }
		size_t block_size() const { return visit([](auto& s) { return s.max_block_size(); }); }
		size_t chunk_size() const { return visit([](auto& s) { return s.chunk_bytes(); }); }
		size_t chunk_elems() const { return chunk_size() / sizeof(T); }
		size_t chunk_size(size_t chunk_index) const { return visit([&](auto& s) { return s.chunk_bytes(chunk_index); }); }
		size_t chunk_elems(size_t chunk_index) const { return chunk_size(chunk_index) / sizeof(T); }

		void get_chunk(std::span<T> buffer, size_t chunk_idx) const
		{
			visit([&](auto& s) { s.chunk(m_DecompressionContext.get(), buffer, chunk_idx); return 0; });
		}
		void set_chunk(std::span<T> buffer, size_t chunk_idx)
		{
			require();
			require_encoder();
			std::visit([&](auto& s) { s.set_chunk(m_CompressionContext, buffer, chunk_idx); }, *m_Schunk);
		}
		/// Decode into caller-owned memory (uncompressed_size() elements): no intermediate vector, no zero fill.
		void decompress_into(std::span<T> out) const
		{
			require();
			if (out.size() != uncompressed_size())
				throw std::invalid_argument(detail::text("decompress_into: buffer holds ", out.size(), " elements, channel has ", uncompressed_size()));
			std::vector<blosc2::batch::target> work;
			std::visit([&](const auto& table) { table.plan_decode(out.data(), work); }, *m_Schunk);
			blosc2::batch::decompress(work);
		}
		std::vector<T> get_decompressed() const
		{
			require();
			return std::visit([&](const auto& s) { return s.to_uncompressed(const_cast<blosc2::context_ptr&>(m_DecompressionContext)); }, *m_Schunk);
		}

		/// The chunk table itself (used by image<T> to batch across channels).
		blosc2::schunk_var<T>& chunks() { require(); return *m_Schunk; }
		const blosc2::schunk_var<T>& chunks() const { require(); return *m_Schunk; }

		bool operator==(const channel<T>& other) const noexcept { return this == &other; }

	private:
		blosc2::schunk_var_ptr<T> m_Schunk = nullptr;
		enums::codec m_Codec = enums::codec::lz4;
		size_t m_Nthreads = std::thread::hardware_concurrency() / 2;
		blosc2::context_ptr m_CompressionContext = nullptr;
		blosc2::context_ptr m_DecompressionContext = nullptr;
		uint8_t m_CompressionLevel = 9;
		bool m_Adopted = false;
		size_t m_Width = 1;
		size_t m_Height = 1;

		// All four codecs of the reference have an encoder on this path since round 3 (lz4hc / zstd: format-valid, not the CPU
		// libraries' bytes -- enums.h), so an adopted chunk table of any of them can be rewritten like one built from pixels.
		void make_contexts(size_t block_size)
		{
			m_CompressionContext = blosc2::create_compression_context<T>(m_Nthreads, m_Codec, m_CompressionLevel, block_size);
			m_DecompressionContext = blosc2::create_decompression_context(m_Nthreads);
		}
		void require_encoder() const
		{
			if (!m_CompressionContext)
				throw std::runtime_error("Internal Error: Channel instance has no compression context");
		}
		void require() const
		{
			if (!m_Schunk)
				throw std::runtime_error("Internal Error: Channel instance is not properly initialized, unable to access its data");
		}
		template <typename F> auto visit(F&& f) const
		{
			require();
			return std::visit(std::forward<F>(f), const_cast<const blosc2::schunk_var<T>&>(*m_Schunk));
		}
	};
}
