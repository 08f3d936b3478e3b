// util.h -- chunk geometry helpers with the reference's names and error behaviour
// (compressed/util.h:22-57 default_init_vector, :59-84 validate_chunk_size, :86-94
// ensure_compression_level, :115-184 scanline / tile alignment).
#pragma once
#include <cstddef>
#include <cstdint>
#include <iostream>
#include <memory>
#include <span>
#include <stdexcept>
#include <string_view>
#include <vector>
#include "macros.h"
#include "detail/text.h"

namespace NAMESPACE_COMPRESSED_IMAGE
{
	namespace util
	{
		namespace detail
		{
			// allocator adaptor whose value-less construct() default-initialises (no zero fill)
			template <typename T, typename Base = std::allocator<T>>
			struct no_init_allocator : Base
			{
				using Base::Base;
				template <typename U> struct rebind { using other = no_init_allocator<U, typename std::allocator_traits<Base>::template rebind_alloc<U>>; };
				template <typename U> void construct(U* p) noexcept(std::is_nothrow_default_constructible_v<U>) { ::new (static_cast<void*>(p)) U; }
				template <typename U, typename... A> void construct(U* p, A&&... a) { std::allocator_traits<Base>::construct(static_cast<Base&>(*this), p, std::forward<A>(a)...); }
			};
		}
		template <typename T> using default_init_vector = std::vector<T, detail::no_init_allocator<T>>;

		template <typename T> constexpr bool ensure_chunk_size(size_t chunk_size) noexcept { return chunk_size % sizeof(T) == 0; }

		template <typename T> void validate_chunk_size(size_t chunk_size, std::string_view context)
		{
			if (!ensure_chunk_size<T>(chunk_size))
				throw std::invalid_argument(compressed::detail::text(context, ": bad chunk size received, expected it to be cleanly divisible by ",
					sizeof(T), " but instead got ", chunk_size));
		}

		inline uint8_t ensure_compression_level(size_t compression_level)
		{
			if (compression_level > 9)
			{
				std::cout << "Blosc2 only supports compression levels from 0-9, truncating value to this" << std::endl;
				compression_level = 9;
			}
			return static_cast<uint8_t>(compression_level);
		}

		template <typename T> std::span<const T> as_const_span(std::span<T> data) { return std::span<const T>(data.data(), data.size()); }

		// whole scanlines only: floor(chunk_size / sizeof(T) / width) rows
		template <typename T> size_t align_chunk_to_scanlines_elems(size_t width, size_t chunk_size)
		{
			const size_t rows = width ? chunk_size / sizeof(T) / width : 0;
			if (rows == 0)
				throw std::runtime_error(compressed::detail::text("Unable to align chunk size to scanlines as the size of a scanline exceeds the chunk size."
					" Got a scanline size of ", width, " x ", sizeof(T), " (sizeof(T)) while the max size of the chunks is ", chunk_size));
			return rows * width;
		}
		template <typename T> size_t align_chunk_to_scanlines_bytes(size_t width, size_t chunk_size) { return align_chunk_to_scanlines_elems<T>(width, chunk_size) * sizeof(T); }

		template <typename T> size_t align_chunk_to_tile_elems(size_t width, size_t tile_height, size_t chunk_size)
		{
			const size_t row_bytes = sizeof(T) * width;
			if (row_bytes > chunk_size)
				throw std::runtime_error(compressed::detail::text("Scanline size (", row_bytes, ") exceeds chunk size (", chunk_size, ")."));
			const size_t rows = (chunk_size / row_bytes / tile_height) * tile_height;
			if (rows == 0)
				throw std::runtime_error(compressed::detail::text("Chunk size (", chunk_size, ") is too small to fit even one tile (", tile_height,
					" scanlines, ", row_bytes, " bytes per scanline)."));
			return rows * width;
		}
		template <typename T> size_t align_chunk_to_tile_bytes(size_t width, size_t tile_height, size_t chunk_size) { return align_chunk_to_tile_elems<T>(width, tile_height, chunk_size) * sizeof(T); }

		template <typename T> constexpr bool is_aligned_to_scanlines(size_t byte_size, size_t width) { return byte_size % (width * sizeof(T)) == 0; }
	}
}
