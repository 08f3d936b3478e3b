// blosc2/wrapper.h -- host-side seam onto the codec.  Same names and error behaviour as the reference's
// compressed/blosc2/wrapper.h (codec mapping :74-119, compress :134-223, decompress :236-277,
// context factories :313-413, size helpers :416-468), served by libcimg_hip.so through
// include/blosc2.h, plus the *batched* helpers the re-shaped chunk loops use (include/cimg_hip.h).
#pragma once
#include <cstddef>
#include <cstdint>
#include <limits>
#include <memory>
#include <mutex>
#include <span>
#include <stdexcept>
#include <string>
#include <vector>

#include "blosc2.h"
#include "cimg_hip.h"
#include "../detail/pinned_pool.h"

#include "../macros.h"
#include "../enums.h"
#include "../detail/text.h"

namespace NAMESPACE_COMPRESSED_IMAGE
{
	namespace blosc2
	{
		inline std::string map_error_code(int rc) { return std::string(print_error(rc)); }   // reference blosc2/util.h:16-19

		template <typename S> struct deleter;
		template <> struct deleter<blosc2_schunk> { void operator()(blosc2_schunk* s) const { blosc2_schunk_free(s); } };
		template <> struct deleter<blosc2_context> { void operator()(blosc2_context* c) const { blosc2_free_ctx(c); } };

		using schunk_ptr = std::unique_ptr<blosc2_schunk, deleter<blosc2_schunk>>;
		using schunk_raw_ptr = blosc2_schunk*;
		using chunk_raw_ptr = void*;
		using context_ptr = std::unique_ptr<blosc2_context, deleter<blosc2_context>>;
		using context_raw_ptr = blosc2_context*;

		inline uint8_t codec_to_blosc2(enums::codec c)
		{
			switch (c)
			{
			case enums::codec::lz4: return BLOSC_LZ4;
			case enums::codec::lz4hc: return BLOSC_LZ4HC;
			case enums::codec::zstd: return BLOSC_ZSTD;
			default: return BLOSC_BLOSCLZ;
			}
		}
		inline enums::codec blosc2_to_codec(uint8_t c)
		{
			switch (c)
			{
			case BLOSC_LZ4: return enums::codec::lz4;
			case BLOSC_LZ4HC: return enums::codec::lz4hc;
			case BLOSC_ZSTD: return enums::codec::zstd;
			default: return enums::codec::blosclz;
			}
		}

		// ---- single chunk (one blosc2_*_ctx call, host pointers) ----------------------------------------------
		template <typename T>
		size_t compress(context_raw_ptr context, std::span<const T> data, std::span<std::byte> chunk)
		{
			const int cbytes = blosc2_compress_ctx(context, data.data(), static_cast<int32_t>(data.size() * sizeof(T)),
				chunk.data(), static_cast<int32_t>(chunk.size()));
			if (cbytes < 0)
				throw std::runtime_error(detail::text("Unable to compress context using Blosc2 with error code ", cbytes));
			return static_cast<size_t>(cbytes);
		}
		template <typename T> size_t compress(context_raw_ptr context, std::span<T> data, std::span<std::byte> chunk) requires(!std::is_const_v<T>)
		{
			return compress<T>(context, std::span<const T>(data.data(), data.size()), chunk);
		}
		template <typename T> size_t compress(context_ptr& context, std::span<T> data, std::span<std::byte> chunk) { return compress<std::remove_const_t<T>>(context.get(), std::span<const std::remove_const_t<T>>(data.data(), data.size()), chunk); }

		template <typename T>
		size_t decompress(context_raw_ptr context, std::span<T> buffer, std::span<const std::byte> chunk)
		{
			if (buffer.size() * sizeof(T) > static_cast<size_t>(std::numeric_limits<int32_t>::max()))
				throw std::out_of_range(detail::text("Blosc2 chunk size may not exceed numeric limit of int32_t, got ", buffer.size() * sizeof(T), " which would exceed that"));
			const int n = blosc2_decompress_ctx(context, chunk.data(), std::numeric_limits<int32_t>::max(), buffer.data(),
				static_cast<int32_t>(buffer.size() * sizeof(T)));
			if (n < 0)
				throw std::runtime_error(detail::text("Error code ", n, " while decompressing blosc2 chunk"));
			return static_cast<size_t>(n);
		}
		template <typename T> size_t decompress(context_ptr& context, std::span<T> buffer, std::span<const std::byte> chunk) { return decompress(context.get(), buffer, chunk); }

		inline size_t append_chunk(schunk_ptr& schunk, std::span<std::byte> chunk)
		{
			const auto n = blosc2_schunk_append_chunk(schunk.get(), reinterpret_cast<uint8_t*>(chunk.data()), true);
			if (n < 0)
				throw std::runtime_error(detail::text("Unable to append chunk into super-chunk with the following blosc2 error code ", n));
			return static_cast<size_t>(n);
		}

		inline schunk_ptr create_default_schunk()
		{
			auto cparams = BLOSC2_CPARAMS_DEFAULTS;
			auto dparams = BLOSC2_DPARAMS_DEFAULTS;
			blosc2_storage storage = BLOSC2_STORAGE_DEFAULTS;
			storage.cparams = &cparams;
			storage.dparams = &dparams;
			return schunk_ptr(blosc2_schunk_new(&storage));
		}

		// ---- contexts -----------------------------------------------------------------------------------------
		template <typename T>
		blosc2_cparams create_blosc2_cparams(size_t nthreads, enums::codec codec, uint8_t compression_level, size_t block_size)
		{
			if (nthreads > static_cast<size_t>(std::numeric_limits<int16_t>::max()))
				throw std::out_of_range(detail::text("Number of threads may not exceed ", std::numeric_limits<int16_t>::max(), ", got ", nthreads));
			blosc2_cparams p = BLOSC2_CPARAMS_DEFAULTS;
			p.blocksize = static_cast<int32_t>(block_size);
			p.typesize = sizeof(T);
			p.splitmode = BLOSC_AUTO_SPLIT;
			p.clevel = compression_level;
			p.nthreads = static_cast<int16_t>(nthreads == 0 ? 1 : nthreads);    // accepted, ignored: the parallelism is the GPU's
			p.compcode = codec_to_blosc2(codec);
			// All four codecs of the reference construct (enums.h:18-24).  lz4 and blosclz chunks are the reference's bytes; lz4hc
			// and zstd chunks are FORMAT-VALID -- every LZ4 / zstd decoder, c-blosc2 included, reads them -- but not liblz4-HC's /
			// libzstd's bytes (csrc/plan.h, csrc/zstd_encode.h).
			return p;
		}
		template <typename T>
		context_ptr create_compression_context(size_t nthreads, enums::codec codec, uint8_t compression_level, size_t block_size)
		{
			return context_ptr(blosc2_create_cctx(create_blosc2_cparams<T>(nthreads, codec, compression_level, block_size)));
		}
		template <typename T>
		context_ptr create_compression_context(schunk_ptr& schunk, size_t nthreads, enums::codec codec, uint8_t compression_level, size_t block_size)
		{
			auto p = create_blosc2_cparams<T>(nthreads, codec, compression_level, block_size);
			p.schunk = schunk.get();
			return context_ptr(blosc2_create_cctx(p));
		}
		inline context_ptr create_decompression_context(size_t nthreads)
		{
			if (nthreads > static_cast<size_t>(std::numeric_limits<int16_t>::max()))
				throw std::out_of_range(detail::text("Number of threads may not exceed ", std::numeric_limits<int16_t>::max(), ", got ", nthreads));
			auto d = BLOSC2_DPARAMS_DEFAULTS;
			d.nthreads = 1;
			return context_ptr(blosc2_create_dctx(d));
		}
		inline context_ptr create_decompression_context(schunk_ptr& schunk, size_t nthreads)
		{
			auto ctx = create_decompression_context(nthreads);
			(void)schunk;
			return ctx;
		}

		// ---- sizes -----------------------------------------------------------------------------------------------
		template <size_t ChunkSize> constexpr size_t min_compressed_size() { return ChunkSize + BLOSC2_MAX_OVERHEAD; }
		inline constexpr size_t min_compressed_size(size_t chunk_size) { return chunk_size + BLOSC2_MAX_OVERHEAD; }
		template <size_t ChunkSize> constexpr size_t min_decompressed_size() { return ChunkSize; }
		inline constexpr size_t min_decompressed_size(size_t chunk_size) { return chunk_size; }

		inline size_t chunk_num_bytes(const std::byte* chunk)
		{
			int32_t nbytes{}, cbytes{}, blocksize{};
			const int rc = blosc2_cbuffer_sizes(chunk, &nbytes, &cbytes, &blocksize);
			if (rc < 0)
				throw std::runtime_error(detail::text("Unable to find buffer sizes due to blosc2 error: ", map_error_code(rc)));
			return static_cast<size_t>(nbytes);
		}
		template <typename T> size_t chunk_num_elements(const std::vector<std::byte>& chunk) { return chunk_num_bytes(chunk.data()) / sizeof(T); }
		template <typename T> size_t chunk_num_elements(std::span<const std::byte> chunk) { return chunk_num_bytes(chunk.data()) / sizeof(T); }

		// ---- batches: what replaces the reference's serial chunk loops --------------------------------------------
		/// The bytes of one compressed chunk: immutable, cheap to move, either a view into the arena a whole batch of
		/// chunks was fetched into (shared ownership) or a vector adopted from the caller.
		class byte_buffer
		{
		public:
			byte_buffer() = default;
			byte_buffer(std::vector<std::byte> bytes)
			{
				auto held = std::make_shared<std::vector<std::byte>>(std::move(bytes));
				m_Data = held->data();
				m_Size = held->size();
				m_Owner = std::move(held);
			}
			byte_buffer(std::shared_ptr<const void> owner, const std::byte* data, size_t size) : m_Owner(std::move(owner)), m_Data(data), m_Size(size) {}

			const std::byte* data() const noexcept { return m_Data; }
			size_t size() const noexcept { return m_Size; }
			bool empty() const noexcept { return m_Size == 0; }
			const std::byte* begin() const noexcept { return m_Data; }
			const std::byte* end() const noexcept { return m_Data + m_Size; }
			const std::byte& operator[](size_t i) const noexcept { return m_Data[i]; }

		private:
			std::shared_ptr<const void> m_Owner;
			const std::byte* m_Data = nullptr;
			size_t m_Size = 0;
		};

		namespace batch
		{
			inline cimg_engine* engine()
			{
				cimg_engine* e = cimg_shared_engine();
				if (!e)
					throw std::runtime_error(detail::text("compressed-image MI355X engine unavailable: ", cimg_last_error(nullptr)));
				return e;
			}

			struct piece { const std::byte* data; size_t nbytes; };          // one chunk's pixels (host memory)

			// Compress many chunks in one engine call.  Every chunk gets destsize = nominal chunk size +
			// BLOSC2_MAX_OVERHEAD, as the reference passes it (schunk.h:73, :200, :225).
			inline std::vector<byte_buffer> compress(context_raw_ptr cctx, const std::vector<piece>& pieces, size_t nominal_chunk_bytes)
			{
				std::vector<byte_buffer> out(pieces.size());
				if (pieces.empty()) return out;
				cimg_cparams cp;
				int rc = cimg_context_cparams(cctx, &cp);
				if (rc < 0) throw std::runtime_error(detail::text("Unable to compress context using Blosc2 with error code ", rc));
				const size_t n = pieces.size();
				std::vector<int64_t> raw_off(n), comp_off(n);
				std::vector<int32_t> nbytes(n), destsize(n), cbytes(n);
				const std::byte* base = pieces[0].data;
				for (const auto& p : pieces) if (p.data < base) base = p.data;
				for (size_t i = 0; i < n; ++i)
				{
					if (pieces[i].nbytes > static_cast<size_t>(BLOSC2_MAX_BUFFERSIZE))
						throw std::out_of_range(detail::text("Blosc2 chunk size may not exceed numeric limit of int32_t, got ", pieces[i].nbytes));
					raw_off[i] = pieces[i].data - base;
					nbytes[i] = static_cast<int32_t>(pieces[i].nbytes);
					destsize[i] = static_cast<int32_t>(min_compressed_size(nominal_chunk_bytes));
				}
				// _begin and _fetch belong together: the engine's own (recursive) lock is held across the pair, so no other
				// user of the shared engine -- batch::decompress, the blosc2_*_ctx shim, another thread's compress -- can
				// reuse its staging area in between
				struct engine_guard
				{
					cimg_engine* e;
					explicit engine_guard(cimg_engine* e_) : e(e_) { cimg_engine_lock(e); }
					~engine_guard() { cimg_engine_unlock(e); }
				} pair_lock(engine());
				// ONE call: upload, compress and bring the chunks back group by group -- as soon as the sizes of a group are known the
				// engine asks for a block of exactly that size (recycled, page-locked: detail/pinned_pool.h) and sends the group's
				// chunks there while the next group is being compressed; the chunks are views into those blocks -- no staging area of
				// nominal size, no second copy.  (Round 3 fetched all chunks behind the last group: 4.2 against 3.0 ms for an image of
				// 4 x 4096^2 float16.)
				struct arenas_t
				{
					std::vector<std::pair<std::shared_ptr<std::byte>, size_t>> blocks;
					static void* take(void* user, size_t bytes)
					{
						auto* self = static_cast<arenas_t*>(user);
						try
						{
							self->blocks.emplace_back(NAMESPACE_COMPRESSED_IMAGE::detail::pinned_pool::get().arena(bytes), bytes);
						}
						catch (...) { return nullptr; }
						return self->blocks.back().first.get();
					}
				} arenas;
				std::vector<void*> where(n, nullptr);
				rc = cimg_compress_batch_host_packed(engine(), &cp, static_cast<int32_t>(n), base, raw_off.data(), nbytes.data(), destsize.data(), cbytes.data(),
					&arenas_t::take, &arenas, where.data());
				if (rc < 0)
					throw std::runtime_error(detail::text("Unable to compress context using Blosc2 with error code ", rc, " (", cimg_last_error(engine()), ")"));
				for (size_t i = 0; i < n; ++i)
				{
					if (cbytes[i] <= 0 || !where[i])
						throw std::runtime_error(detail::text("Unable to compress context using Blosc2 with error code ", cbytes[i]));
					std::byte* at = static_cast<std::byte*>(where[i]);
					const std::shared_ptr<std::byte>* owner = nullptr;
					for (const auto& blk : arenas.blocks)
						if (at >= blk.first.get() && at < blk.first.get() + blk.second) { owner = &blk.first; break; }
					if (!owner)
						throw std::runtime_error("compressed chunk arrived outside the blocks handed to the engine");
					out[i] = byte_buffer(*owner, at, static_cast<size_t>(cbytes[i]));
				}
				return out;
			}

			// The same for pixels that arrive INTERLEAVED (R G B A R G B A ...) in host memory: uploaded once, split into planes on
			// the device (the reference's image_algo::deinterleave, a host loop between reading and compressing, image.h:1880),
			// compressed from there.  A piece names its bytes in the PLANAR layout: channel c occupies
			// [c * plane_stride, c * plane_stride + npixels * typesize), plane_stride = that size rounded up to 16.
			struct planar_piece { size_t offset; size_t nbytes; };
			inline size_t planar_stride(size_t npixels, size_t typesize) { return (npixels * typesize + 15) & ~size_t{ 15 }; }
			inline std::vector<byte_buffer> compress_interleaved(context_raw_ptr cctx, const std::byte* interleaved, size_t nchannels, size_t npixels,
				const std::vector<planar_piece>& pieces, size_t nominal_chunk_bytes)
			{
				std::vector<byte_buffer> out(pieces.size());
				if (pieces.empty()) return out;
				cimg_cparams cp;
				int rc = cimg_context_cparams(cctx, &cp);
				if (rc < 0) throw std::runtime_error(detail::text("Unable to compress context using Blosc2 with error code ", rc));
				const size_t n = pieces.size();
				std::vector<int64_t> raw_off(n), comp_off(n);
				std::vector<int32_t> nbytes(n), destsize(n), cbytes(n);
				for (size_t i = 0; i < n; ++i)
				{
					if (pieces[i].nbytes > static_cast<size_t>(BLOSC2_MAX_BUFFERSIZE))
						throw std::out_of_range(detail::text("Blosc2 chunk size may not exceed numeric limit of int32_t, got ", pieces[i].nbytes));
					raw_off[i] = static_cast<int64_t>(pieces[i].offset);
					nbytes[i] = static_cast<int32_t>(pieces[i].nbytes);
					destsize[i] = static_cast<int32_t>(min_compressed_size(nominal_chunk_bytes));
				}
				struct engine_guard
				{
					cimg_engine* e;
					explicit engine_guard(cimg_engine* e_) : e(e_) { cimg_engine_lock(e); }
					~engine_guard() { cimg_engine_unlock(e); }
				} pair_lock(engine());
				rc = cimg_compress_batch_host_interleaved_begin(engine(), &cp, static_cast<int32_t>(nchannels), static_cast<int64_t>(npixels), interleaved,
					static_cast<int32_t>(n), raw_off.data(), nbytes.data(), destsize.data(), cbytes.data());
				if (rc < 0)
					throw std::runtime_error(detail::text("Unable to compress context using Blosc2 with error code ", rc, " (", cimg_last_error(engine()), ")"));
				size_t total = 0;
				for (size_t i = 0; i < n; ++i)
				{
					if (cbytes[i] <= 0)
						throw std::runtime_error(detail::text("Unable to compress context using Blosc2 with error code ", cbytes[i]));
					comp_off[i] = static_cast<int64_t>(total);
					total += (static_cast<size_t>(cbytes[i]) + 63) & ~size_t{ 63 };
				}
				std::shared_ptr<std::byte> arena = NAMESPACE_COMPRESSED_IMAGE::detail::pinned_pool::get().arena(total);
				rc = cimg_compress_batch_host_fetch(engine(), static_cast<int32_t>(n), arena.get(), comp_off.data());
				if (rc < 0)
					throw std::runtime_error(detail::text("Unable to compress context using Blosc2 with error code ", rc, " (", cimg_last_error(engine()), ")"));
				for (size_t i = 0; i < n; ++i)
					out[i] = byte_buffer(arena, arena.get() + comp_off[i], static_cast<size_t>(cbytes[i]));
				return out;
			}

			// one chunk -> its pixels; chunk_bytes = what the chunk buffer really holds (0: unknown, trust the header)
			struct target { const std::byte* chunk; std::byte* out; size_t capacity; size_t chunk_bytes = 0; };

			inline void decompress(const std::vector<target>& items)
			{
				if (items.empty()) return;
				const size_t n = items.size();
				const std::byte* cbase = items[0].chunk;
				std::byte* rbase = items[0].out;
				for (const auto& t : items) { if (t.chunk < cbase) cbase = t.chunk; if (t.out < rbase) rbase = t.out; }
				std::vector<int64_t> comp_off(n), raw_off(n);
				std::vector<int32_t> cap(n), status(n), held(n);
				bool sized = true;
				for (size_t i = 0; i < n; ++i)
				{
					sized = sized && items[i].chunk_bytes > 0 && items[i].chunk_bytes <= static_cast<size_t>(std::numeric_limits<int32_t>::max());
					held[i] = static_cast<int32_t>(items[i].chunk_bytes);
					if (items[i].capacity > static_cast<size_t>(std::numeric_limits<int32_t>::max()))
						throw std::out_of_range(detail::text("Blosc2 chunk size may not exceed numeric limit of int32_t, got ", items[i].capacity, " which would exceed that"));
					comp_off[i] = items[i].chunk - cbase;
					raw_off[i] = items[i].out - rbase;
					cap[i] = static_cast<int32_t>(items[i].capacity);
				}
				const int rc = cimg_decompress_batch_host_sized(engine(), static_cast<int32_t>(n), cbase, comp_off.data(), sized ? held.data() : nullptr,
					rbase, raw_off.data(), cap.data(), status.data());
				if (rc < 0)
					throw std::runtime_error(detail::text("Error code ", rc, " while decompressing blosc2 chunk"));
			}
		}
	}
}
