// blosc2/schunk.h -- host bookkeeping of a channel's chunks: `schunk<T>` (every chunk compressed) and
// `lazy_schunk<T>` (a chunk is either compressed bytes or a single fill value), with the public
// surface of the reference's compressed/blosc2/{schunk.h, lazyschunk.h, schunk_mixin.h}.
//
// What is different from the reference is the shape of the hot loops: the constructor from pixels
// (reference schunk.h:65-105, one blosc2_compress_ctx per chunk, serially) and to_uncompressed
// (schunk.h:123-141 / lazyschunk.h:176-194, one blosc2_decompress_ctx per chunk on one thread) hand
// ALL chunks to the GPU engine in one batched call (blosc2/wrapper.h: batch::compress / decompress).
// Both flavours share one implementation: a table of slots {bytes | fill value, element count}.
#pragma once
#include <algorithm>
#include <cstddef>
#include <span>
#include <stdexcept>
#include <variant>
#include <vector>

#include "../constants.h"
#include "../macros.h"
#include "../util.h"
#include "wrapper.h"

namespace NAMESPACE_COMPRESSED_IMAGE
{
	namespace blosc2
	{
		namespace detail
		{
			template <typename T>
			struct lazy_chunk
			{
				std::variant<byte_buffer, T> value;
				size_t num_elements = 0;
				size_t byte_size() const noexcept { return num_elements * sizeof(T); }
				bool is_lazy() const noexcept { return std::holds_alternative<T>(value); }
				const byte_buffer& bytes() const { return std::get<byte_buffer>(value); }
			};

			// Lazy == false: slots always hold bytes and the element count is read from the chunk header
			// (as the reference's schunk does); Lazy == true: slots may hold a fill value.
			template <typename T, bool Lazy>
			struct chunk_table
			{
				using slot = lazy_chunk<T>;

				chunk_table() = default;

				// ---- geometry ------------------------------------------------------------------------------------
				size_t chunk_bytes() const { return m_ChunkSize; }
				size_t chunk_bytes(size_t index) const
				{
					validate_chunk_index(index);
					return m_Chunks[index].num_elements * sizeof(T);
				}
				size_t chunk_elements() const { return elements_of(chunk_bytes()); }
				size_t chunk_elements(size_t index) const { return elements_of(chunk_bytes(index)); }
				size_t num_chunks() const noexcept { return m_Chunks.size(); }
				size_t max_chunk_size() const { return m_ChunkSize; }
				size_t max_block_size() const { return m_BlockSize; }
				size_t byte_size() const noexcept { return size() * sizeof(T); }

				// total elements
				size_t size() const noexcept
				{
					size_t n = 0;
					for (const auto& c : m_Chunks) n += c.num_elements;
					return n;
				}
				// total stored bytes (a lazy slot costs sizeof(T), reference lazyschunk.h csize())
				size_t csize() const noexcept
				{
					size_t n = 0;
					for (const auto& c : m_Chunks) n += c.is_lazy() ? sizeof(T) : c.bytes().size();
					return n;
				}

				// ---- export to a blosc2 super-chunk (reference schunk.h:107-121, lazyschunk.h:122-172) -----------
				schunk_ptr to_schunk()
				{
					schunk_ptr out = create_default_schunk();
					std::vector<std::byte> filler;        // one compressed chunk of the fill value, lz4 level 9 as the reference does
					for (auto& c : m_Chunks)
					{
						if (!c.is_lazy())
						{
							blosc2_schunk_append_chunk(out.get(), reinterpret_cast<uint8_t*>(const_cast<std::byte*>(c.bytes().data())), true);
							continue;
						}
						if (filler.empty())
						{
							std::vector<T> pixels(chunk_elements(), std::get<T>(c.value));
							auto ctx = create_compression_context<T>(out, 1, enums::codec::lz4, 9, m_BlockSize);
							filler.resize(min_compressed_size(m_ChunkSize));
							const size_t n = blosc2::compress<T>(ctx.get(), std::span<const T>(pixels), std::span<std::byte>(filler));
							filler.resize(n);
						}
						blosc2_schunk_append_chunk(out.get(), reinterpret_cast<uint8_t*>(filler.data()), true);
					}
					return out;
				}

				// ---- decode ---------------------------------------------------------------------------------------
				// all chunks -> one vector, ONE engine call for every compressed chunk
				std::vector<T> to_uncompressed(context_ptr& /*decompression_ctx*/) const
				{
					std::vector<T> pixels(size(), fill_value());
					std::vector<batch::target> work;
					work.reserve(m_Chunks.size());
					size_t offset = 0;
					for (const auto& c : m_Chunks)
					{
						if (!c.is_lazy())
							work.push_back({ c.bytes().data(), reinterpret_cast<std::byte*>(pixels.data() + offset), c.num_elements * sizeof(T), c.bytes().size() });
						offset += c.num_elements;
					}
					batch::decompress(work);
					return pixels;
				}

				/// Batch building block: fill the lazy chunks of `pixels` (size() elements) and queue the decode of
				/// every compressed chunk, so that a caller can decode MANY tables with one engine call.
				void plan_decode(T* pixels, std::vector<batch::target>& work) const
				{
					size_t offset = 0;
					for (const auto& c : m_Chunks)
					{
						if (c.is_lazy()) std::fill(pixels + offset, pixels + offset + c.num_elements, std::get<T>(c.value));
						else work.push_back({ c.bytes().data(), reinterpret_cast<std::byte*>(pixels + offset), c.num_elements * sizeof(T), c.bytes().size() });
						offset += c.num_elements;
					}
				}

				/// Same for the chunks [first, first + count): `pixels` receives them back to back.
				void plan_decode_range(T* pixels, size_t first, size_t count, std::vector<batch::target>& work) const
				{
					size_t offset = 0;
					for (size_t i = first; i < first + count; ++i)
					{
						validate_chunk_index(i);
						const auto& c = m_Chunks[i];
						if (c.is_lazy()) std::fill(pixels + offset, pixels + offset + c.num_elements, std::get<T>(c.value));
						else work.push_back({ c.bytes().data(), reinterpret_cast<std::byte*>(pixels + offset), c.num_elements * sizeof(T), c.bytes().size() });
						offset += c.num_elements;
					}
				}

				std::vector<T> chunk(context_ptr& ctx, size_t index) const { return chunk(ctx.get(), index); }
				std::vector<T> chunk(context_raw_ptr ctx, size_t index) const
				{
					validate_chunk_index(index);
					std::vector<T> pixels(chunk_elements(index));
					chunk(ctx, std::span<T>(pixels), index);
					return pixels;
				}
				void chunk(context_ptr& ctx, std::span<T> buffer, size_t index) const { chunk(ctx.get(), buffer, index); }
				void chunk(context_raw_ptr ctx, std::span<T> buffer, size_t index) const
				{
					validate_chunk_index(index);
					const auto& c = m_Chunks[index];
					if (buffer.size() < c.num_elements)
						throw std::invalid_argument(compressed::detail::text("Unable to decompress chunk at idx ", index,
							" into buffer as the buffer needs to at least have the size ", c.num_elements, ". Instead got ", buffer.size()));
					if (c.is_lazy())
						std::fill(buffer.begin(), buffer.end(), std::get<T>(c.value));
					else
						blosc2::decompress(ctx, buffer, std::span<const std::byte>(c.bytes().data(), c.bytes().size()));
				}

				// ---- replace / append -----------------------------------------------------------------------------
				void set_chunk(std::vector<std::byte> compressed, size_t index)
				{
					validate_chunk_index(index);
					const size_t n = chunk_num_elements<T>(compressed);
					m_Chunks[index].value = byte_buffer(std::move(compressed));
					m_Chunks[index].num_elements = n;
					validate_chunk_sizes();
				}
				void set_chunk(byte_buffer compressed, size_t index)
				{
					validate_chunk_index(index);
					const size_t n = chunk_num_elements<T>(std::span<const std::byte>(compressed.data(), compressed.size()));
					m_Chunks[index].value = std::move(compressed);
					m_Chunks[index].num_elements = n;
					validate_chunk_sizes();
				}
				void set_chunk(std::span<const std::byte> compressed, size_t index)
				{
					set_chunk(std::vector<std::byte>(compressed.begin(), compressed.end()), index);
				}
				void set_chunk(context_ptr& compression_ctx, std::span<T> uncompressed, size_t index)
				{
					validate_chunk_index(index);
					util::default_init_vector<std::byte> scratch(min_compressed_size(m_ChunkSize));
					const size_t n = blosc2::compress<T>(compression_ctx.get(), std::span<const T>(uncompressed.data(), uncompressed.size()), std::span<std::byte>(scratch.data(), scratch.size()));
					m_Chunks[index].value = byte_buffer(std::vector<std::byte>(scratch.begin(), scratch.begin() + n));
					m_Chunks[index].num_elements = uncompressed.size();
					validate_chunk_sizes();
				}

				void append_chunk(std::vector<std::byte> compressed)
				{
					const size_t n = chunk_num_elements<T>(compressed);
					m_Chunks.push_back(slot{ byte_buffer(std::move(compressed)), n });
					validate_chunk_sizes();
				}
				void append_chunk(byte_buffer compressed)
				{
					const size_t n = chunk_num_elements<T>(std::span<const std::byte>(compressed.data(), compressed.size()));
					m_Chunks.push_back(slot{ std::move(compressed), n });
					validate_chunk_sizes();
				}
				void append_chunk(context_ptr& compression_ctx, std::span<T> uncompressed)
				{
					util::default_init_vector<std::byte> scratch(min_compressed_size(m_ChunkSize));
					append_chunk(compression_ctx, uncompressed, std::span<std::byte>(scratch.data(), scratch.size()));
				}
				void append_chunk(context_ptr& compression_ctx, std::span<T> uncompressed, std::span<std::byte> compression_buff)
				{
					if (compression_buff.size() < min_compressed_size(m_ChunkSize))
						throw std::runtime_error(compressed::detail::text("Error while appending chunk to super-chunk. Expected compression buffer to be at least ",
							min_compressed_size(m_ChunkSize), " bytes but instead we got ", compression_buff.size(), " bytes"));
					const size_t n = blosc2::compress<T>(compression_ctx.get(), std::span<const T>(uncompressed.data(), uncompressed.size()), compression_buff);
					m_Chunks.push_back(slot{ byte_buffer(std::vector<std::byte>(compression_buff.begin(), compression_buff.begin() + n)), uncompressed.size() });
					validate_chunk_sizes();
				}
				// many chunks at once (one engine call): what image::read-style producers should use
				void append_chunks(context_ptr& compression_ctx, std::span<const T> pixels)
				{
					auto made = compress_pieces(compression_ctx.get(), pixels);
					for (auto& m : made) m_Chunks.push_back(std::move(m));
					validate_chunk_sizes();
				}

			protected:
				std::vector<slot> m_Chunks{};
				size_t m_ChunkSize = s_default_chunksize;
				size_t m_BlockSize = s_default_blocksize;

				static size_t elements_of(size_t bytes)
				{
					if (bytes % sizeof(T) != 0)
						throw std::runtime_error(compressed::detail::text("Internal Error: The chunk byte size is not cleanly divisible by the sizeof T."
							" Chunk size is ", bytes, " while sizeof(T) is ", sizeof(T)));
					return bytes / sizeof(T);
				}

				T fill_value() const noexcept
				{
					for (const auto& c : m_Chunks) if (c.is_lazy()) return std::get<T>(c.value);
					return T{};
				}

				void validate_chunk_index(size_t index) const
				{
					if (index >= m_Chunks.size())
						throw std::out_of_range(compressed::detail::text("Cannot access index ", index, " in schunk. Total amount of chunks is ", m_Chunks.size()));
				}

				// all chunks but the last hold exactly m_ChunkSize bytes; the last at most that (reference schunk_mixin.h:216-246)
				void validate_chunk_sizes() const
				{
					if (m_Chunks.empty()) return;
					for (size_t i = 0; i + 1 < m_Chunks.size(); ++i)
						if (m_Chunks[i].byte_size() != m_ChunkSize)
							throw std::invalid_argument(compressed::detail::text("Error while validating chunk sizes; Expected all chunks to have a size equivalent to ",
								m_ChunkSize, " (m_ChunkSize). However, chunk ", i, " instead has a chunk size of ", m_Chunks[i].byte_size(),
								". Having a size different from the rest of the chunks is only supported for the last chunk (blosc2 limitation)."));
					if (m_Chunks.back().byte_size() > m_ChunkSize)
						throw std::runtime_error(compressed::detail::text("Error while validating chunk sizes; Expected the last chunk to be at most ",
							m_ChunkSize, " bytes, instead got ", m_Chunks.back().byte_size(), " bytes."));
				}

				// cut `pixels` into full chunks + remainder (reference schunk.h:79-104) and compress them in one call
				std::vector<slot> compress_pieces(context_raw_ptr cctx, std::span<const T> pixels) const
				{
					const size_t total = pixels.size() * sizeof(T);
					const auto* base = reinterpret_cast<const std::byte*>(pixels.data());
					std::vector<batch::piece> pieces;
					for (size_t off = 0; off < total; off += m_ChunkSize)
						pieces.push_back({ base + off, std::min(m_ChunkSize, total - off) });
					auto bytes = batch::compress(cctx, pieces, m_ChunkSize);
					std::vector<slot> out;
					out.reserve(bytes.size());
					for (size_t i = 0; i < bytes.size(); ++i)
						out.push_back(slot{ std::move(bytes[i]), pieces[i].nbytes / sizeof(T) });
					return out;
				}
			};
		} // detail

		/// Every chunk compressed (reference: compressed/blosc2/schunk.h).
		template <typename T>
		struct schunk final : public detail::chunk_table<T, false>
		{
			schunk() = default;
			/// empty table with a geometry, to be filled with append_chunk (reference schunk.h:50-55)
			schunk(size_t block_size, size_t chunk_size)
			{
				util::validate_chunk_size<T>(chunk_size, "schunk");
				this->m_ChunkSize = chunk_size;
				this->m_BlockSize = block_size;
			}
			/// compress `data` (reference schunk.h:65-105); one batched engine call for all chunks
			schunk(std::span<const T> data, size_t block_size, size_t chunk_size, context_ptr& compression_ctx)
			{
				util::validate_chunk_size<T>(chunk_size, "schunk");
				this->m_BlockSize = block_size;
				this->m_ChunkSize = chunk_size;
				this->m_Chunks = this->compress_pieces(compression_ctx.get(), data);
			}
		};

		/// Chunks start as a single fill value and become real on set_chunk (reference: compressed/blosc2/lazyschunk.h).
		template <typename T>
		struct lazy_schunk final : public detail::chunk_table<T, true>
		{
			lazy_schunk() = default;
			lazy_schunk(T value, size_t num_elements, size_t block_size, size_t chunk_size)
			{
				util::validate_chunk_size<T>(chunk_size, "lazy_schunk");
				this->m_BlockSize = block_size;
				this->m_ChunkSize = chunk_size;
				const size_t total = num_elements * sizeof(T);
				for (size_t off = 0; off < total; off += chunk_size)
					this->m_Chunks.push_back(detail::lazy_chunk<T>{ value, std::min(chunk_size, total - off) / sizeof(T) });
			}
		};

		template <typename T> using schunk_var = std::variant<schunk<T>, lazy_schunk<T>>;
		template <typename T> using schunk_var_ptr = std::shared_ptr<schunk_var<T>>;
	} // blosc2
} // NAMESPACE_COMPRESSED_IMAGE
