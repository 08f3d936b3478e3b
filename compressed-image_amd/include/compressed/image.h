// image.h -- compressed::image<T>: a set of equally sized compressed channels with names and metadata
// (reference compressed/image.h: constructors :87-325, add/remove/extract :940-1118, statistics
// :1122-1170, accessors :1192-1440, get_decompressed :1307-1315).
//
// Construction from planar pixels and get_decompressed() batch ACROSS channels: all chunks of all
// channels go to the MI355X engine in one call (the reference loops channels serially, image.h:119-159).
// Out of scope in this build: every read() overload and read_oiio_metadata (OpenImageIO is absent; what the read path does
// with the scanlines once it has them -- deinterleave, compress -- is image::from_interleaved) and
// JSON metadata (nlohmann-json is absent; metadata is an ordered string map here).
#pragma once
#include <cstddef>
#include <iostream>
#include <map>
#include <optional>
#include <span>
#include <stdexcept>
#include <string>
#include <string_view>
#include <vector>

#include "channel.h"
#include "constants.h"
#include "enums.h"
#include "macros.h"
#include "util.h"

namespace NAMESPACE_COMPRESSED_IMAGE
{
	using metadata_map = std::map<std::string, std::string>;

	template <typename T>
	struct image
	{
		image() = default;
		image(image&&) = default;
		image& operator=(image&&) = default;
		image(const image&) = delete;
		image& operator=(const image&) = delete;

		/// Compress planar channels (each width * height elements).
		image(std::vector<std::span<const T>> channels, size_t width, size_t height, std::vector<std::string> channel_names = {},
			enums::codec compression_codec = enums::codec::lz4, size_t compression_level = 9,
			size_t block_size = s_default_blocksize, size_t chunk_size = s_default_chunksize)
			: m_Width(width), m_Height(height)
		{
			adopt_names(channel_names, channels.size());
			const uint8_t level = util::ensure_compression_level(compression_level);
			for (const auto& c : channels)
				if (c.size() != width * height)
					throw std::runtime_error(detail::text("Invalid channel data passed. Expected its size to match up to width * height (", width, " * ", height,
						") which would be ", width * height, ". Instead received ", c.size()));
			if (channels.empty()) return;

			// one engine call for every chunk of every channel
			const size_t aligned = util::align_chunk_to_scanlines_bytes<T>(width, chunk_size);
			auto cctx = blosc2::create_compression_context<T>(1, compression_codec, level, block_size);
			std::vector<blosc2::batch::piece> pieces;
			std::vector<size_t> first(channels.size() + 1, 0);
			for (size_t ch = 0; ch < channels.size(); ++ch)
			{
				const auto* base = reinterpret_cast<const std::byte*>(channels[ch].data());
				const size_t total = channels[ch].size() * sizeof(T);
				for (size_t off = 0; off < total; off += aligned) pieces.push_back({ base + off, std::min(aligned, total - off) });
				first[ch + 1] = pieces.size();
			}
			auto chunks = blosc2::batch::compress(cctx.get(), pieces, aligned);
			for (size_t ch = 0; ch < channels.size(); ++ch)
			{
				blosc2::schunk<T> table(block_size, aligned);
				for (size_t i = first[ch]; i < first[ch + 1]; ++i) table.append_chunk(std::move(chunks[i]));
				m_Channels.push_back(compressed::channel<T>(blosc2::schunk_var<T>(std::move(table)), width, height, compression_codec, level));
			}
		}

		image(std::vector<std::vector<T>> channels, size_t width, size_t height, std::vector<std::string> channel_names = {},
			enums::codec compression_codec = enums::codec::lz4, size_t compression_level = 9,
			size_t block_size = s_default_blocksize, size_t chunk_size = s_default_chunksize)
			: image(as_spans(channels), width, height, std::move(channel_names), compression_codec, compression_level, block_size, chunk_size) {}

		/// Compress channels that arrive INTERLEAVED (R G B A R G B A ..., what an image reader hands out per block of
		/// scanlines; the reference's read path deinterleaves them on the host before compressing, image.h:1555-1880 with
		/// image_algo::deinterleave at :1880).  Not a constructor of the reference -- its read path is tied to OpenImageIO --
		/// but the same work: the interleaved pixels are uploaded once, split into planes on the device and compressed
		/// from there, all channels in one engine call.
		static image from_interleaved(std::span<const T> interleaved, size_t width, size_t height, size_t nchannels, std::vector<std::string> channel_names = {},
			enums::codec compression_codec = enums::codec::lz4, size_t compression_level = 9,
			size_t block_size = s_default_blocksize, size_t chunk_size = s_default_chunksize)
		{
			if (nchannels == 0 || interleaved.size() != width * height * nchannels)
				throw std::runtime_error(detail::text("Invalid interleaved data passed. Expected its size to match up to width * height * channels (", width, " * ", height,
					" * ", nchannels, "). Instead received ", interleaved.size()));
			image out;
			out.m_Width = width; out.m_Height = height;
			out.adopt_names(channel_names, nchannels);
			const uint8_t level = util::ensure_compression_level(compression_level);
			const size_t aligned = util::align_chunk_to_scanlines_bytes<T>(width, chunk_size);
			auto cctx = blosc2::create_compression_context<T>(1, compression_codec, level, block_size);
			const size_t npixels = width * height, total = npixels * sizeof(T);
			const size_t stride = blosc2::batch::planar_stride(npixels, sizeof(T));
			std::vector<blosc2::batch::planar_piece> pieces;
			std::vector<size_t> first(nchannels + 1, 0);
			for (size_t ch = 0; ch < nchannels; ++ch)
			{
				for (size_t off = 0; off < total; off += aligned) pieces.push_back({ ch * stride + off, std::min(aligned, total - off) });
				first[ch + 1] = pieces.size();
			}
			auto chunks = blosc2::batch::compress_interleaved(cctx.get(), reinterpret_cast<const std::byte*>(interleaved.data()), nchannels, npixels, pieces, aligned);
			for (size_t ch = 0; ch < nchannels; ++ch)
			{
				blosc2::schunk<T> table(block_size, aligned);
				for (size_t i = first[ch]; i < first[ch + 1]; ++i) table.append_chunk(std::move(chunks[i]));
				out.m_Channels.push_back(compressed::channel<T>(blosc2::schunk_var<T>(std::move(table)), width, height, compression_codec, level));
			}
			return out;
		}

		/// Adopt already compressed channels.
		image(std::vector<compressed::channel<T>> channels, size_t width, size_t height, std::vector<std::string> channel_names = {})
			: m_Width(width), m_Height(height)
		{
			adopt_names(channel_names, channels.size());
			for (auto& c : channels)
			{
				check_dims(c.width(), c.height(), "");
				m_Channels.push_back(std::move(c));
			}
		}

		// ---- channel management ----------------------------------------------------------------------------
		void add_channel(compressed::channel<T> _channel, std::optional<std::string> name = std::nullopt)
		{
			check_dims(_channel.width(), _channel.height(), name.value_or(""));
			push_name(name);
			m_Channels.push_back(std::move(_channel));
		}
		/// note the reference's default level for THIS overload is 5, not 9 (image.h:999)
		void add_channel(std::span<const T> data, size_t width, size_t height, std::optional<std::string> name = std::nullopt,
			enums::codec compression_codec = enums::codec::lz4, uint8_t compression_level = 5)
		{
			check_dims(width, height, name.value_or(""));
			push_name(name);
			m_Channels.push_back(compressed::channel<T>(data, width, height, compression_codec, compression_level));
		}

		void remove_channel(size_t index) { (void)extract_channel(index); }
		void remove_channel(const std::string_view name) { (void)extract_channel(name); }

		compressed::channel<T> extract_channel(size_t index)
		{
			if (index >= m_Channels.size()) throw std::out_of_range("Channel index out of range");
			auto out = std::move(m_Channels[index]);
			m_Channels.erase(m_Channels.begin() + static_cast<std::ptrdiff_t>(index));
			if (index < m_ChannelNames.size()) m_ChannelNames.erase(m_ChannelNames.begin() + static_cast<std::ptrdiff_t>(index));
			return out;
		}
		compressed::channel<T> extract_channel(const std::string_view name) { return extract_channel(get_channel_offset(name)); }

		compressed::channel<T>& channel(size_t index)
		{
			if (index >= m_Channels.size())
				throw std::out_of_range(detail::text("Channel index ", index, " is out of range for an image with ", m_Channels.size(), " channels"));
			return m_Channels[index];
		}
		compressed::channel<T>& channel(const std::string_view name) { return m_Channels[get_channel_offset(name)]; }
		std::vector<compressed::channel<T>>& channels() { return m_Channels; }
		const std::vector<compressed::channel<T>>& channels() const { return m_Channels; }

		size_t get_channel_offset(const std::string_view channelname) const
		{
			for (size_t i = 0; i < m_ChannelNames.size(); ++i) if (m_ChannelNames[i] == channelname) return i;
			throw std::invalid_argument(detail::text("Unknown channelname '", channelname, "' encountered"));
		}

		// ---- pixels -------------------------------------------------------------------------------------------
		/// All channels, decompressed -- one engine call for every chunk of the image.
		std::vector<std::vector<T>> get_decompressed() const
		{
			std::vector<std::vector<T>> out(m_Channels.size());
			std::vector<blosc2::batch::target> work;
			for (size_t ch = 0; ch < m_Channels.size(); ++ch)
			{
				out[ch].resize(m_Channels[ch].uncompressed_size());
				std::visit([&](const auto& table) { table.plan_decode(out[ch].data(), work); }, m_Channels[ch].chunks());
			}
			blosc2::batch::decompress(work);
			return out;
		}

		/// All channels into caller-owned memory (num_channels * height * width elements, channel-major): no
		/// intermediate vectors, no zero fill -- what the Python binding uses to decode straight into a numpy array.
		void decompress_into(std::span<T> out) const
		{
			size_t need = 0;
			for (const auto& c : m_Channels) need += c.uncompressed_size();
			if (out.size() != need)
				throw std::invalid_argument(detail::text("decompress_into: buffer holds ", out.size(), " elements, image has ", need));
			std::vector<blosc2::batch::target> work;
			size_t at = 0;
			for (const auto& c : m_Channels)
			{
				std::visit([&](const auto& table) { table.plan_decode(out.data() + at, work); }, c.chunks());
				at += c.uncompressed_size();
			}
			blosc2::batch::decompress(work);
		}

		// ---- statistics ----------------------------------------------------------------------------------------
		void print_statistics()
		{
			size_t csize = 0, usize = 0, nchunks = 0;
			for (const auto& c : m_Channels) { csize += c.compressed_bytes(); usize += c.uncompressed_size() * sizeof(T); nchunks += c.num_chunks(); }
			std::cout << "Statistics for image buffer:\n Width: " << m_Width << "\n Height: " << m_Height << "\n Channels: " << m_Channels.size()
				<< "\n Channelnames: [";
			for (size_t i = 0; i < m_ChannelNames.size(); ++i) std::cout << (i ? ", " : "") << m_ChannelNames[i];
			std::cout << "]\n --------------\n Compressed Size: " << csize << "\n Uncompressed Size: " << usize << "\n Compression ratio: "
				<< static_cast<double>(usize) / static_cast<double>(csize ? csize : 1) << "x\n Num Chunks: " << nchunks << std::endl;
		}
		/// uncompressed / compressed bytes; both sums start at 1 as in the reference (image.h:1162-1163)
		double compression_ratio() const noexcept
		{
			size_t csize = 1, usize = 1;
			for (const auto& c : m_Channels) { csize += c.compressed_bytes(); usize += c.uncompressed_size() * sizeof(T); }
			return static_cast<double>(usize) / static_cast<double>(csize);
		}

		// ---- plain accessors ---------------------------------------------------------------------------------------
		size_t width() const noexcept { return m_Width; }
		size_t height() const noexcept { return m_Height; }
		size_t num_channels() const noexcept { return m_Channels.size(); }
		std::vector<std::string> channelnames() const noexcept { return m_ChannelNames; }
		void channelnames(std::vector<std::string> names)
		{
			if (names.size() != m_Channels.size())
				throw std::invalid_argument(detail::text("Invalid number of arguments received for setting channelnames. Expected vector size to be exactly ",
					m_Channels.size(), " but instead got ", names.size()));
			m_ChannelNames = std::move(names);
		}
		void metadata(const metadata_map& m) noexcept { m_Metadata = m; }
		metadata_map& metadata() noexcept { return m_Metadata; }
		const metadata_map& metadata() const noexcept { return m_Metadata; }

		void update_nthreads(size_t nthreads) { for (auto& c : m_Channels) c.update_nthreads(nthreads, c.block_size()); }
		size_t chunk_size() const
		{
			if (m_Channels.empty()) throw std::runtime_error("Unable to get chunk size from image without channels");
			return m_Channels.front().chunk_size();
		}
		size_t block_size() const
		{
			if (m_Channels.empty()) throw std::runtime_error("Unable to get block size from image without channels");
			return m_Channels.front().block_size();
		}

	private:
		std::vector<compressed::channel<T>> m_Channels{};
		std::vector<std::string> m_ChannelNames{};
		metadata_map m_Metadata{};
		size_t m_Width = 1;
		size_t m_Height = 1;

		static std::vector<std::span<const T>> as_spans(const std::vector<std::vector<T>>& v)
		{
			std::vector<std::span<const T>> out;
			for (const auto& c : v) out.emplace_back(c.data(), c.size());
			return out;
		}
		void adopt_names(const std::vector<std::string>& names, size_t nchannels)
		{
			if (names.size() != nchannels && !names.empty())
				std::cout << "Invalid number of channel names received, expected " << nchannels << " but instead got " << names.size()
					<< ". Ignoring channel names" << std::endl;
			else
				m_ChannelNames = names;
		}
		void push_name(const std::optional<std::string>& name)
		{
			if (name.has_value() && m_ChannelNames.size() == m_Channels.size()) m_ChannelNames.push_back(*name);
			else if (!m_ChannelNames.empty()) m_ChannelNames.push_back(name.value_or(""));
		}
		void check_dims(size_t width, size_t height, const std::string& name) const
		{
			if (width != m_Width)
				throw std::invalid_argument(detail::text("Cannot add channel '", name, "' to the image as its width does not match that of the image. Expected ",
					m_Width, " pixels but instead got ", width, " pixels"));
			if (height != m_Height)
				throw std::invalid_argument(detail::text("Cannot add channel '", name, "' to the image as its height does not match that of the image. Expected ",
					m_Height, " pixels but instead got ", height, " pixels"));
		}
	};
}
