// python/module.cpp -- the `compressed_image` Python module: Codec, Channel, Image, mirroring the
// reference's pybind11 surface (python/py_module/compressed_image/_stubs/_compressed_image.pyi:11-306;
// bind_channel.h:17-255, bind_image.h:34-480, bind_enums.h:15-24) on top of the host mirror in
// ../include/compressed.  dtype -> T dispatch over the same nine element types as the reference's
// variant_t.h:92-103.  Out of scope here: Image.read / dtype(s)_from_file (OpenImageIO is absent).
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <memory>
#include <mutex>
#include <optional>
#include <string>
#include <variant>
#include <vector>

#include "compressed/channel.h"
#include "compressed/half.h"
#include "compressed/image.h"

namespace py = pybind11;
using compressed::enums::codec;

namespace
{
	template <typename... Ts> struct type_list {};
	using pixel_types = type_list<compressed::half, float, double, uint8_t, int8_t, uint16_t, int16_t, uint32_t, int32_t>;

	template <typename T> const char* dtype_code();
	template <> const char* dtype_code<compressed::half>() { return "float16"; }
	template <> const char* dtype_code<float>() { return "float32"; }
	template <> const char* dtype_code<double>() { return "float64"; }
	template <> const char* dtype_code<uint8_t>() { return "uint8"; }
	template <> const char* dtype_code<int8_t>() { return "int8"; }
	template <> const char* dtype_code<uint16_t>() { return "uint16"; }
	template <> const char* dtype_code<int16_t>() { return "int16"; }
	template <> const char* dtype_code<uint32_t>() { return "uint32"; }
	template <> const char* dtype_code<int32_t>() { return "int32"; }
	template <typename T> py::dtype np_dtype() { return py::dtype(dtype_code<T>()); }

	py::dtype as_dtype(const py::object& o) { return py::dtype::from_args(o); }

	// call f.template operator()<T>() for the T whose numpy dtype equals `dt`; ValueError otherwise
	template <typename F, typename... Ts>
	auto dispatch(const py::dtype& dt, F&& f, type_list<Ts...>)
	{
		using R = decltype(f.template operator()<uint8_t>());
		std::optional<R> out;
		const bool hit = ((dt.is(np_dtype<Ts>()) ? (out.emplace(f.template operator()<Ts>()), true) : false) || ...);
		if (!hit)
			throw py::value_error("Unsupported dtype '" + std::string(py::str(dt)) + "': supported are float16/32/64, (u)int8/16/32");
		return std::move(*out);
	}
	template <typename F> auto dispatch(const py::dtype& dt, F&& f) { return dispatch(dt, std::forward<F>(f), pixel_types{}); }

	template <typename T> T from_scalar(const py::object& v)
	{
		if constexpr (std::is_same_v<T, compressed::half>) return compressed::half(v.cast<double>());
		else if constexpr (std::is_floating_point_v<T>) return static_cast<T>(v.cast<double>());
		else return static_cast<T>(v.cast<long long>());
	}

	// contiguous view of the array's elements as T (copying only if it is not C-contiguous)
	template <typename T> std::pair<py::array, std::span<const T>> elements(const py::array& a)
	{
		py::array c = py::array::ensure(a, py::array::c_style);
		if (!c) throw py::value_error("array is not convertible to a C-contiguous buffer");
		return { c, std::span<const T>(static_cast<const T*>(c.data()), static_cast<size_t>(c.size())) };
	}
	// Result arrays of get_decompressed() own a recycled page-locked buffer (compressed/detail/pinned_pool.h): the
	// engine's D2H copy lands in it by DMA and there are no first-touch page faults.  Dropping the array returns the
	// buffer to the pool.
	template <typename T> py::array pooled_array(std::vector<py::ssize_t> shape)
	{
		size_t count = 1;
		for (auto d : shape) count *= static_cast<size_t>(d);
		using pool = compressed::detail::pinned_pool;
		auto* s = new pool::block(pool::get().take(std::max<size_t>(count * sizeof(T), 1)));
		py::capsule owner(s, [](void* p) { auto* b = static_cast<pool::block*>(p); pool::get().give(*b); delete b; });
		return py::array(np_dtype<T>(), std::move(shape), s->p, owner);
	}

	template <typename T> py::array to_array(std::vector<T>&& pixels, std::vector<py::ssize_t> shape)
	{
		auto* heap = new std::vector<T>(std::move(pixels));
		py::capsule owner(heap, [](void* p) { delete static_cast<std::vector<T>*>(p); });
		return py::array(np_dtype<T>(), std::move(shape), heap->data(), owner);
	}

	// ---- Channel ------------------------------------------------------------------------------------------
	template <typename T> using chan_ptr = std::shared_ptr<compressed::channel<T>>;
	using any_channel = std::variant<chan_ptr<compressed::half>, chan_ptr<float>, chan_ptr<double>, chan_ptr<uint8_t>, chan_ptr<int8_t>,
		chan_ptr<uint16_t>, chan_ptr<int16_t>, chan_ptr<uint32_t>, chan_ptr<int32_t>>;

	struct Channel
	{
		any_channel impl;

		template <typename F> auto visit(F&& f) const { return std::visit([&](auto& p) { return f(*p); }, impl); }

		static Channel from_array(const py::array& data, size_t width, size_t height, codec c, size_t level, size_t block, size_t chunk)
		{
			return dispatch(data.dtype(), [&]<typename T>() {
				auto [keep, px] = elements<T>(data);
				if (px.size() != width * height)
					throw py::value_error("Channel data has " + std::to_string(px.size()) + " elements, expected width * height = " + std::to_string(width * height));
				return Channel{ std::make_shared<compressed::channel<T>>(px, width, height, c, static_cast<uint8_t>(std::min<size_t>(level, 255)), block, chunk) };
			});
		}
		static Channel full(const py::object& dtype, const py::object& fill, size_t width, size_t height, codec c, size_t level, size_t block, size_t chunk)
		{
			return dispatch(as_dtype(dtype), [&]<typename T>() {
				return Channel{ std::make_shared<compressed::channel<T>>(compressed::channel<T>::full(width, height, from_scalar<T>(fill), c,
					static_cast<uint8_t>(std::min<size_t>(level, 255)), block, chunk)) };
			});
		}
		static Channel full_like(const Channel& other, const py::object& fill)
		{
			return other.visit([&]<typename T>(compressed::channel<T>& ch) {
				return Channel{ std::make_shared<compressed::channel<T>>(compressed::channel<T>::full_like(ch, from_scalar<T>(fill))) };
			});
		}

		py::dtype dtype() const { return visit([]<typename T>(compressed::channel<T>&) { return np_dtype<T>(); }); }
		size_t width() const { return visit([](auto& c) { return c.width(); }); }
		size_t height() const { return visit([](auto& c) { return c.height(); }); }

		py::array get_chunk(size_t index) const
		{
			return visit([&]<typename T>(compressed::channel<T>& ch) {
				if (index >= ch.num_chunks()) throw py::index_error("chunk index " + std::to_string(index) + " out of range, channel has " + std::to_string(ch.num_chunks()) + " chunks");
				std::vector<T> px(ch.chunk_elems(index));
				ch.get_chunk(std::span<T>(px), index);
				return to_array<T>(std::move(px), { static_cast<py::ssize_t>(ch.chunk_elems(index)) });
			});
		}
		py::array get_chunk_into(size_t index, py::array buffer) const
		{
			visit([&]<typename T>(compressed::channel<T>& ch) {
				if (index >= ch.num_chunks()) throw py::index_error("chunk index out of range");
				check_chunk_array<T>(buffer, ch.chunk_elems(index), /*exact=*/false);
				ch.get_chunk(std::span<T>(static_cast<T*>(buffer.mutable_data()), static_cast<size_t>(buffer.size())), index);
				return 0;
			});
			return buffer;
		}
		void set_chunk(size_t index, const py::array& array)
		{
			visit([&]<typename T>(compressed::channel<T>& ch) {
				if (index >= ch.num_chunks()) throw py::index_error("chunk index " + std::to_string(index) + " out of range, channel has " + std::to_string(ch.num_chunks()) + " chunks");
				check_chunk_array<T>(array, ch.chunk_elems(index), /*exact=*/true);
				auto [keep, px] = elements<T>(array);
				std::vector<T> copy(px.begin(), px.end());
				ch.set_chunk(std::span<T>(copy), index);
				return 0;
			});
		}
		py::array get_decompressed() const
		{
			return visit([]<typename T>(compressed::channel<T>& ch) {
				// numpy owns the pixels from the start (np.empty): the engine's D2H copy is the only pass over them
				py::array out = pooled_array<T>({ static_cast<py::ssize_t>(ch.height()), static_cast<py::ssize_t>(ch.width()) });
				ch.decompress_into(std::span<T>(static_cast<T*>(out.mutable_data()), static_cast<size_t>(out.size())));
				return out;
			});
		}

		template <typename T> static void check_chunk_array(const py::array& a, size_t elems, bool exact)
		{
			if (!a.dtype().is(np_dtype<T>())) throw py::value_error("array dtype does not match the channel dtype");
			if (a.ndim() != 1) throw py::value_error("chunk arrays must be one-dimensional, got " + std::to_string(a.ndim()) + " dimensions");
			const size_t n = static_cast<size_t>(a.size());
			if (exact ? n != elems : n < elems)
				throw py::value_error("chunk array has " + std::to_string(n) + " elements, expected " + std::to_string(elems));
		}
	};

	// ---- Image ------------------------------------------------------------------------------------------------
	template <typename T> using img_ptr = std::shared_ptr<compressed::image<T>>;
	using any_image = std::variant<img_ptr<compressed::half>, img_ptr<float>, img_ptr<double>, img_ptr<uint8_t>, img_ptr<int8_t>,
		img_ptr<uint16_t>, img_ptr<int16_t>, img_ptr<uint32_t>, img_ptr<int32_t>>;

	struct Image
	{
		any_image impl;
		py::dict metadata;

		template <typename F> auto visit(F&& f) const { return std::visit([&](auto& p) { return f(p); }, impl); }

		Image(const py::object& dtype, const std::vector<py::array>& channels, size_t width, size_t height, std::vector<std::string> names,
			codec c, size_t level, size_t block, size_t chunk)
		{
			impl = dispatch(as_dtype(dtype), [&]<typename T>() -> any_image {
				std::vector<py::array> keep;
				std::vector<std::span<const T>> spans;
				for (const auto& a : channels)
				{
					if (!a.dtype().is(np_dtype<T>())) throw py::value_error("channel dtype does not match the image dtype");
					auto [k, px] = elements<T>(a);
					keep.push_back(k);
					spans.push_back(px);
				}
				try { return std::make_shared<compressed::image<T>>(spans, width, height, std::move(names), c, level, block, chunk); }
				catch (const std::runtime_error& e) { throw py::value_error(e.what()); }
			});
		}

		void add_channel(const py::array& data, size_t width, size_t height, std::optional<std::string> name, codec c, size_t level, size_t block, size_t chunk)
		{
			visit([&]<typename T>(const img_ptr<T>& img) {
				if (!data.dtype().is(np_dtype<T>())) throw py::value_error("channel dtype does not match the image dtype");
				if (data.ndim() == 2 && (static_cast<size_t>(data.shape(0)) != height || static_cast<size_t>(data.shape(1)) != width))
					throw py::value_error("array shape does not match (height, width)");
				auto [keep, px] = elements<T>(data);
				if (px.size() != width * height) throw py::value_error("array size does not match width * height");
				compressed::channel<T> ch(px, width, height, c, static_cast<uint8_t>(std::min<size_t>(level, 255)), block, chunk);
				img->add_channel(std::move(ch), std::move(name));
				return 0;
			});
		}
		void remove_channel(const std::variant<std::string, size_t>& key)
		{
			visit([&](auto& img) {
				if (std::holds_alternative<size_t>(key)) img->remove_channel(std::get<size_t>(key));
				else img->remove_channel(std::string_view(std::get<std::string>(key)));
				return 0;
			});
		}
		Channel channel(const std::variant<std::string, size_t>& key) const
		{
			return visit([&]<typename T>(const img_ptr<T>& img) {
				auto& ch = std::holds_alternative<size_t>(key) ? img->channel(std::get<size_t>(key)) : img->channel(std::string_view(std::get<std::string>(key)));
				return Channel{ chan_ptr<T>(img, &ch) };          // aliases the image: keeps it alive
			});
		}
		std::vector<Channel> channels() const
		{
			std::vector<Channel> out;
			const size_t n = visit([](auto& img) { return img->num_channels(); });
			for (size_t i = 0; i < n; ++i) out.push_back(channel(i));
			return out;
		}
		py::array get_decompressed() const
		{
			return visit([]<typename T>(const img_ptr<T>& img) {
				py::array out = pooled_array<T>({ static_cast<py::ssize_t>(img->num_channels()),
					static_cast<py::ssize_t>(img->height()), static_cast<py::ssize_t>(img->width()) });
				img->decompress_into(std::span<T>(static_cast<T*>(out.mutable_data()), static_cast<size_t>(out.size())));
				return out;
			});
		}
	};
}

PYBIND11_MODULE(compressed_image, m)
{
	m.doc() = "MI355X-native chunked image compression (drop-in for EmilDohne/compressed-image's Python module)";

	py::enum_<codec>(m, "Codec", py::module_local())
		.value("blosclz", codec::blosclz)
		.value("lz4", codec::lz4)
		.value("lz4hc", codec::lz4hc)
		.value("zstd", codec::zstd)
		.export_values();

	const auto d_block = compressed::s_default_blocksize, d_chunk = compressed::s_default_chunksize;

	py::class_<Channel>(m, "Channel", py::module_local())
		.def(py::init(&Channel::from_array), py::arg("data"), py::arg("width"), py::arg("height"), py::arg("compression_codec") = codec::lz4,
			py::arg("compression_level") = 9, py::arg("block_size") = d_block, py::arg("chunk_size") = d_chunk)
		.def_static("full", &Channel::full, py::arg("dtype"), py::arg("fill_value"), py::arg("width"), py::arg("height"),
			py::arg("compression_codec") = codec::lz4, py::arg("compression_level") = 9, py::arg("block_size") = d_block, py::arg("chunk_size") = d_chunk)
		.def_static("zeros", [](const py::object& dtype, size_t w, size_t h, codec c, size_t level, size_t block, size_t chunk) {
				return Channel::full(dtype, py::int_(0), w, h, c, level, block, chunk); },
			py::arg("dtype"), py::arg("width"), py::arg("height"), py::arg("compression_codec") = codec::lz4, py::arg("compression_level") = 9,
			py::arg("block_size") = d_block, py::arg("chunk_size") = d_chunk)
		.def_static("full_like", &Channel::full_like, py::arg("other"), py::arg("fill_value"))
		.def_static("zeros_like", [](const Channel& other) { return Channel::full_like(other, py::int_(0)); }, py::arg("other"))
		.def_property_readonly("dtype", &Channel::dtype)
		.def_property_readonly("shape", [](const Channel& c) { return py::make_tuple(c.height(), c.width()); })
		.def_property_readonly("width", &Channel::width)
		.def_property_readonly("height", &Channel::height)
		.def("block_size", [](const Channel& c) { return c.visit([](auto& ch) { return ch.block_size(); }); })
		.def("chunk_size", [](const Channel& c) { return c.visit([](auto& ch) { return ch.chunk_size(); }); })
		.def("chunk_size", [](const Channel& c, size_t i) { return c.visit([&](auto& ch) { return ch.chunk_size(i); }); }, py::arg("chunk_index"))
		.def("chunk_elems", [](const Channel& c) { return c.visit([](auto& ch) { return ch.chunk_elems(); }); })
		.def("chunk_elems", [](const Channel& c, size_t i) { return c.visit([&](auto& ch) { return ch.chunk_elems(i); }); }, py::arg("chunk_index"))
		.def("compressed_bytes", [](const Channel& c) { return c.visit([](auto& ch) { return ch.compressed_bytes(); }); })
		.def("uncompressed_size", [](const Channel& c) { return c.visit([](auto& ch) { return ch.uncompressed_size(); }); })
		.def("num_chunks", [](const Channel& c) { return c.visit([](auto& ch) { return ch.num_chunks(); }); })
		.def("compression", [](const Channel& c) { return c.visit([](auto& ch) { return ch.compression(); }); })
		.def("compression_level", [](const Channel& c) { return c.visit([](auto& ch) { return static_cast<size_t>(ch.compression_level()); }); })
		.def("update_nthreads", [](Channel& c, size_t n, size_t block) { c.visit([&](auto& ch) { ch.update_nthreads(n, block); return 0; }); },
			py::arg("nthreads"), py::arg("block_size") = d_block)
		.def("get_chunk", &Channel::get_chunk, py::arg("chunk_index"))
		.def("get_chunk", &Channel::get_chunk_into, py::arg("chunk_index"), py::arg("array"))
		.def("set_chunk", &Channel::set_chunk, py::arg("chunk_index"), py::arg("array"))
		.def("get_decompressed", &Channel::get_decompressed);

	py::class_<Image>(m, "Image", py::module_local())
		.def(py::init<const py::object&, const std::vector<py::array>&, size_t, size_t, std::vector<std::string>, codec, size_t, size_t, size_t>(),
			py::arg("dtype"), py::arg("channels"), py::arg("width"), py::arg("height"), py::arg("channel_names") = std::vector<std::string>{},
			py::arg("compression_codec") = codec::lz4, py::arg("compression_level") = 9, py::arg("block_size") = d_block, py::arg("chunk_size") = d_chunk)
		.def("add_channel", &Image::add_channel, py::arg("data"), py::arg("width"), py::arg("height"), py::arg("name") = std::nullopt,
			py::arg("compression_codec") = codec::lz4, py::arg("compression_level") = 9, py::arg("block_size") = d_block, py::arg("chunk_size") = d_chunk)
		.def("remove_channel", &Image::remove_channel, py::arg("name_or_index"))
		.def("__getitem__", &Image::channel, py::arg("key"))
		.def("__len__", [](const Image& i) { return i.visit([](auto& img) { return img->num_channels(); }); })
		.def("channel", &Image::channel, py::arg("key"))
		.def("channels", &Image::channels)
		.def("get_decompressed", &Image::get_decompressed)
		.def("get_channel_index", [](const Image& i, const std::string& name) { return i.visit([&](auto& img) { return img->get_channel_offset(name); }); }, py::arg("channelname"))
		.def("print_statistics", [](const Image& i) { i.visit([](auto& img) { img->print_statistics(); return 0; }); })
		.def("compression_ratio", [](const Image& i) { return i.visit([](auto& img) { return img->compression_ratio(); }); })
		.def_property_readonly("dtype", [](const Image& i) { return i.visit([]<typename T>(const img_ptr<T>&) { return np_dtype<T>(); }); })
		.def_property_readonly("shape", [](const Image& i) { return i.visit([](auto& img) { return py::make_tuple(img->num_channels(), img->height(), img->width()); }); })
		.def_property_readonly("width", [](const Image& i) { return i.visit([](auto& img) { return img->width(); }); })
		.def_property_readonly("height", [](const Image& i) { return i.visit([](auto& img) { return img->height(); }); })
		.def_property_readonly("num_channels", [](const Image& i) { return i.visit([](auto& img) { return img->num_channels(); }); })
		.def("get_channel_names", [](const Image& i) { return i.visit([](auto& img) { return img->channelnames(); }); })
		.def("set_channel_names", [](Image& i, std::vector<std::string> names) { i.visit([&](auto& img) { img->channelnames(std::move(names)); return 0; }); }, py::arg("channel_names"))
		.def("update_nthreads", [](Image& i, size_t n) { i.visit([&](auto& img) { img->update_nthreads(n); return 0; }); }, py::arg("nthreads"))
		.def("block_size", [](const Image& i) { return i.visit([](auto& img) { return img->block_size(); }); })
		.def("chunk_size", [](const Image& i) { return i.visit([](auto& img) { return img->chunk_size(); }); })
		.def("set_metadata", [](Image& i, py::dict md) { i.metadata = std::move(md); }, py::arg("metadata"))
		.def("get_metadata", [](const Image& i) { return i.metadata; });
}
