// ranges.h -- compressed::ranges::zip: walk several ranges in lock-step (reference compressed/ranges.h:33-105,
// a stand-in for C++23 std::views::zip).  Dereferencing yields a tuple of whatever the underlying
// iterators yield (references for spans, chunk_span values for channels); iteration stops at the
// shortest range.
#pragma once
#include <iterator>
#include <tuple>
#include <utility>
#include "macros.h"

namespace NAMESPACE_COMPRESSED_IMAGE
{
	namespace ranges
	{
		template <typename... Ranges>
		class zip
		{
			std::tuple<Ranges&...> m_Ranges;

		public:
			explicit zip(Ranges&... ranges) : m_Ranges(ranges...) {}

			template <typename... Its>
			struct cursor
			{
				std::tuple<Its...> its;
				cursor& operator++() { std::apply([](auto&... i) { (++i, ...); }, its); return *this; }
				auto operator*() { return std::apply([](auto&... i) { return std::tuple<decltype(*i)...>(*i...); }, its); }
				// "not equal" while NO component has reached its end: stops at the shortest range
				bool operator!=(const cursor& other) const { return all_differ(other, std::index_sequence_for<Its...>{}); }
				bool operator==(const cursor& other) const { return !(*this != other); }
			private:
				template <size_t... I> bool all_differ(const cursor& o, std::index_sequence<I...>) const { return ((std::get<I>(its) != std::get<I>(o.its)) && ...); }
			};

			auto begin() { return std::apply([](auto&... r) { return cursor<decltype(std::begin(r))...>{ { std::begin(r)... } }; }, m_Ranges); }
			auto end() { return std::apply([](auto&... r) { return cursor<decltype(std::end(r))...>{ { std::end(r)... } }; }, m_Ranges); }
		};
		template <typename... Ranges> zip(Ranges&...) -> zip<Ranges...>;
	}
}
