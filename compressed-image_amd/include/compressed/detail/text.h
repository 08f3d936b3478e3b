// detail/text.h -- tiny message builder; the reference formats its exception texts with std::format,
// which libstdc++ 11 (this image) does not ship.
#pragma once
#include <sstream>
#include <string>
#include "../macros.h"
namespace NAMESPACE_COMPRESSED_IMAGE
{
	namespace detail
	{
		template <typename... Args>
		std::string text(const Args&... parts)
		{
			std::ostringstream os;
			(os << ... << parts);
			return os.str();
		}
	}
}
