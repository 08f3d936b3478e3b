// constants.h -- default geometry of the reference (compressed/constants.h:9,11).
#pragma once
#include <cstddef>
#include "macros.h"
namespace NAMESPACE_COMPRESSED_IMAGE
{
	inline constexpr std::size_t s_default_chunksize = 4'194'304;   // 4 MiB per chunk
	inline constexpr std::size_t s_default_blocksize = 32'768;      // 32 KiB per block = one GPU work item
}
