// detail/config.h -- the small fixed points of the public surface in one place: the namespace macro, the profiling
// hooks (no-ops here), the default chunk / block geometry and the forward declarations.  The reference spreads
// them over compressed/macros.h, constants.h and fwd.h; those names exist here too and simply include this file.
#pragma once
#include <cstddef>

#define NAMESPACE_COMPRESSED_IMAGE compressed

// the reference's chrome-tracing hooks (detail/scoped_timer.h:24-30) compile to nothing here: kernel time comes from
// HIP events / rocprofv3 (cimg_engine_kernel_time)
#define _COMPRESSED_PROFILE_FUNCTION()
#define _COMPRESSED_PROFILE_SCOPE(name)

namespace NAMESPACE_COMPRESSED_IMAGE
{
	// default geometry of the reference (compressed/constants.h:9,11)
	inline constexpr std::size_t s_default_chunksize = 4'194'304;   // 4 MiB per chunk = one batch item
	inline constexpr std::size_t s_default_blocksize = 32'768;      // 32 KiB per block = one GPU work item

	template <typename T> struct channel;
	template <typename T> struct image;
}
