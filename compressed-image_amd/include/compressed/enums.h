// enums.h -- codec selector (reference: compressed/enums.h:18-24).  The OIIO TypeDesc helpers of the
// reference are out of scope (no OpenImageIO in this build).
#pragma once
#include "macros.h"
namespace NAMESPACE_COMPRESSED_IMAGE
{
	namespace enums
	{
		enum class codec
		{
			blosclz,   // not built on the GPU path yet: requests fail with BLOSC2_ERROR_CODEC_SUPPORT
			lz4,       // the GPU path
			lz4hc,     // not built
			zstd       // not built
		};
	}
}
