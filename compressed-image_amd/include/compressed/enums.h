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
			blosclz,   // GPU encode + decode (bytes of BloscLZ 2.3.0, csrc/blosclz_kernel.h)
			lz4,       // GPU encode + decode (the reference's default, channel.h:101)
			lz4hc,     // decode only (codec format 1 = LZ4 blocks); compressing fails with BLOSC2_ERROR_CODEC_SUPPORT
			zstd       // decode only (codec format 4, csrc/zstd_kernel.h: a slow path); compressing fails with BLOSC2_ERROR_CODEC_SUPPORT
		};
	}
}
