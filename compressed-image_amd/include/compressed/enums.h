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
			lz4hc,     // codec format 1 (LZ4 blocks).  Written here by the FAST match finder at acceleration 1: valid for every LZ4 decoder, not LZ4_compress_HC's bytes
			zstd       // codec format 4 (one zstd frame per stream).  Written by csrc/zstd_encode.h: valid for every zstd decoder, not libzstd's bytes; read by csrc/zstd_kernel.h
		};
	}
}
