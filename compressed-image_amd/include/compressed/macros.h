// macros.h -- namespace macro kept for source compatibility with the reference (compressed/macros.h).
#pragma once
#define NAMESPACE_COMPRESSED_IMAGE compressed
// the reference's chrome-tracing hooks (detail/scoped_timer.h:24-30) compile to nothing here: kernel
// time comes from HIP events / rocprofv3 (cimg_engine_kernel_time)
#define _COMPRESSED_PROFILE_FUNCTION()
#define _COMPRESSED_PROFILE_SCOPE(name)
