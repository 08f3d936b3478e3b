#pragma once
#include "macros.h"
namespace NAMESPACE_COMPRESSED_IMAGE
{
	template <typename T> struct channel;
	template <typename T> struct image;
}
