// half.h -- a 16-bit IEEE binary16 storage type.  The reference uses Imath::half for float16 channels
// (python/bindings/util/npy_half.h:17-59); Imath is not available here and the codec path treats pixels
// as opaque bytes anyway, so a POD with exact bit semantics is all that is needed.
#pragma once
#include <cstdint>
#include <cstring>
#include "macros.h"

namespace NAMESPACE_COMPRESSED_IMAGE
{
	struct half
	{
		uint16_t bits = 0;

		half() = default;
		/// round-to-nearest-even conversion from float
		explicit half(float f)
		{
			uint32_t x;
			std::memcpy(&x, &f, 4);
			const uint32_t sign = (x >> 16) & 0x8000u;
			const int32_t exp = static_cast<int32_t>((x >> 23) & 0xFF) - 127 + 15;
			uint32_t man = x & 0x7FFFFFu;
			if (((x >> 23) & 0xFF) == 0xFF) { bits = static_cast<uint16_t>(sign | 0x7C00u | (man ? 0x200u : 0)); return; }   // inf / nan
			if (exp >= 31) { bits = static_cast<uint16_t>(sign | 0x7C00u); return; }                                              // overflow
			if (exp <= 0)
			{
				if (exp < -10) { bits = static_cast<uint16_t>(sign); return; }
				man |= 0x800000u;
				const uint32_t shift = static_cast<uint32_t>(14 - exp);
				uint32_t h = man >> shift;
				const uint32_t rem = man & ((1u << shift) - 1), halfway = 1u << (shift - 1);
				if (rem > halfway || (rem == halfway && (h & 1))) ++h;
				bits = static_cast<uint16_t>(sign | h);
				return;
			}
			uint32_t h = (static_cast<uint32_t>(exp) << 10) | (man >> 13);
			const uint32_t rem = man & 0x1FFFu;
			if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) ++h;
			bits = static_cast<uint16_t>(sign | h);
		}
		explicit half(double d) : half(static_cast<float>(d)) {}
		explicit half(int i) : half(static_cast<float>(i)) {}

		explicit operator float() const
		{
			const uint32_t sign = static_cast<uint32_t>(bits & 0x8000u) << 16;
			uint32_t exp = (bits >> 10) & 0x1F, man = bits & 0x3FFu, x;
			if (exp == 0)
			{
				if (man == 0) x = sign;
				else { int e = -1; do { ++e; man <<= 1; } while (!(man & 0x400u)); x = sign | static_cast<uint32_t>(127 - 15 - e) << 23 | (man & 0x3FFu) << 13; }
			}
			else if (exp == 31) x = sign | 0x7F800000u | man << 13;
			else x = sign | (exp + 127 - 15) << 23 | man << 13;
			float f;
			std::memcpy(&f, &x, 4);
			return f;
		}
		bool operator==(const half& o) const noexcept { return bits == o.bits; }
		bool operator!=(const half& o) const noexcept { return bits != o.bits; }
	};
	static_assert(sizeof(half) == 2, "half must be two bytes");
}
