// detail/pinned_pool.h -- recycled page-locked host buffers (not in the reference: its codec is on the CPU).
//
// Everything the host mirror allocates as the *destination of a device-to-host copy* comes from here: the arena
// a batch of compressed chunks lives in, and the pixel buffers the Python binding returns.  Two measured reasons
// (tools/diag_pymodule2.py): a page-locked destination is filled by DMA at PCIe speed, and a recycled buffer has
// no first-touch page faults -- decoding 64 MiB into a fresh numpy array took 22 ms, into a recycled page-locked
// buffer 3.6 ms.  Live page-locked memory is budgeted (CIMG_PINNED_LIMIT_MB, default 8192); past the budget the
// pool hands out ordinary memory, which is slower but always works.
#pragma once
#include <atomic>
#include <cstddef>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <new>
#include <vector>

#include "cimg_hip.h"
#include "../macros.h"

namespace NAMESPACE_COMPRESSED_IMAGE
{
	namespace detail
	{
		class pinned_pool
		{
		public:
			struct block { void* p = nullptr; size_t cap = 0; bool pinned = false; };

			static pinned_pool& get()
			{
				static pinned_pool* pool = new pinned_pool();     // never destroyed: buffers may outlive static teardown
				return *pool;
			}

			block take(size_t bytes)
			{
				bytes = bytes ? bytes : 1;
				{
					std::lock_guard<std::mutex> lk(m_Mutex);
					size_t best = m_Idle.size();
					for (size_t i = 0; i < m_Idle.size(); ++i)
						if (m_Idle[i].cap >= bytes && m_Idle[i].cap <= 2 * bytes + 4096 && (best == m_Idle.size() || m_Idle[i].cap < m_Idle[best].cap)) best = i;
					if (best != m_Idle.size())
					{
						block b = m_Idle[best];
						m_Idle.erase(m_Idle.begin() + static_cast<std::ptrdiff_t>(best));
						m_IdleBytes -= b.cap;
						return b;
					}
				}
				const size_t cap = (bytes + 4095) & ~size_t{ 4095 };
				if (m_LivePinned.load() + cap <= m_Limit)
				{
					if (void* p = cimg_host_malloc(cap))
					{
						m_LivePinned += cap;
						return { p, cap, true };
					}
				}
				void* p = std::malloc(cap);
				if (!p) throw std::bad_alloc();
				return { p, cap, false };
			}

			void give(block b) noexcept
			{
				if (!b.p) return;
				if (b.pinned)
				{
					std::lock_guard<std::mutex> lk(m_Mutex);
					if (m_IdleBytes + b.cap <= s_IdleLimit) { m_Idle.push_back(b); m_IdleBytes += b.cap; return; }
				}
				if (b.pinned) { m_LivePinned -= b.cap; cimg_host_free(b.p); }
				else std::free(b.p);
			}

			/// `bytes` of storage owned by the returned pointer; released to the pool when the last copy dies
			std::shared_ptr<std::byte> arena(size_t bytes)
			{
				const block b = take(bytes);
				return std::shared_ptr<std::byte>(static_cast<std::byte*>(b.p), [b](std::byte*) { pinned_pool::get().give(b); });
			}

		private:
			pinned_pool()
			{
				if (const char* s = std::getenv("CIMG_PINNED_LIMIT_MB")) m_Limit = static_cast<size_t>(std::strtoull(s, nullptr, 10)) << 20;
			}
			static constexpr size_t s_IdleLimit = size_t{ 1 } << 30;
			std::mutex m_Mutex;
			std::vector<block> m_Idle;
			size_t m_IdleBytes = 0;
			std::atomic<size_t> m_LivePinned{ 0 };
			size_t m_Limit = size_t{ 8192 } << 20;
		};
	}
}
