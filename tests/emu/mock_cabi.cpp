// tests/emu/mock_cabi.cpp -- TEST INFRASTRUCTURE: the cimg_* engine entry points of include/cimg_hip.h
// implemented on the host lane emulator (tests/emu/emu.cpp), so that the host-side C++ mirror
// (compressed-image_amd/include/compressed) and the pybind11 module can be exercised in a container
// without a GPU.  Linked together with csrc/blosc2_shim.cpp into tests/emu/libcimg_hip_mock.so, which
// only tests load; the product links libcimg_hip.so and has no such path.
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/cimg_hip.h"

extern "C" {
struct EmuCParams { int32_t typesize, clevel, blocksize, compcode, splitmode; uint8_t filters[6], filters_meta[6]; };
int emu_compress_batch(const EmuCParams* p, int nchunks, const uint8_t* raw, const int64_t* raw_off, const int32_t* nbytes,
                       uint8_t* comp, const int64_t* comp_off, const int32_t* destsize, int32_t* cbytes);
int emu_decompress_batch(int nchunks, const uint8_t* comp, const int64_t* comp_off, const int32_t* nbytes, const int32_t* blocksize,
                         uint8_t* raw, const int64_t* raw_off, int32_t* status);
}

struct cimg_engine { std::string err; std::vector<uint8_t> stage; std::vector<int64_t> off; std::vector<int32_t> len; std::recursive_mutex mu; };
#define LOCK_ENGINE(e) std::lock_guard<std::recursive_mutex> lock_((e)->mu)
static std::string g_err;

extern "C" {

void cimg_cparams_init(cimg_cparams* p, int32_t typesize)
{
    memset(p, 0, sizeof(*p));
    p->typesize = typesize; p->clevel = 9; p->blocksize = 32768; p->compcode = 1; p->splitmode = 3; p->filters[5] = 1;
}
int cimg_engine_create(int, cimg_engine** out) { *out = new cimg_engine(); return 0; }
void cimg_engine_destroy(cimg_engine* e) { delete e; }
const char* cimg_last_error(const cimg_engine* e) { return e ? e->err.c_str() : g_err.c_str(); }
int cimg_engine_synchronize(cimg_engine*) { return 0; }
void cimg_engine_lock(cimg_engine* e) { e->mu.lock(); }
void cimg_engine_unlock(cimg_engine* e) { e->mu.unlock(); }
void* cimg_host_malloc(size_t bytes) { return malloc(bytes ? bytes : 16); }
void cimg_host_free(void* p) { free(p); }

int cimg_compress_batch_host_begin(cimg_engine* e, const cimg_cparams* p, int32_t n, const void* h_raw, const int64_t* raw_off,
                                   const int32_t* nbytes, const int32_t* destsize, int32_t* cbytes)
{
    LOCK_ENGINE(e);
    e->off.clear();
    if (n <= 0) return 0;
    EmuCParams ep;
    ep.typesize = p->typesize; ep.clevel = p->clevel; ep.blocksize = p->blocksize; ep.compcode = p->compcode; ep.splitmode = p->splitmode;
    memcpy(ep.filters, p->filters, 6); memcpy(ep.filters_meta, p->filters_meta, 6);
    // the emulated kernels write whole destsize windows: stage like the engine does
    std::vector<int64_t> off((size_t)n);
    int64_t total = 0;
    for (int i = 0; i < n; i++) { off[(size_t)i] = total; total += ((int64_t)destsize[i] + 63) & ~63ll; }
    e->stage.resize((size_t)total + 64);
    const int rc = emu_compress_batch(&ep, n, (const uint8_t*)h_raw, raw_off, nbytes, e->stage.data(), off.data(), destsize, cbytes);
    if (rc < 0) { e->err = "emulated compress batch rejected, code " + std::to_string(rc); return rc; }
    e->off = off;
    e->len.assign(cbytes, cbytes + n);
    return 0;
}

extern "C" int emu_deinterleave(const uint8_t* src, int nch, int ts, int64_t npixels, uint8_t* dst, int64_t plane_stride);
int cimg_deinterleave_device(cimg_engine* e, const void* d_interleaved, int32_t nch, int32_t ts, int64_t npixels, void* d_planar, int64_t plane_stride)
{
    LOCK_ENGINE(e);
    const int rc = emu_deinterleave((const uint8_t*)d_interleaved, nch, ts, npixels, (uint8_t*)d_planar, plane_stride);
    if (rc < 0) e->err = "deinterleave: invalid arguments";
    return rc;
}
int cimg_compress_batch_host_interleaved_begin(cimg_engine* e, const cimg_cparams* p, int32_t nch, int64_t npixels, const void* h_interleaved,
                                               int32_t n, const int64_t* raw_off, const int32_t* nbytes, const int32_t* destsize, int32_t* cbytes)
{
    LOCK_ENGINE(e);
    e->off.clear();
    if (n <= 0) return 0;
    const int64_t plane_stride = (npixels * p->typesize + 15) & ~15ll;
    for (int i = 0; i < n; i++) if (raw_off[i] < 0 || raw_off[i] + nbytes[i] > plane_stride * nch) { e->err = "chunk outside the planes"; return -12; }
    std::vector<uint8_t> planar((size_t)(plane_stride * nch) + 64);
    const int rc = emu_deinterleave((const uint8_t*)h_interleaved, nch, p->typesize, npixels, planar.data(), plane_stride);
    if (rc < 0) { e->err = "deinterleave: invalid arguments"; return rc; }
    return cimg_compress_batch_host_begin(e, p, n, planar.data(), raw_off, nbytes, destsize, cbytes);
}

int cimg_compress_batch_host_fetch(cimg_engine* e, int32_t n, void* h_comp, const int64_t* comp_off)
{
    LOCK_ENGINE(e);
    if (n <= 0) return 0;
    if ((size_t)n != e->off.size()) { e->err = "no compressed batch is waiting to be fetched"; return -12; }
    for (int i = 0; i < n; i++) if (e->len[(size_t)i] > 0) memcpy((uint8_t*)h_comp + comp_off[i], e->stage.data() + e->off[(size_t)i], (size_t)e->len[(size_t)i]);
    e->off.clear();
    return 0;
}

// (the engine cuts a batch into groups and asks for a block per group: here two groups, so that the host mirror meets chunks in
// more than one block)
int cimg_compress_batch_host_packed(cimg_engine* e, const cimg_cparams* p, int32_t n, const void* h_raw, const int64_t* raw_off,
                                    const int32_t* nbytes, const int32_t* destsize, int32_t* cbytes, cimg_alloc_fn alloc, void* user, void** chunk_ptr)
{
    LOCK_ENGINE(e);
    if (n <= 0) return 0;
    if (!alloc || !chunk_ptr) { e->err = "null argument"; return -12; }
    const int rc = cimg_compress_batch_host_begin(e, p, n, h_raw, raw_off, nbytes, destsize, cbytes);
    if (rc) return rc;
    const int cut = n > 1 ? n / 2 : n;
    for (int g = 0; g < 2; g++) {
        const int a = g ? cut : 0, b = g ? n : cut;
        size_t total = 0;
        for (int i = a; i < b; i++) { if (cbytes[i] < 0) { e->err = "chunk failed to compress"; e->off.clear(); return cbytes[i]; } total += ((size_t)cbytes[i] + 63) & ~(size_t)63; }
        if (!total) { for (int i = a; i < b; i++) chunk_ptr[i] = nullptr; continue; }
        uint8_t* block = (uint8_t*)alloc(user, total);
        if (!block) { e->err = "the caller's allocator returned no memory"; e->off.clear(); return -4; }
        size_t at = 0;
        for (int i = a; i < b; i++) {
            chunk_ptr[i] = cbytes[i] > 0 ? block + at : nullptr;
            if (cbytes[i] > 0) memcpy(block + at, e->stage.data() + e->off[(size_t)i], (size_t)cbytes[i]);
            at += ((size_t)cbytes[i] + 63) & ~(size_t)63;
        }
    }
    e->off.clear();
    return 0;
}

int cimg_compress_batch_host(cimg_engine* e, const cimg_cparams* p, int32_t n, const void* h_raw, const int64_t* raw_off,
                             const int32_t* nbytes, void* h_comp, const int64_t* comp_off, const int32_t* destsize, int32_t* cbytes)
{
    LOCK_ENGINE(e);
    if (n <= 0) return 0;
    const int rc = cimg_compress_batch_host_begin(e, p, n, h_raw, raw_off, nbytes, destsize, cbytes);
    if (rc) return rc;
    return cimg_compress_batch_host_fetch(e, n, h_comp, comp_off);
}

int cimg_decompress_batch_host_sized(cimg_engine* e, int32_t n, const void* h_comp, const int64_t* comp_off, const int32_t* comp_size,
                                     void* h_raw, const int64_t* raw_off, const int32_t* cap, int32_t* status);
int cimg_decompress_batch_host(cimg_engine* e, int32_t n, const void* h_comp, const int64_t* comp_off, void* h_raw,
                               const int64_t* raw_off, const int32_t* cap, int32_t* status)
{
    return cimg_decompress_batch_host_sized(e, n, h_comp, comp_off, nullptr, h_raw, raw_off, cap, status);
}
int cimg_decompress_batch_host_sized(cimg_engine* e, int32_t n, const void* h_comp, const int64_t* comp_off, const int32_t* comp_size,
                                     void* h_raw, const int64_t* raw_off, const int32_t* cap, int32_t* status)
{
    LOCK_ENGINE(e);
    if (n <= 0) return 0;
    if (comp_size)
        for (int i = 0; i < n; i++) {
            int32_t cb = 0;
            if (comp_size[i] >= 16) memcpy(&cb, (const uint8_t*)h_comp + comp_off[i] + 12, 4);
            if (comp_size[i] < 32 || cb > comp_size[i]) { e->err = "chunk buffer shorter than its header says"; return -5; }
        }
    e->off.clear();                                          // as the engine: the staging area is reused, a pending _fetch is void
    std::vector<int32_t> nb((size_t)n), bs((size_t)n), st((size_t)n, 0);
    for (int i = 0; i < n; i++) {
        const uint8_t* c = (const uint8_t*)h_comp + comp_off[i];
        memcpy(&nb[(size_t)i], c + 4, 4); memcpy(&bs[(size_t)i], c + 8, 4);
        if (c[0] > 5) { e->err = "format version"; return -10; }
        if (bs[(size_t)i] <= 0 || (nb[(size_t)i] > 0 && bs[(size_t)i] > nb[(size_t)i]) || c[3] == 0) { e->err = "invalid header"; return -11; }
        if (nb[(size_t)i] > cap[i]) { e->err = "buffer too small"; return -6; }
    }
    const int rc = emu_decompress_batch(n, (const uint8_t*)h_comp, comp_off, nb.data(), bs.data(), (uint8_t*)h_raw, raw_off, st.data());
    if (status) memcpy(status, st.data(), sizeof(int32_t) * (size_t)n);
    if (rc < 0) { e->err = "emulated decompress batch rejected"; return rc; }
    for (int i = 0; i < n; i++) if (st[(size_t)i] < 0) { e->err = "chunk decode failed"; return st[(size_t)i]; }
    return 0;
}

}  // extern "C"
