// deinterleave_kernel.h -- interleaved pixels (R G B A R G B A ...) -> one plane per channel, on the device.
//
// The reference does this on the host between reading scanlines and compressing them (image_algo::deinterleave,
// compressed/image_algo.h:84-111, called by the read path image.h:1880).  Here it is the step in front of the codec
// kernels: the interleaved scanlines cross PCIe once, the planes they are split into never leave the device, and the
// encode launch reads them from there (engine.hip: cimg_compress_batch_host_interleaved_begin).
//
// One wave per tile.  A tile is up to 16 KiB of interleaved bytes (a multiple of 16 pixels): it is staged into LDS
// with 16-byte coalesced loads, then for every channel each lane gathers the 16 / typesize elements of 16 consecutive
// bytes of that channel's plane from LDS (stride nch * typesize) and stores them with one 16-byte coalesced store.
// HBM traffic is the algorithmic minimum: every byte read once, written once.
#pragma once
#include "decode_kernel.h"   // wave copies, u128 helpers

namespace cimg {

struct DeinterleaveArgs {
    const uint8_t* src;       // npixels * nch elements of ts bytes, interleaved
    uint8_t* dst;             // plane c starts at dst + c * plane_stride
    int64_t plane_stride;     // bytes, a multiple of 16
    int64_t npixels;
    int32_t nch, ts;          // channels, bytes per element (1, 2, 4 or 8)
    int32_t tile_pixels;      // a multiple of 16; tile_pixels * nch * ts <= lds_bytes
    int32_t lds_bytes;
};

// pixels per tile for a channel count and element size: as many as fit 16 KiB, a multiple of 16, at least 16
CIMG_HD int deinterleave_tile_pixels(int nch, int ts)
{
    const int fit = 16384 / (nch * ts);
    return fit >= 16 ? (fit & ~15) : 16;
}
CIMG_HD int deinterleave_lds_bytes(int nch, int ts) { return ((deinterleave_tile_pixels(nch, ts) * nch * ts + 15) & ~15) + 16; }

template <int TS> CIMG_DEV uint64_t lds_element(const uint8_t* lds, int off)
{
    if constexpr (TS == 1) return lds[off];
    else if constexpr (TS == 2) return *reinterpret_cast<const uint16_t*>(lds + off);
    else if constexpr (TS == 4) return *reinterpret_cast<const uint32_t*>(lds + off);
    else return (uint64_t)*reinterpret_cast<const uint32_t*>(lds + off) | ((uint64_t)*reinterpret_cast<const uint32_t*>(lds + off + 4) << 32);
}

template <int TS> CIMG_DEV void deinterleave_tile(const DeinterleaveArgs& a, uint8_t* lds, int64_t tile)
{
    constexpr int G = 16 / TS;                                  // elements of one channel in 16 output bytes
    const int nch = a.nch;
    const int64_t p0 = tile * a.tile_pixels;
    const int np = (int)(a.npixels - p0 < a.tile_pixels ? a.npixels - p0 : a.tile_pixels);
    const int bytes = np * nch * TS;
    const uint8_t* src = a.src + p0 * nch * TS;
    // ---- stage the tile ---------------------------------------------------------------------------------------
    wave_copy_g2l(src, lds, 0, bytes & ~15);
    for (int t = bytes & ~15; t < bytes; t += 64) {
        FOR_LANES_W(l) { if (t + l < bytes) lds[t + l] = src[t + l]; }
    }
    // ---- one channel after the other: 16 bytes of its plane per lane --------------------------------------------
    const int units = np / G;
    const int stride = nch * TS;
    for (int c = 0; c < nch; c++) {
        uint8_t* plane = a.dst + (int64_t)c * a.plane_stride + p0 * TS;
        for (int u0 = 0; u0 < units; u0 += 64) {
            LV<u128> o;
            FOR_LANES(l) {
                const int u = u0 + l < units ? u0 + l : 0;
                const int base = (u * G) * stride + c * TS;
                uint32_t w[4] = {0, 0, 0, 0};
                CIMG_UNROLL
                for (int k = 0; k < G; k++) {
                    const uint64_t e = lds_element<TS>(lds, base + k * stride);
                    if constexpr (TS == 8) { w[2 * k] = (uint32_t)e; w[2 * k + 1] = (uint32_t)(e >> 32); }
                    else w[(k * TS) >> 2] |= (uint32_t)e << (8 * ((k * TS) & 3));
                }
                o[l].x = w[0]; o[l].y = w[1]; o[l].z = w[2]; o[l].w = w[3];
            }
            FOR_LANES_W(l) { if (u0 + l < units) st128u(plane + 16 * (size_t)(u0 + l), o[l]); }
        }
        // the pixels of a last, short tile that do not fill 16 bytes
        const int done = units * G;
        const int tail = (np - done) * TS;                      // < 16
        FOR_LANES_W(l) {
            if (l < tail) plane[(size_t)done * TS + l] = lds[(done + l / TS) * stride + c * TS + l % TS];
        }
    }
}

CIMG_DEV void deinterleave_wave(const DeinterleaveArgs& a, uint8_t* lds, int64_t tile)
{
    switch (a.ts) {
    case 1: deinterleave_tile<1>(a, lds, tile); break;
    case 2: deinterleave_tile<2>(a, lds, tile); break;
    case 4: deinterleave_tile<4>(a, lds, tile); break;
    default: deinterleave_tile<8>(a, lds, tile); break;
    }
}

}  // namespace cimg
