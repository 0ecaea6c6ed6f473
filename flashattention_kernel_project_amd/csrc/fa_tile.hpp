// fa_tile.hpp -- tile geometry and LDS images shared by the tiled attention-forward kernels.
#pragma once
#include "fa_common.hpp"

namespace fa {

constexpr int kWaves  = 8;
constexpr int kBlockM = 32 * kWaves;   // query rows per workgroup
constexpr int kBlockN = 64;            // keys per tile
constexpr float kThr  = 8.0f;          // lazy-rescale threshold, log2 domain (P <= 2^8 fits fp16)

template <int D> struct TileGeom {
    static constexpr int kRowBytes  = D * 2;
    static constexpr int kChunks    = D / 8;                 // 16-B chunks per row
    static constexpr int kTileBytes = kBlockN * kRowBytes;   // one K (or V) tile
    static constexpr int kBufBytes  = 2 * kTileBytes;        // K + V
    static constexpr int kLdsBytes  = 2 * kBufBytes;         // double buffered
    static constexpr int kLoads     = (kBlockN * kChunks) / (64 * kWaves);  // 16-B loads / thread / tile
    static constexpr int kKSteps    = D / 16;                // MFMA k-steps over d
    static constexpr int kDBlocks   = D / 32;                // 32-row blocks of O^T
    // K image: row-major rows of D*2 bytes, 16-B chunk index XORed with a row-derived value so
    // that the 16 lanes of a ds_read_b128 group (16 different rows, same chunk) hit 16 slots.
    static __device__ __forceinline__ unsigned k_swz(unsigned row) {
        return D == 64 ? ((row >> 1) & 7u) : (row & 15u);
    }
    static __device__ __forceinline__ unsigned k_off(unsigned row, unsigned chunk) {
        return row * kRowBytes + ((chunk ^ k_swz(row)) << 4);
    }
    // V image: [key/4][d/32] blocks of 256 B, inside a block [key%4][32 cols] (64-B rows).  The
    // 64-B row slot is rotated by the column block so a row's 16-B chunk writes spread over banks.
    static __device__ __forceinline__ unsigned v_off(unsigned key, unsigned chunk) {
        const unsigned dblk = chunk >> 2;
        return ((key >> 2) * kDBlocks + dblk) * 256u + (((key & 3u) ^ (dblk & 1u)) << 6) + ((chunk & 3u) << 4);
    }
};

}  // namespace fa
