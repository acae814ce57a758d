// Host-side weight packer: turns nn.Linear weights ([out,in] row-major fp32) into the
// fragment-ordered fp16 (hi [+ lo]) stream the fused kernels consume through the LDS ring,
// plus a padded fp32 bias table.  The k-slot -> input-column maps live in mlp_core.h.
#pragma once
#include <cmath>
#include <cstring>
#include <vector>

#include "common.h"
#include "mlp_core.h"

namespace tgtc {

enum SegKind { SEG_ACT = 0, SEG_PE63 = 1, SEG_PE27 = 2, SEG_VEC32 = 3 };

struct Seg {
    SegKind kind;
    int col0;    // first column of this source inside the reference's concatenated input
    int ksteps;  // 32-deep k-steps this source occupies
};

struct LayerSpec {
    const float* W;
    const float* b;
    int out, in;
    std::vector<Seg> segs;
    int ksteps() const {
        int k = 0;
        for (auto& s : segs) k += s.ksteps;
        return k;
    }
    int row_tiles() const { return (out + 15) / 16; }
};

inline int seg_col(const Seg& s, int ks_in_seg, int g, int j) {
    int c = -1;
    switch (s.kind) {
        case SEG_ACT: c = act_col(ks_in_seg, g, j); break;
        case SEG_PE63: c = pe63_col(ks_in_seg, g, j); break;
        case SEG_PE27: c = pe27_col(g, j); break;
        case SEG_VEC32: c = vec32_col(g, j); break;
    }
    return c < 0 ? -1 : s.col0 + c;
}

struct PackedNet {
    std::vector<half_t> stream;  // fragments, chunk-padded
    std::vector<float> bias;     // 16 floats per row tile, layer after layer
    std::vector<int> frag0;      // first fragment of each layer
    std::vector<int> bias0;      // first bias float of each layer
    int n_frags = 0;
};

inline PackedNet pack_layers(const std::vector<LayerSpec>& layers, bool split) {
    PackedNet p;
    const int frag_halves = kFragHalves * (split ? 2 : 1);
    for (const LayerSpec& L : layers) {
        p.frag0.push_back(p.n_frags);
        p.bias0.push_back((int)p.bias.size());
        const int RT = L.row_tiles();
        for (int rt = 0; rt < RT; ++rt) {
            for (int r = 0; r < 16; ++r) {
                const int row = 16 * rt + r;
                p.bias.push_back(row < L.out ? L.b[row] : 0.0f);
            }
            int ks = 0;
            for (const Seg& s : L.segs) {
                for (int k = 0; k < s.ksteps; ++k, ++ks) {
                    const size_t base = p.stream.size();
                    p.stream.resize(base + frag_halves, (half_t)0.0f);
                    for (int lane = 0; lane < 64; ++lane) {
                        const int m = lane & 15, g = lane >> 4, row = 16 * rt + m;
                        for (int j = 0; j < 8; ++j) {
                            const int col = seg_col(s, k, g, j);
                            float w = 0.0f;
                            if (row < L.out && col >= 0 && col < L.in) w = L.W[(size_t)row * L.in + col];
                            const half_t hi = (half_t)w;
                            p.stream[base + lane * 8 + j] = hi;
                            if (split) p.stream[base + kFragHalves + lane * 8 + j] = (half_t)(w - (float)hi);
                        }
                    }
                    ++p.n_frags;
                }
            }
        }
    }
    const size_t chunk_halves = kChunkBytes / sizeof(half_t);
    p.stream.resize((p.stream.size() + chunk_halves - 1) / chunk_halves * chunk_halves, (half_t)0.0f);
    return p;
}

// The wide (32x32x16, mlp_wide.h) order of the same data: 32-row tiles, 16-deep k-steps (Seg::ksteps counts those),
// lane (r = lane&31, h = lane>>5); `col(seg, k, h, j)` maps a k-slot to the logical input column or -1.
// The bias table holds 32 floats per row tile in the accumulator's register order [h][q]: row 8*(q>>2) + 4h + (q&3).
template <class ColFn>
inline PackedNet pack_layers_w(const std::vector<LayerSpec>& layers, bool split, ColFn&& col) {
    PackedNet p;
    const int frag_halves = kFragHalves * (split ? 2 : 1);
    for (const LayerSpec& L : layers) {
        p.frag0.push_back(p.n_frags);
        p.bias0.push_back((int)p.bias.size());
        const int RT = (L.out + 31) / 32;
        for (int rt = 0; rt < RT; ++rt) {
            for (int h = 0; h < 2; ++h)
                for (int q = 0; q < 16; ++q) {
                    const int row = 32 * rt + 8 * (q >> 2) + 4 * h + (q & 3);
                    p.bias.push_back(row < L.out ? L.b[row] : 0.0f);
                }
            for (const Seg& s : L.segs) {
                for (int k = 0; k < s.ksteps; ++k) {
                    const size_t base = p.stream.size();
                    p.stream.resize(base + frag_halves, (half_t)0.0f);
                    for (int lane = 0; lane < 64; ++lane) {
                        const int r = lane & 31, h = lane >> 5, row = 32 * rt + r;
                        for (int j = 0; j < 8; ++j) {
                            const int c0 = col(s, k, h, j);
                            const int c = c0 < 0 ? -1 : s.col0 + c0;
                            float w = 0.0f;
                            if (row < L.out && c >= 0 && c < L.in) w = L.W[(size_t)row * L.in + c];
                            const half_t hi = (half_t)w;
                            p.stream[base + lane * 8 + j] = hi;
                            if (split) p.stream[base + kFragHalves + lane * 8 + j] = (half_t)(w - (float)hi);
                        }
                    }
                    ++p.n_frags;
                }
            }
        }
    }
    const size_t chunk_halves = kChunkBytes / sizeof(half_t);
    p.stream.resize((p.stream.size() + chunk_halves - 1) / chunk_halves * chunk_halves, (half_t)0.0f);
    return p;
}

}  // namespace tgtc

// The opaque handle of the C ABI.
struct tgtc_net {
    int kind;        // 0 = NeRF (StyleNerf), 1 = style pair (concat MLP + style MLP)
    int precision;   // TGTC_PREC_*
    char* dev;       // one device allocation: [bias table, padded][weight stream]
    size_t bias_bytes;
    size_t stream_bytes;
    int n_frags;
    // style pair only: second stream (the style MLP) inside the same allocation
    size_t bias2_off, bias2_bytes, stream2_off, stream2_bytes;
    int n_frags2;
    size_t stash_off;  // per-workgroup scratch slabs of the fused stylised kernel
    int n_wg;          // persistent grid size (= CUs)
    // NeRF handles of the parity precisions: the same network in the wide (32x32x16) order, [bias 16 KiB][stream][slack]
    size_t wide_off = 0, wide_stream_bytes = 0;
};
