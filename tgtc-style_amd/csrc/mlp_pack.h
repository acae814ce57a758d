// Host-side weight packer: turns nn.Linear weights ([out,in] row-major fp32) into the
// fragment-ordered fp16 (hi [+ lo]) stream the fused kernels consume through the LDS ring,
// plus a padded fp32 bias table.  The k-slot -> input-column maps live in mlp_core.h.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <vector>

#include "../../include/tgtc_hip.h"
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

// ------------------------------------------------------------------------------------------------
// Power-of-two cross-layer equalisation of ReLU layers before packing (every precision).
//
// fp16 operands have a 30-binade exponent range and the parity modes lean on its lower end: the lo halves of fp16x3 are
// 2^-11 of their values (an activation of 1e-3 has a lo of 5e-7, an fp16 subnormal with three significant bits; below
// 6e-8 it is gone), and the block-scaled corrections of fp16mx lose their benefit with the dynamic range inside a
// 32-value block (activations) or a weight row.  A network whose features live on very different scales -- what trained
// MLPs look like -- therefore loses accuracy in EVERY mode (tests/probes/emu_mx_e2e.py: log-normal per-feature scales
// put 6e-3 on a rendered ray in fp16x3, 4e-3 in fp16mx; equalised: 2e-6 / 2e-4).  A ReLU layer is positively
// homogeneous, so row i of layer l (and its bias) times 2^-k and column i of every consumer of that feature times 2^k is
// the SAME function -- exactly: powers of two commute with every rounding short of overflow -- and
// k_i = round(log2(rowmax_i / colmax_i) / 2), taken relative to the layer's median, balances the row's range against
// its consumers' column range feature by feature (cross-layer range equalisation, as used for integer quantisation).
// Layers whose outputs leave the operator (base_remap, concat_features, the heads) keep their rows.
// TGTC_NO_EQUALISE=1 packs the weights as given (development; tests/test_precision_robustness_gpu.py then fails).
struct EqualisedNet {
    std::vector<std::vector<float>> w, b;
    std::vector<tgtc_linear> lin;
    struct Cons {
        int layer, col0;
    };
    static bool enabled() {
        const char* e = std::getenv("TGTC_NO_EQUALISE");
        return !(e && e[0] == '1');
    }
    void copy(const tgtc_linear* layers, int n) {
        w.resize(n), b.resize(n), lin.resize(n);
        for (int l = 0; l < n; ++l) {
            w[l].assign(layers[l].weight, layers[l].weight + (size_t)layers[l].out_features * layers[l].in_features);
            b[l].assign(layers[l].bias, layers[l].bias + layers[l].out_features);
            lin[l] = tgtc_linear{w[l].data(), b[l].data(), layers[l].out_features, layers[l].in_features};
        }
    }
    // rows of ReLU layer l against columns [col0, col0 + rows) of its consumers
    void run(int l, std::initializer_list<Cons> cons) {
        if (!enabled()) return;
        const int n = lin[l].out_features, in = lin[l].in_features;
        std::vector<int> k(n, 0);
        for (int i = 0; i < n; ++i) {
            float r1 = 0.0f, r2 = 0.0f;
            for (int c = 0; c < in; ++c) r1 = std::fmax(r1, std::fabs(w[l][(size_t)i * in + c]));
            for (const Cons& q : cons)
                for (int r = 0; r < lin[q.layer].out_features; ++r)
                    r2 = std::fmax(r2, std::fabs(w[q.layer][(size_t)r * lin[q.layer].in_features + q.col0 + i]));
            if (r1 > 0.0f && r2 > 0.0f) k[i] = (int)std::nearbyint(0.5 * std::log2((double)r1 / (double)r2));
        }
        std::vector<int> sorted(k);
        std::nth_element(sorted.begin(), sorted.begin() + n / 2, sorted.end());
        const int med = sorted[n / 2];   // only the spread matters: the layer keeps its typical activation scale
        for (int i = 0; i < n; ++i) {
            const int ki = std::max(-8, std::min(8, k[i] - med));
            if (ki == 0) continue;
            const float down = std::ldexp(1.0f, -ki), up = std::ldexp(1.0f, ki);
            for (int c = 0; c < in; ++c) w[l][(size_t)i * in + c] *= down;
            b[l][i] *= down;
            for (const Cons& q : cons)
                for (int r = 0; r < lin[q.layer].out_features; ++r)
                    w[q.layer][(size_t)r * lin[q.layer].in_features + q.col0 + i] *= up;
        }
    }
};

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
};
