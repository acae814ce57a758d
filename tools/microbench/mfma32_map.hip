// Lane maps of v_mfma_f32_32x32x16_f16 on gfx950, found by probing: A[i][k] = 1 at one (lane, element), B = all ones -> which
// D entries light up gives the row of that A slot; likewise for B; k pairing by A one-hot x B one-hot.
//   hipcc --offload-arch=gfx950 -O2 -o tools/microbench/mfma32_map tools/microbench/mfma32_map.hip && tools/microbench/mfma32_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

__global__ void probe(int a_lane, int a_el, int b_lane, int b_el, float* out) {
    const int l = threadIdx.x;
    half8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (_Float16)((a_lane < 0) ? 1.0f : (l == a_lane && j == a_el ? 1.0f : 0.0f));
        b[j] = (_Float16)((b_lane < 0) ? 1.0f : (l == b_lane && j == b_el ? 1.0f : 0.0f));
    }
    float16v c = {};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int v = 0; v < 16; ++v) out[l * 16 + v] = c[v];
}

int main() {
    float* d;
    hipMalloc(&d, 64 * 16 * 4);
    std::vector<float> h(64 * 16);
    auto run = [&](int al, int ae, int bl, int be) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, al, ae, bl, be, d);
        hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    };
    // 1. D layout: A one-hot row slot x B ones: lit entries share the A slot's row; B one-hot x A ones: the column
    printf("A slot (lane, el) -> lit D entries (lane range, vgpr set)\n");
    for (int al : {0, 1, 31, 32, 33, 63})
        for (int ae : {0, 7}) {
            run(al, ae, -1, 0);
            int lmin = 64, lmax = -1;
            unsigned vset = 0;
            for (int l = 0; l < 64; ++l)
                for (int v = 0; v < 16; ++v)
                    if (h[l * 16 + v] != 0.f) lmin = l < lmin ? l : lmin, lmax = l > lmax ? l : lmax, vset |= 1u << v;
            printf("  A(%2d,%d): lanes %d..%d vgprs 0x%04x\n", al, ae, lmin, lmax, vset);
        }
    printf("B slot (lane, el) -> lit D entries\n");
    for (int bl : {0, 1, 31, 32, 33, 63})
        for (int be : {0, 7}) {
            run(-1, 0, bl, be);
            int lmin = 64, lmax = -1;
            unsigned vset = 0;
            for (int l = 0; l < 64; ++l)
                for (int v = 0; v < 16; ++v)
                    if (h[l * 16 + v] != 0.f) lmin = l < lmin ? l : lmin, lmax = l > lmax ? l : lmax, vset |= 1u << v;
            printf("  B(%2d,%d): lanes %d..%d vgprs 0x%04x\n", bl, be, lmin, lmax, vset);
        }
    // 2. rows of D: for A row slot lane r (el 0), B ones: which (lane, vgpr) are lit -> row r's location
    printf("row r (A lane r, el 0) -> D (lane half, vgpr) holding it\n");
    for (int r = 0; r < 32; ++r) {
        run(r, 0, -1, 0);
        for (int l : {0, 32})
            for (int v = 0; v < 16; ++v)
                if (h[l * 16 + v] != 0.f) printf("  row %2d: lane half %d vgpr %d\n", r, l / 32, v);
    }
    // 3. k pairing: A(lane 0, el e) x B(lane bl, el be) lit?
    printf("k pairing: A(lane al, el ae) meets B(lane bl, el be)\n");
    for (int al : {0, 32})
        for (int ae : {0, 3, 7})
            for (int bl : {0, 32})
                for (int be = 0; be < 8; ++be) {
                    run(al, ae, bl, be);
                    float s = 0;
                    for (auto x : h) s += x;
                    if (s != 0.f) printf("  A(%d,%d) <-> B(%d,%d)\n", al, ae, bl, be);
                }
    return 0;
}
