// Stylised path (reference models.py:120-180, rendering.py:118-178).  Placeholder entry points:
// they fail loudly until the fused kernels of this file land (no CPU fallback, no silent success).
#include "common.h"
using namespace tgtc;

#define TGTC_NOT_YET(name) return fail(TGTC_ERR_UNSUPPORTED, name ": stylised kernels are not built into this library yet")

extern "C" int tgtc_style_create(const tgtc_linear*, int, const tgtc_linear*, int, int, tgtc_net**) { TGTC_NOT_YET("style_create"); }
extern "C" int tgtc_concat_mlp_forward(const tgtc_net*, const float*, const float*, int64_t, float*, void*) { TGTC_NOT_YET("concat_mlp_forward"); }
extern "C" int tgtc_style_mlp_forward(const tgtc_net*, const float*, const float*, const float*, int64_t, float*, void*) { TGTC_NOT_YET("style_mlp_forward"); }
extern "C" int tgtc_styled_forward_rays(const tgtc_net*, const tgtc_net*, const double*, const double*, const float*, const float*, int64_t, int, float*, float*, void*) { TGTC_NOT_YET("styled_forward_rays"); }
extern "C" int tgtc_render_rays_styled(const tgtc_net*, const tgtc_net*, const tgtc_net*, const double*, const double*, const float*, int64_t, int, int, float, float, const float*, void*, size_t, float*, float*, float*, float*, void*) { TGTC_NOT_YET("render_rays_styled"); }
