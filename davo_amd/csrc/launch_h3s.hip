// launch_h3s.hip — conv_igemm_h3s (f16x3, 208-pixel tiles: 256 channels on eight waves for cnv5, cnv6 and cnv7, 128 on four for cnv4).
#include "conv_igemm_h3s.h"
#include "launch.h"

#ifndef DAVO_H3S_NSA7
#define DAVO_H3S_NSA7 3
#endif
#ifndef DAVO_H3S_NSA56
#define DAVO_H3S_NSA56 3
#endif

namespace davo {
namespace {

template <int KS, int STRIDE, int LAYER, int NSA, int WAVES = 8>
hipError_t launch_s(const ConvParamsH& p, dim3 grid, hipStream_t s) {
    auto kern = conv_igemm_h3s<KS, STRIDE, LAYER, NSA, WAVES>;
    constexpr int lds = TileSW<WAVES>::lds_bytes(NSA);
    static_assert(lds <= 160 * 1024, "LDS per workgroup");
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, grid, dim3(TileSW<WAVES>::THREADS), lds, s, p);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_layer_h3s(int layer, const ConvParamsH& p, dim3 grid, hipStream_t s) {
    if (p.cb_log2 != 5) return hipErrorInvalidValue;          // whole 32-channel blocks per chunk (Cin >= 32)
    switch (layer) {
        case 3: return launch_s<3, 1, 4, 3, 4>(p, grid, s);               // cnv4: 208 x 128, four waves, three pixel ring slots
        case 4: return launch_s<3, 1, 5, DAVO_H3S_NSA56>(p, grid, s);
        case 5: return launch_s<3, 1, 6, DAVO_H3S_NSA56>(p, grid, s);
        case 6: return launch_s<3, 2, 7, DAVO_H3S_NSA7>(p, grid, s);     // pixel DMA two chunks ahead (stride-2 gather: first-touch L2 misses)
    }
    return hipErrorInvalidValue;
}

}  // namespace davo
