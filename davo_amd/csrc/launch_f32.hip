// launch_f32.hip — instantiations and dispatch of conv_igemm_f32 (conv_igemm.h).
#include "conv_igemm.h"
#include "launch.h"

namespace davo {
namespace {

template <int KS, int STRIDE, int BN, int LAYER>
hipError_t launch_conv_t(const ConvParams& p, dim3 grid, hipStream_t s) {
    auto kern = conv_igemm_f32<KS, STRIDE, BN, LAYER>;
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), Tile<BN>::LDS_BYTES);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, grid, dim3(256), Tile<BN>::LDS_BYTES, s, p);
    return hipGetLastError();
}

template <int KS, int STRIDE, int LAYER>
hipError_t launch_bn(int BN, const ConvParams& p, dim3 grid, hipStream_t s) {
    switch (BN) {
        case 32: return launch_conv_t<KS, STRIDE, 32, LAYER>(p, grid, s);
        case 64: return launch_conv_t<KS, STRIDE, 64, LAYER>(p, grid, s);
        case 128: return launch_conv_t<KS, STRIDE, 128, LAYER>(p, grid, s);
    }
    return hipErrorInvalidValue;
}

template <int KS, int STRIDE, int RBN, int LAYER>
hipError_t launch_mainrem_t(const ConvParams& pm, const ConvParams& pr, int n_main, int n_rem, int groups, hipStream_t s) {
    auto kern = conv_igemm_f32_mainrem<KS, STRIDE, RBN, LAYER>;
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), Tile<128>::LDS_BYTES);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((n_main + n_rem) * groups), dim3(256), Tile<128>::LDS_BYTES, s, pm, pr, n_main, n_rem, groups);
    return hipGetLastError();
}

template <int LAYER>
hipError_t launch_n256_t(const ConvParams& p, dim3 grid, hipStream_t s) {
    auto kern = conv_igemm_f32_n256<3, 1, LAYER>;
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), Tile<256>::LDS_BYTES);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, grid, dim3(512), Tile<256>::LDS_BYTES, s, p);
    return hipGetLastError();
}

}  // namespace

// cnv5 / cnv6 on the 128 x 256 tile (eight waves): experiment, "f32_n256"
hipError_t launch_layer_n256(int layer, const ConvParams& p, dim3 grid, hipStream_t s) {
    if (layer == 4) return launch_n256_t<5>(p, grid, s);
    if (layer == 5) return launch_n256_t<6>(p, grid, s);
    return hipErrorInvalidValue;
}

// main (128-column tiles) + remainder (rbn-column tiles) of cnv4..cnv7 as one grid (conv_igemm.h: conv_igemm_f32_mainrem)
hipError_t launch_layer_mainrem(int layer, int rbn, const ConvParams& pm, const ConvParams& pr, int n_main, int n_rem, int groups, hipStream_t s) {
    if (n_main < 8 || n_main % 8 || n_rem < 1 || (groups > 1 && n_rem % 8)) return hipErrorInvalidValue;
    switch (layer * 1000 + rbn) {
        case 3032: return launch_mainrem_t<3, 1, 32, 4>(pm, pr, n_main, n_rem, groups, s);
        case 3064: return launch_mainrem_t<3, 1, 64, 4>(pm, pr, n_main, n_rem, groups, s);
        case 4032: return launch_mainrem_t<3, 1, 32, 5>(pm, pr, n_main, n_rem, groups, s);
        case 4064: return launch_mainrem_t<3, 1, 64, 5>(pm, pr, n_main, n_rem, groups, s);
        case 5032: return launch_mainrem_t<3, 1, 32, 6>(pm, pr, n_main, n_rem, groups, s);
        case 5064: return launch_mainrem_t<3, 1, 64, 6>(pm, pr, n_main, n_rem, groups, s);
        case 6032: return launch_mainrem_t<3, 2, 32, 7>(pm, pr, n_main, n_rem, groups, s);
        case 6064: return launch_mainrem_t<3, 2, 64, 7>(pm, pr, n_main, n_rem, groups, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_conv(int KS, int stride, int BN, const ConvParams& p, dim3 grid, hipStream_t s) {
    if (stride == 1) {
        switch (KS) {
            case 1: return launch_bn<1, 1, 0>(BN, p, grid, s);
            case 3: return launch_bn<3, 1, 0>(BN, p, grid, s);
            case 5: return launch_bn<5, 1, 0>(BN, p, grid, s);
            case 7: return launch_bn<7, 1, 0>(BN, p, grid, s);
        }
    } else if (stride == 2) {
        switch (KS) {
            case 1: return launch_bn<1, 2, 0>(BN, p, grid, s);
            case 3: return launch_bn<3, 2, 0>(BN, p, grid, s);
            case 5: return launch_bn<5, 2, 0>(BN, p, grid, s);
            case 7: return launch_bn<7, 2, 0>(BN, p, grid, s);
        }
    }
    return hipErrorInvalidValue;
}

hipError_t launch_layer(int layer, int BN, const ConvParams& p, dim3 grid, hipStream_t s) {
    if (layer == 0 && BN == 16) return launch_conv_t<7, 2, 16, 1>(p, grid, s);      // cnv1: 16 output channels, no padded columns
    switch (layer) {
        case 0: return launch_bn<7, 2, 1>(BN, p, grid, s);
        case 1: return launch_bn<5, 2, 2>(BN, p, grid, s);
        case 2: return launch_bn<3, 1, 3>(BN, p, grid, s);
        case 3: return launch_bn<3, 1, 4>(BN, p, grid, s);
        case 4: return launch_bn<3, 1, 5>(BN, p, grid, s);
        case 5: return launch_bn<3, 1, 6>(BN, p, grid, s);
        case 6: return launch_bn<3, 2, 7>(BN, p, grid, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace davo
