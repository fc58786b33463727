// launch_h3.hip — conv_igemm_h3 (f16x3) instantiations of the seven PoseNN layers, each under its
// own kernel name (LAYER tag) so per-kernel profiles separate layers that share a tile shape.
#include "launch_h3_impl.h"

namespace davo {

hipError_t launch_layer_h3(int layer, int tile, const ConvParamsH& p, dim3 grid, hipStream_t s) {
    using namespace h3impl;
    if (is_208(tile)) return launch_layer_h3s(layer, p, grid, s);
    switch (layer) {
        case 0: return launch_tile<7, 2, 1, true, 32>(tile, p, grid, s);
        case 1: return launch_tile<5, 2, 2, true, 32>(tile, p, grid, s);
        case 2: return launch_tile<3, 1, 3, false, 64>(tile, p, grid, s);
        case 3: return launch_tile<3, 1, 4, false, 128>(tile, p, grid, s);
        case 4: return launch_tile<3, 1, 5, false, 256>(tile, p, grid, s);
        case 5: return launch_tile<3, 1, 6, false, 256>(tile, p, grid, s);
        case 6: return launch_tile<3, 2, 7, false, 256>(tile, p, grid, s);
    }
    return hipErrorInvalidValue;
}

namespace {

// the shared-tap instantiation's own conditions (launch_h3_impl.h, launch_m) and the grid shape the id order assumes
template <int LAYER>
bool mainrem_ok(const ConvParamsH& pm, int n_main, int n_rem) {
    using namespace h3impl;
    constexpr int RATE = layer_rate(LAYER);
    if (!(pm.xs && pm.rate == RATE && pm.pad_l == RATE && pm.pad_t == RATE && pm.Hin == pm.Hout && pm.Win == pm.Wout &&
          pm.nchunks % 3 == 0 && pm.Wout > 2 * RATE && pm.x_pix_log2 >= 7 && use_m16()))
        return false;
    return !(n_rem < 64 || n_rem % 64 || n_rem > 256 || n_main < n_rem / 2 || n_main % 8);
}

template <int LAYER, int TNM = 4>
hipError_t launch_mainrem(const ConvParamsH& pm, int n_main, const ConvParamsH& pr, int n_rem, int order, hipStream_t s) {
    using namespace h3impl;
    constexpr int RATE = layer_rate(LAYER);
    using TM_ = TileH<4, 2, 2, TNM, 2>;
    using TX = TileX<4, 2, 2, TNM, 2, RATE>;
    using TR = TileH<4, 2, 1, 2, DAVO_REM_STAGES>;
    static_assert(DAVO_REM_STAGES == 3, "conv_igemm_h3_mainrem instantiates the three-slot remainder tile");
    if (!mainrem_ok<LAYER>(pm, n_main, n_rem)) return hipErrorNotSupported;
    constexpr int lds = TX::LDS_BYTES > TR::LDS_BYTES_DMA ? TX::LDS_BYTES : TR::LDS_BYTES_DMA;
    auto kern = conv_igemm_h3_mainrem<LAYER, RATE, TNM>;
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(n_main + n_rem), dim3(TM_::THREADS), lds, s, pm, pr, n_main, n_rem, order);
    return hipGetLastError();
}

}  // namespace

bool layer_h3_mainrem_supported(int layer, const ConvParamsH& pm, int n_main, int n_rem) {
    if (layer == 3) return mainrem_ok<4>(pm, n_main, n_rem);
    if (layer == 4) return mainrem_ok<5>(pm, n_main, n_rem);
    if (layer == 5) return mainrem_ok<6>(pm, n_main, n_rem);
    return false;
}

hipError_t launch_layer_h3_mainrem(int layer, const ConvParamsH& pm, int n_main, const ConvParamsH& pr, int n_rem, int order, hipStream_t s) {
    if (order == 1 && (n_rem % 8 || n_main % 8)) order = 0;
    if (layer == 3) return launch_mainrem<4, 2>(pm, n_main, pr, n_rem, order, s);     // cnv4: 256x128 main tiles (N = 128)
    if (layer == 4) return launch_mainrem<5>(pm, n_main, pr, n_rem, order, s);
    if (layer == 5) return launch_mainrem<6>(pm, n_main, pr, n_rem, order, s);
    return hipErrorNotSupported;
}

}  // namespace davo
