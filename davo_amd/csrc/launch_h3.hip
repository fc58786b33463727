// launch_h3.hip — conv_igemm_h3 (f16x3) instantiations of the seven PoseNN layers, each under its
// own kernel name (LAYER tag) so per-kernel profiles separate layers that share a tile shape.
#include "launch_h3_impl.h"

namespace davo {

hipError_t launch_layer_h3(int layer, int tile, const ConvParamsH& p, dim3 grid, hipStream_t s) {
    using namespace h3impl;
    if (tile == TILE_208x256) return launch_layer_h3s(layer, p, grid, s);
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

}  // namespace davo
