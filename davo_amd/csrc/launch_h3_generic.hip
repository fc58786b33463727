// launch_h3_generic.hip — conv_igemm_h3 for davo_conv2d_same's generic shapes (test hook): LAYER tag 0,
// SMALLC follows the channel-block size.
#include "launch_h3_impl.h"

namespace davo {
namespace {

template <int KS, int STRIDE>
hipError_t generic(int tile, const ConvParamsH& p, dim3 grid, hipStream_t s) {
    using namespace h3impl;
    return p.cb_log2 < 5 ? launch_tile<KS, STRIDE, 0, true, 256>(tile, p, grid, s)
                         : launch_tile<KS, STRIDE, 0, false, 256>(tile, p, grid, s);
}

}  // namespace

hipError_t launch_h3_generic(int KS, int stride, int tile, const ConvParamsH& p, dim3 grid, hipStream_t s) {
    if (stride == 1) {
        if (KS == 1) return generic<1, 1>(tile, p, grid, s);
        if (KS == 3) return generic<3, 1>(tile, p, grid, s);
        if (KS == 5) return generic<5, 1>(tile, p, grid, s);
        if (KS == 7) return generic<7, 1>(tile, p, grid, s);
    } else if (stride == 2) {
        if (KS == 1) return generic<1, 2>(tile, p, grid, s);
        if (KS == 3) return generic<3, 2>(tile, p, grid, s);
        if (KS == 5) return generic<5, 2>(tile, p, grid, s);
        if (KS == 7) return generic<7, 2>(tile, p, grid, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace davo
