// launch_h3w.hip — conv_igemm_h3w (f16x3, 256x256 tiles on four waves of 128x128: cnv5 and cnv6; option "wave128").
#include "conv_igemm_h3w.h"
#include "launch.h"

namespace davo {
namespace {

template <int LAYER, int RATE>
hipError_t launch_w(const ConvParamsH& p, dim3 grid, hipStream_t s) {
    auto kern = conv_igemm_h3w<LAYER, RATE>;
    constexpr int lds = TileW<RATE>::LDS_BYTES;
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, grid, dim3(TileW<RATE>::THREADS), lds, s, p);
    return hipGetLastError();
}

}  // namespace

// what the kernel assumes of a launch: whole 256-row tiles of a 3x3 stride-1 layer with 256 output channels in blocks of 32, dilation
// = padding = the instantiation's rate, input in whole 32-channel blocks at a power-of-two pixel pitch, split-fp16 output
bool layer_h3w_supported(int layer, const ConvParamsH& p) {
    const int rate = layer == 4 ? 8 : (layer == 5 ? 2 : 0);
    return rate > 0 && p.rate == rate && p.pad_l == rate && p.pad_t == rate && p.Hin == p.Hout && p.Win == p.Wout && p.Wout > 2 * rate &&
           p.cb_log2 == 5 && p.cpb == 9 && p.nchunks % 9 == 0 && p.x_pix_log2 >= 7 && p.y_mode == 1 && p.y_ld >= 32 && p.y_ld % 32 == 0 &&
           p.y_coff % 32 == 0 && p.Cout == 256 && p.ntiles_n == 1 && p.M % 256 == 0 && p.M <= p.Mtot &&
           p.Wout > 16 && 32 / p.Wout + 2 <= p.Hout &&                               // the patch slots' pixel walk: one wrap per step
           ((long)p.Mtot << p.x_pix_log2) < 0xFFFFFF00l && 256l * p.w_row_bytes < 0x7FFFFFFFl;     // 32-bit buffer offsets; the marker of an out-of-range row
}

// the layer's remainder rows on 256 x 64 tiles (conv_igemm_h3w64): grid = 256-row tiles x four N tiles
template <int LAYER, int RATE>
static hipError_t launch_w64(const ConvParamsH& p, dim3 grid, hipStream_t s) {
    auto kern = conv_igemm_h3w64<LAYER, RATE>;
    constexpr int lds = TileW64<RATE>::LDS_BYTES;
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, p);
    return hipGetLastError();
}

// p as for the 256x256 tile but with ntiles_n = 4
hipError_t launch_layer_h3w64(int layer, const ConvParamsH& p, hipStream_t s) {
    ConvParamsH q = p;
    q.ntiles_n = 1;
    if (p.ntiles_n != 4 || !layer_h3w_supported(layer, q)) return hipErrorNotSupported;
    const dim3 grid((unsigned)((p.M - p.mtile0 * 256) / 256 * 4));
    if (layer == 4) return launch_w64<5, 8>(p, grid, s);
    return launch_w64<6, 2>(p, grid, s);
}

// cnv4 on 256 x 128 tiles (conv_igemm_h3w128): every row of the launch, grid = 256-row tiles
bool layer_h3w128_supported(const ConvParamsH& p) {
    return p.rate == 4 && p.pad_l == 4 && p.pad_t == 4 && p.Hin == p.Hout && p.Win == p.Wout && p.Wout > 16 &&
           p.cb_log2 == 5 && p.cpb == 9 && p.nchunks % 9 == 0 && p.x_pix_log2 >= 7 && p.y_mode == 1 && p.y_ld >= 32 && p.y_ld % 32 == 0 &&
           p.y_coff % 32 == 0 && p.Cout == 128 && p.M % 256 == 0 && p.M <= p.Mtot && 32 / p.Wout + 2 <= p.Hout &&
           ((long)p.Mtot << p.x_pix_log2) < 0xFFFFFF00l && 128l * p.w_row_bytes < 0x7FFFFFFFl;
}

hipError_t launch_layer_h3w128(const ConvParamsH& p, hipStream_t s) {
    if (!layer_h3w128_supported(p)) return hipErrorNotSupported;
    auto kern = conv_igemm_h3w128<4, 4>;
    constexpr int lds = TileW128<4>::LDS_BYTES;
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned)((p.M - p.mtile0 * 256) / 256)), dim3(256), lds, s, p);
    return hipGetLastError();
}

hipError_t launch_layer_h3w(int layer, const ConvParamsH& p, dim3 grid, hipStream_t s) {
    if (!layer_h3w_supported(layer, p) || grid.y != 1) return hipErrorNotSupported;
    if (layer == 4) return launch_w<5, 8>(p, grid, s);
    return launch_w<6, 2>(p, grid, s);
}

}  // namespace davo
