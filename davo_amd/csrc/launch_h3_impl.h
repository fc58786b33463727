// launch_h3_impl.h — templates shared by the two translation units that instantiate conv_igemm_h3
// (launch_h3.hip: the seven PoseNN layers; launch_h3_generic.hip: davo_conv2d_same's generic shapes).
#pragma once
#include <cstdio>
#include <cstdlib>

#include "conv_igemm_h3.h"
#include "launch.h"

namespace davo {
namespace h3impl {

#ifndef DAVO_REM_STAGES
#define DAVO_REM_STAGES 3
#endif

// v_mfma 16x16x32 instead of 32x32x16 for the large tiles (higher held clock under matrix-dense load);
// tuning build only: DAVO_H3_M16=0 selects the 32x32x16 form (A/B measurements)
inline bool use_m16() {
    static int v = -1;
    if (v < 0) { const char* e = tuning_env("DAVO_H3_M16"); v = e ? atoi(e) : 1; }
    return v != 0;
}

// dilation rate of the PoseNN layer behind a LAYER tag (nets/posenn.py:213-215,238): cnv3 2, cnv4 4, cnv5 8, cnv6 2;
// 0 = the layer has no shared-tap instantiation (stride 2, 5x5 / 7x7, generic shapes)
constexpr int layer_rate(int layer) { return layer == 3 ? 2 : layer == 4 ? 4 : layer == 5 ? 8 : layer == 6 ? 2 : 0; }

// All f16x3 launches are LDS-DMA staged.  SMALLC (Cin < 32) is a property of the layer.
template <int KS, int STRIDE, int WM, int WN, int TM, int TN, int LAYER, bool SMALLC, bool M16, int NSTG = 2>
hipError_t launch_m(const ConvParamsH& p, dim3 grid, hipStream_t s) {
    using T = TileH<WM, WN, TM, TN, NSTG>;
    // measured per tile shape (gpurun_out/ab_r02v.log, B=32 and B=128): -3 % on the 256x256 tile (cnv5, cnv6 main launches),
    // -6 % on 256x128 (ab_r02w.log), +15 % on 256x64 (cnv3), +18 % on the 3-slot 128x128 remainder tile, level on 128x128 (cnv4)
    // ... and the deep-ring 128x128 tile (TM == 1, launches of at most one workgroup per CU): there a chunk is bound by what a CU
    // can take in from L2 (32 KB per chunk at ~45 GB/s per CU: 0.7 us against 0.3 us of matrix work), and the shared patch
    // cuts the pixel operand's 16 KB to 5.7 KB
    if constexpr (M16 && !SMALLC && KS == 3 && STRIDE == 1 && layer_rate(LAYER) > 0 && WM == 4 && WN == 2 &&
                  ((TM == 2 && (TN == 4 || TN == 2)) || (TM == 1 && TN == 2 && NSTG >= 4))) {
        // shared-tap staging (conv_igemm_h3.h, RATE > 0): one pixel patch per filter row serves its three taps
        constexpr int RATE = layer_rate(LAYER);
        if (p.xs && p.rate == RATE && p.pad_l == RATE && p.pad_t == RATE && p.Hin == p.Hout && p.Win == p.Wout &&
            p.nchunks % 3 == 0 && p.Wout > 2 * RATE && p.x_pix_log2 >= 7) {
            using TX = TileX<WM, WN, TM, TN, NSTG, RATE>;
            auto kx = conv_igemm_h3<KS, STRIDE, WM, WN, TM, TN, LAYER, true, SMALLC, M16, NSTG, RATE>;
            hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kx), TX::LDS_BYTES);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(kx, grid, dim3(T::THREADS), TX::LDS_BYTES, s, p);
            return hipGetLastError();
        }
    }
    auto kern = conv_igemm_h3<KS, STRIDE, WM, WN, TM, TN, LAYER, true, SMALLC, M16, NSTG>;
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), T::LDS_BYTES_DMA);
    if (e != hipSuccess) return e;
#ifdef DAVO_TUNING
    if (tuning_env("DAVO_PRINT_OCC")) {          // tuning build only: resident workgroups per CU the runtime computes
        static bool once = false;
        if (!once) {
            once = true;
            int nb = -1;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, T::THREADS, T::LDS_BYTES_DMA);
            fprintf(stderr, "occupancy layer %d tile %dx%d stages %d: %d workgroups/CU (LDS %d B, grid %u)\n", LAYER, T::BMH, T::BNH, NSTG, nb,
                    T::LDS_BYTES_DMA, grid.x * grid.y);
        }
    }
#endif
    hipLaunchKernelGGL(kern, grid, dim3(T::THREADS), T::LDS_BYTES_DMA, s, p);
    return hipGetLastError();
}

// The 16x16x32 form is used for EVERY tile shape of cnv3..cnv7, so that an output element is summed in the
// same order whatever tile the launch plan gives it (batch-size invariance to the bit).
template <int KS, int STRIDE, int WM, int WN, int TM, int TN, int LAYER, bool SMALLC>
hipError_t launch_c(const ConvParamsH& p, dim3 grid, hipStream_t s) {
    if constexpr (LAYER >= 3 && !SMALLC) {
        if (use_m16()) return launch_m<KS, STRIDE, WM, WN, TM, TN, LAYER, SMALLC, true>(p, grid, s);
    }
    return launch_m<KS, STRIDE, WM, WN, TM, TN, LAYER, SMALLC, false>(p, grid, s);
}

// Deep rings.  A launch of at most one workgroup per CU (batch 1..4: cnv6 at B = 1 is 104 tiles of 128x128 on 256 CUs) has
// no second workgroup to hide the LDS-DMA latency behind, and with two ring slots only one chunk is in flight: a chunk then
// costs a memory round trip (0.7 us on cnv6, 1.3 us on cnv7's stride-2 gather, against 0.1-0.3 us of matrix work).  Such
// launches take as many ring slots as the LDS holds: the wait is counted (vmcnt), so NSTG - 1 chunks stay in flight.
// Same tiles, same products in the same order: bit-identical to the two-slot kernels.
template <int KS, int STRIDE, int LAYER, int MAXBN>
hipError_t launch_tile_deep(int tile, const ConvParamsH& p, dim3 grid, hipStream_t s, bool* handled) {
    *handled = true;
    if (tile == TILE_128x32) return launch_m<KS, STRIDE, 4, 1, 1, 1, LAYER, false, true, 6>(p, grid, s);           // 6 x 20 KB
    if constexpr (MAXBN >= 64)
        if (tile == TILE_256x64) return launch_m<KS, STRIDE, 4, 2, 2, 1, LAYER, false, true, 3>(p, grid, s);       // 3 x 40 KB
    if constexpr (MAXBN >= 128) {
        if (tile == TILE_128x128) {
#ifndef DAVO_DEEP128_WAVES4
#define DAVO_DEEP128_WAVES4 0   /* measured: eight waves (with the shared patch) 52.4 / 30.6 us for cnv6 / cnv5 at B = 1, four waves 53.7 / 31.2 */
#endif
            if constexpr (LAYER == 7 || DAVO_DEEP128_WAVES4) return launch_m<KS, STRIDE, 2, 2, 2, 2, LAYER, false, true, 4>(p, grid, s);  // 4 x 32 KB
            else return launch_m<KS, STRIDE, 4, 2, 1, 2, LAYER, false, true, 4>(p, grid, s);
        }
        if (tile == TILE_256x128) return launch_m<KS, STRIDE, 4, 2, 2, 2, LAYER, false, true, 3>(p, grid, s);      // 3 x 48 KB
    }
    if constexpr (MAXBN >= 256)
        if (tile == TILE_128x256) return launch_m<KS, STRIDE, 2, 4, 2, 2, LAYER, false, true, 3>(p, grid, s);
    *handled = false;
    return hipSuccess;
}

// MAXBN bounds the instantiations to the N tiles a layer can use (its padded Cout)
template <int KS, int STRIDE, int LAYER, bool SMALLC, int MAXBN>
hipError_t launch_tile(int tile, const ConvParamsH& p, dim3 grid, hipStream_t s) {
    if constexpr (LAYER >= 4 && !SMALLC) {
        if (p.deep && use_m16()) {
            bool handled = false;
            const hipError_t e = launch_tile_deep<KS, STRIDE, LAYER, MAXBN>(tile, p, grid, s, &handled);
            if (handled) return e;
        }
    }
    if (tile == TILE_128x32) return launch_c<KS, STRIDE, 4, 1, 1, 1, LAYER, SMALLC>(p, grid, s);
    if constexpr (MAXBN >= 64)
        if (tile == TILE_256x64) return launch_c<KS, STRIDE, 4, 2, 2, 1, LAYER, SMALLC>(p, grid, s);
    if constexpr (MAXBN >= 128) {
        if (tile == TILE_256x128) return launch_c<KS, STRIDE, 4, 2, 2, 2, LAYER, SMALLC>(p, grid, s);
        if (tile == TILE_128x128) {
            // cnv7 (stride 2, pose head in the epilogue): four waves of 64x64 measured 7 % faster than eight of
            // 32x64 (fewer LDS fragment reads per MFMA); the stride-1 layers measured the other way round
            if constexpr (LAYER == 7) return launch_c<KS, STRIDE, 2, 2, 2, 2, LAYER, SMALLC>(p, grid, s);
            else {
                // a launch of at most one workgroup per CU (remainder rows) has no second workgroup to hide the
                // DMA latency behind: three ring slots instead of two (cnv6.rem 0.073 -> 0.061 ms, cnv5.rem 0.041 -> 0.036)
                if constexpr (LAYER >= 3 && !SMALLC)
                    if ((long)grid.x * grid.y <= 256 && use_m16())
                        return launch_m<KS, STRIDE, 4, 2, 1, 2, LAYER, SMALLC, true, DAVO_REM_STAGES>(p, grid, s);
                return launch_c<KS, STRIDE, 4, 2, 1, 2, LAYER, SMALLC>(p, grid, s);
            }
        }
    }
    if constexpr (MAXBN >= 256) {
        if (tile == TILE_128x256) return launch_c<KS, STRIDE, 2, 4, 2, 2, LAYER, SMALLC>(p, grid, s);
        if constexpr (LAYER != 0)
            if (tile == TILE_256x256) return launch_c<KS, STRIDE, 4, 2, 2, 4, LAYER, SMALLC>(p, grid, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace h3impl
}  // namespace davo
