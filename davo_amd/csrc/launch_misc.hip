// launch_misc.hip — the small kernels of the path (SE squeeze/excite, mask + pack, pose head, the
// cnv1 LDS-patch kernel, the direct-convolution cross-check) and the per-device attribute cache.
#include <mutex>
#include <set>
#include <utility>

#include "conv_patch_h3.h"
#include "conv_patch_f32.h"
#include "launch.h"
#include "prologue.h"

namespace davo {

hipError_t ensure_dynamic_lds(const void* kernel, int bytes) {
    static std::mutex mu;
    static std::set<std::pair<int, const void*>> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    const auto key = std::make_pair(dev, kernel);
    if (done.count(key)) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done.insert(key);
    return e;
}

hipError_t launch_se_squeeze(const float* d_flow, int B, int HW, const Variant& v, float* d_partial, hipStream_t s) {
    hipLaunchKernelGGL(se_squeeze_partial, dim3(SQ_CHUNKS, 2, B), dim3(256), 0, s, d_flow, HW, v.norm_flow, v.abs_mode, d_partial);
    return hipGetLastError();
}

hipError_t launch_se_excite(const float* d_partial, int B, int HW, const Variant& v, const float* w1, const float* b1,
                            const float* w2, const float* b2, const float* wstatic, float* d_tab, unsigned* d_range_reset, hipStream_t s) {
    hipLaunchKernelGGL(se_excite, dim3(B, 3), dim3(64), 0, s, d_partial, HW, v, w1, b1, w2, b2, wstatic, d_tab, d_range_reset);
    return hipGetLastError();
}

hipError_t launch_se_squeeze_excite(const float* d_flow, int B, int HW, const Variant& v, float* d_partial, unsigned* d_counters,
                                    const float* w1, const float* b1, const float* w2, const float* b2, const float* wstatic,
                                    float* d_tab, unsigned* d_range_reset, hipStream_t s) {
    hipLaunchKernelGGL(se_squeeze_excite, dim3(SQ_CHUNKS, 2, B), dim3(256), 0, s, d_flow, HW, v, d_partial, d_counters, w1, b1, w2, b2,
                       wstatic, d_tab, d_range_reset);
    return hipGetLastError();
}

hipError_t launch_mask_pack(int ld, const uint8_t* d_img, const float* d_flow, const float* d_seg, const float* d_tab,
                            const Variant& v, int B, int H, int W, float* d_packed, hipStream_t s) {
    const long nthreads = (long)2 * B * H * (W / 4);
    const dim3 grid((unsigned)((nthreads + 255) / 256));
    if (ld == 16) hipLaunchKernelGGL(mask_pack<16>, grid, dim3(256), 0, s, d_img, d_flow, d_seg, d_tab, v, B, H, W, d_packed);
    else if (ld == 8) hipLaunchKernelGGL(mask_pack<8>, grid, dim3(256), 0, s, d_img, d_flow, d_seg, d_tab, v, B, H, W, d_packed);
    else if (ld == 10) hipLaunchKernelGGL(mask_pack<10>, grid, dim3(256), 0, s, d_img, d_flow, d_seg, d_tab, v, B, H, W, d_packed);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_cnv1_patch(bool fused, const ConvPatchParams& p, int nblk, hipStream_t s) {
    hipError_t e = fused ? ensure_dynamic_lds(reinterpret_cast<const void*>(conv_patch_cnv1_h3<true>), cp1::LDS_BYTES)
                         : ensure_dynamic_lds(reinterpret_cast<const void*>(conv_patch_cnv1_h3<false>), cp1::LDS_BYTES);
    if (e != hipSuccess) return e;
    if (fused) hipLaunchKernelGGL(conv_patch_cnv1_h3<true>, dim3(nblk), dim3(cp1::THREADS), cp1::LDS_BYTES, s, p);
    else hipLaunchKernelGGL(conv_patch_cnv1_h3<false>, dim3(nblk), dim3(cp1::THREADS), cp1::LDS_BYTES, s, p);
    return hipGetLastError();
}

hipError_t launch_cnv2_patch(const ConvPatchParams& p, int nblk, hipStream_t s) {
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(conv_patch_cnv2_h3), cp2::LDS_BYTES);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(conv_patch_cnv2_h3, dim3(nblk), dim3(cp2::THREADS), cp2::LDS_BYTES, s, p);
    return hipGetLastError();
}

hipError_t launch_cnv1_patch_f32(const ConvPatchParams& p, int nblk, hipStream_t s) {
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(conv_patch_cnv1_f32), cp1::LDS_BYTES);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(conv_patch_cnv1_f32, dim3(nblk), dim3(cp1::THREADS), cp1::LDS_BYTES, s, p);
    return hipGetLastError();
}

hipError_t launch_cnv2_patch_f32(const ConvPatchParams& p, int nblk, hipStream_t s) {
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(conv_patch_cnv2_f32), cp2::LDS_BYTES);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(conv_patch_cnv2_f32, dim3(nblk), dim3(cp2::THREADS), cp2::LDS_BYTES, s, p);
    return hipGetLastError();
}

hipError_t launch_cnv3_patch_f32(const ConvPatchParams& p, int nblk, hipStream_t s) {
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(conv_patch_cnv3_f32), cp3::LDS_BYTES);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(conv_patch_cnv3_f32, dim3(nblk), dim3(cp3::THREADS), cp3::LDS_BYTES, s, p);
    return hipGetLastError();
}

hipError_t launch_cnv3_patch(const ConvPatchParams& p, int nblk, hipStream_t s) {
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(conv_patch_cnv3_h3), cp3::LDS_BYTES);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(conv_patch_cnv3_h3, dim3(nblk), dim3(cp3::THREADS), cp3::LDS_BYTES, s, p);
    return hipGetLastError();
}

// every workgroup records the XCD it runs on (HW_REG_XCC_ID)
__global__ __launch_bounds__(64) void xcd_probe_kernel(unsigned* out) {
    if (threadIdx.x == 0) out[blockIdx.y * gridDim.x + blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15u;
}

hipError_t xcd_round_robin_probe(hipStream_t s, int* ok) {
    constexpr int NX = 64, NY = 4;
    unsigned* d = nullptr;
    unsigned h[NX * NY];
    *ok = 0;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&d), sizeof h);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(xcd_probe_kernel, dim3(NX, NY), dim3(64), 0, s, d);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d);
    if (e != hipSuccess) return e;
    bool same = true;
    for (int y = 1; y < NY; ++y)
        for (int x = 0; x < NX; ++x) same = same && h[y * NX + x] == h[x];
    *ok = same ? 1 : 0;
    return hipSuccess;
}

hipError_t launch_splitk_fixup(const float* d_part, long M, int N, int S, int relu, uint8_t* d_y, unsigned* d_range, hipStream_t s) {
    if (N < 32 || N % 32 || S < 2 || M < 1) return hipErrorInvalidValue;
    const long pairs = M * (N / 2);
    hipLaunchKernelGGL(splitk_fixup, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, s, d_part, pairs, N, S, relu, d_y, d_range);
    return hipGetLastError();
}

hipError_t launch_pose_from_tiles(const float* d_tiles, int NB, int P, int bm, int mtiles, int ntiles_n,
                                  const float* d_bpred, float* d_pose, const SnapArgs& snap, hipStream_t s) {
    hipLaunchKernelGGL(pose_from_tiles, dim3(NB * 6), dim3(64), 0, s, d_tiles, NB, P, bm, mtiles, ntiles_n, d_bpred, d_pose, snap);
    return hipGetLastError();
}

hipError_t launch_range_guard_snapshot(const SnapArgs& snap, hipStream_t s) {
    hipLaunchKernelGGL(range_guard_snapshot, dim3(256), dim3(256), 0, s, snap);
    return hipGetLastError();
}

hipError_t launch_pose_head(const float* d_c7, int NB, int P, const float* d_wpred, const float* d_bpred,
                            float* d_partial, float* d_pose, hipStream_t s) {
    hipLaunchKernelGGL(pose_head_partial, dim3(PH_SPLIT, NB, 2), dim3(256), 0, s, d_c7, P, d_wpred, d_partial);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(pose_finish, dim3((NB * 6 + 63) / 64), dim3(64), 0, s, d_partial, NB, P, d_bpred, d_pose);
    return hipGetLastError();
}

hipError_t launch_conv_direct(const float* x, int N, int Hin, int Win, int cin, int x_ld, int x_coff, const float* w, int KS,
                              int cout, const float* bias, int stride, int rate, int pt, int pl, int Ho, int Wo, int relu,
                              float* y, int y_ld, int y_coff, hipStream_t s) {
    const long total = (long)N * Ho * Wo * cout;
    hipLaunchKernelGGL(conv_direct, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, N, Hin, Win, cin, x_ld, x_coff,
                       w, KS, cout, bias, stride, rate, pt, pl, Ho, Wo, relu, y, y_ld, y_coff);
    return hipGetLastError();
}

}  // namespace davo
