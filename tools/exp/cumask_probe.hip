// Probe: does hipExtStreamCreateWithCUMask work here, and how do mask bits map to XCDs?
// hipcc --offload-arch=gfx950 -O2 -o cumask_probe tools/exp/cumask_probe.hip && ./cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstring>
#include <chrono>

__global__ void where(int* xcc, int* cu, long spin) {
    if (threadIdx.x == 0) {
        unsigned x = 0, hw = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        xcc[blockIdx.x] = x & 0xf;
        cu[blockIdx.x] = hw;
    }
    long t0 = clock64();
    while (clock64() - t0 < spin) {}
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
    int n = 2048;
    int *dx, *dc;
    CK(hipMalloc(&dx, n * 4)); CK(hipMalloc(&dc, n * 4));
    std::vector<int> hx(n), hc(n);
    for (int variant = 0; variant < 8; ++variant) {
        uint32_t mask[8];
        memset(mask, 0, sizeof mask);
        const char* name = "";
        for (int i = 0; i < 256; ++i) {
            bool on = false;
            if (variant == 0) { on = true; name = "all"; }
            if (variant == 1) { on = (i % 8) < 4; name = "bits i%8<4"; }
            if (variant == 2) { on = i < 128; name = "bits i<128"; }
            if (variant == 3) { on = (i % 8) == 5; name = "bits i%8==5"; }
            if (variant == 4) { on = i >= 128; name = "bits i>=128"; }
            if (variant == 5) { on = (i / 8) % 2 == 0; name = "bits (i/8)%2==0"; }
            if (variant == 6) { on = (i / 8) % 2 == 1; name = "bits (i/8)%2==1"; }
            if (variant == 7) { on = i < 64; name = "bits i<64"; }
            if (on) mask[i / 32] |= 1u << (i % 32);
        }
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask);
        if (e != hipSuccess) { printf("variant %s: hipExtStreamCreateWithCUMask -> %s\n", name, hipGetErrorString(e)); continue; }
        CK(hipMemsetAsync(dx, 0xff, n * 4, s));
        hipLaunchKernelGGL(where, dim3(n), dim3(64), 0, s, dx, dc, 20000L);
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(hx.data(), dx, n * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hc.data(), dc, n * 4, hipMemcpyDeviceToHost));
        int cnt[16] = {0};
        for (int i = 0; i < n; ++i) cnt[hx[i] & 15]++;
        printf("variant %-14s: WGs per XCC:", name);
        for (int k = 0; k < 8; ++k) printf(" %d", cnt[k]);
        // distinct (xcc, hw_id cu/se bits) count
        std::vector<unsigned> seen;
        for (int i = 0; i < n; ++i) {
            unsigned key = ((unsigned)hx[i] << 20) | ((unsigned)hc[i] & 0xFF00);   // SE/SA/CU id bits vary by arch; coarse
            bool f = false; for (auto k : seen) if (k == key) { f = true; break; }
            if (!f) seen.push_back(key);
        }
        printf("  distinct (xcc,hwid&0xFF00) = %zu\n", seen.size());
        CK(hipStreamDestroy(s));
    }
    // two disjoint halves: do they map to disjoint CUs, and do kernels on them run side by side?
    uint32_t mlo[8], mhi[8];
    for (int w = 0; w < 8; ++w) { mlo[w] = w < 4 ? 0xffffffffu : 0u; mhi[w] = w < 4 ? 0u : 0xffffffffu; }
    hipStream_t sa, sb;
    CK(hipExtStreamCreateWithCUMask(&sa, 8, mlo));
    CK(hipExtStreamCreateWithCUMask(&sb, 8, mhi));
    int *dx2, *dc2;
    CK(hipMalloc(&dx2, n * 4)); CK(hipMalloc(&dc2, n * 4));
    std::vector<int> hx2(n), hc2(n);
    hipEvent_t e0, e1, e2;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    const long spin = 2000000L;     // ~1 ms per workgroup
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, sa));
        hipLaunchKernelGGL(where, dim3(128), dim3(64), 0, sa, dx, dc, spin);
        CK(hipEventRecord(e1, sa));
        CK(hipStreamSynchronize(sa));
        float one = 0; CK(hipEventElapsedTime(&one, e0, e1));
        // both at once: 128 WGs each, one per CU of its half
        auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(where, dim3(128), dim3(64), 0, sa, dx, dc, spin);
        hipLaunchKernelGGL(where, dim3(128), dim3(64), 0, sb, dx2, dc2, spin);
        CK(hipStreamSynchronize(sa)); CK(hipStreamSynchronize(sb));
        double both = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        // same stream twice for comparison
        t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(where, dim3(128), dim3(64), 0, sa, dx, dc, spin);
        hipLaunchKernelGGL(where, dim3(128), dim3(64), 0, sa, dx2, dc2, spin);
        CK(hipStreamSynchronize(sa));
        double serial = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        printf("rep %d: one kernel %.3f ms; two kernels on the two half-masks %.3f ms (wall); two on one stream %.3f ms\n", rep, one, both, serial);
    }
    CK(hipMemcpy(hx.data(), dx, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hc.data(), dc, n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hx2.data(), dx2, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hc2.data(), dc2, n * 4, hipMemcpyDeviceToHost));
    int overlap = 0, da = 0, db = 0;
    std::vector<unsigned> A, B;
    for (int i = 0; i < 128; ++i) {
        unsigned ka = ((unsigned)hx[i] << 20) | ((unsigned)hc[i] & 0xFF00), kb = ((unsigned)hx2[i] << 20) | ((unsigned)hc2[i] & 0xFF00);
        bool f = false; for (auto k : A) if (k == ka) f = true; if (!f) A.push_back(ka);
        f = false; for (auto k : B) if (k == kb) f = true; if (!f) B.push_back(kb);
    }
    for (auto a : A) for (auto b : B) if (a == b) ++overlap;
    da = (int)A.size(); db = (int)B.size();
    printf("half masks: %d and %d distinct CUs, %d in common\n", da, db, overlap);
    return 0;
}
