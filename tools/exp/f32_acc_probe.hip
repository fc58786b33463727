// f32_acc_probe.hip — does the FP32 matrix pipe's sustained rate depend on how many accumulators a wave cycles through?
// (measurement only; DESIGN.md section 9: the float32 convolutions hold 0.88-0.90 busy where mfma_peak_probe's loop - four accumulators -
// sustains 0.98.)  Loops of nothing but v_mfma_f32_32x32x2_f32, accumulators in place, NACC independent chains per wave:
//   hipcc --offload-arch=gfx950 -O3 -o f32_acc_probe f32_acc_probe.hip && ./f32_acc_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__device__ inline float rnd(unsigned s) { s = s * 747796405u + 2891336453u; s = ((s >> ((s >> 28) + 4)) ^ s) * 277803737u; return ((s >> 9) & 0xffff) * (1.0f / 32768.0f) - 1.0f; }

template <int NACC, bool AGPR, int THREADS, int MINW>
__global__ __launch_bounds__(THREADS, MINW) void k(float* out, int iters) {
    const unsigned t = blockIdx.x * THREADS + threadIdx.x;
    float a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = rnd(t * 16 + i); b[i] = rnd(t * 16 + 8 + i) * 0.02f; }
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 64 / NACC; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if (AGPR) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a[u & 3]), "v"(b[i & 3]));
                else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[u & 3]), "v"(b[i & 3]));
            }
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) out[t] = s;
}

template <typename F> static void run(const char* name, F launch, int waves_per_cu) {
    float* out;
    CHECK(hipMalloc(&out, 1 << 24));
    const int iters = 20000;
    launch(out, 200);
    CHECK(hipDeviceSynchronize());
    for (int rep = 0; rep < 3; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        launch(out, iters);
        CHECK(hipDeviceSynchronize());
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const double flops = 256.0 * waves_per_cu * iters * 64.0 * (32.0 * 32 * 2 * 2);
        printf("%-58s %7.1f TFLOP/s = %.3f of 157.3\n", name, flops / s / 1e12, flops / s / 157.3e12);
    }
    CHECK(hipFree(out));
}

int main() {
    run("4 accumulators in VGPRs, 4 waves per CU (1 per SIMD)", [](float* o, int it) { hipLaunchKernelGGL((k<4, false, 256, 1>), dim3(256), dim3(256), 0, 0, o, it); }, 4);
    run("4 accumulators in VGPRs, 8 waves per CU (2 per SIMD)", [](float* o, int it) { hipLaunchKernelGGL((k<4, false, 512, 2>), dim3(256), dim3(512), 0, 0, o, it); }, 8);
    run("16 accumulators in AGPRs, 4 waves per CU (1 per SIMD)", [](float* o, int it) { hipLaunchKernelGGL((k<16, true, 256, 1>), dim3(256), dim3(256), 0, 0, o, it); }, 4);
    run("8 accumulators in AGPRs, 8 waves per CU (2 per SIMD)", [](float* o, int it) { hipLaunchKernelGGL((k<8, true, 512, 2>), dim3(256), dim3(512), 0, 0, o, it); }, 8);
    run("4 accumulators in AGPRs, 8 waves per CU (2 per SIMD)", [](float* o, int it) { hipLaunchKernelGGL((k<4, true, 512, 2>), dim3(256), dim3(512), 0, 0, o, it); }, 8);
    run("16 accumulators in VGPRs, 4 waves per CU (1 per SIMD)", [](float* o, int it) { hipLaunchKernelGGL((k<16, false, 256, 1>), dim3(256), dim3(256), 0, 0, o, it); }, 4);
    return 0;
}
