// mfma_peak_probe.hip — what a loop of nothing but matrix instructions sustains on this chip (no LDS, no memory traffic),
// with board power and shader clock sampled while it runs: the practical ceiling that the conv kernels' roofline
// fractions (quoted against the nominal 2.5 PFLOP/s fp16 / 157.3 TFLOP/s fp32 peaks) should be read against.
// Operands: "random" (uniform in [-1, 1): what a matrix pipe toggles on real activations) or "zero" (least switching).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_peak_probe mfma_peak_probe.hip && ./mfma_peak_probe [seconds per case]
#include <hip/hip_runtime.h>
#include <dirent.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ inline float rnd(unsigned s) { s = s * 747796405u + 2891336453u; s = ((s >> ((s >> 28) + 4)) ^ s) * 277803737u; return ((s >> 9) & 0xffff) * (1.0f / 32768.0f) - 1.0f; }

// every kernel: 16 matrix instructions per iteration, accumulators updated in place (dst = src C), independent chains
// mode: 0 random, 1 zeros, 2 / 3 / 4: random with the low 4 / 7 / 10 mantissa bits of BOTH operands cleared (how the power of a
// product depends on the populated bits of its operands)
__device__ inline _Float16 mask_low(_Float16 x, int mode) {
    if (mode < 2) return x;
    unsigned short u = __builtin_bit_cast(unsigned short, x);
    u &= (unsigned short)(0xffffu << (mode == 2 ? 4 : mode == 3 ? 7 : 10));
    return __builtin_bit_cast(_Float16, u);
}
__global__ __launch_bounds__(256) void k_f16_16(float* out, int iters, int zero) {
    const unsigned t = blockIdx.x * 256 + threadIdx.x;
    half8 a[2], b[2];
    for (int i = 0; i < 2; ++i) for (int k = 0; k < 8; ++k) { a[i][k] = zero == 1 ? (_Float16)0 : mask_low((_Float16)rnd(t * 64 + i * 8 + k), zero); b[i][k] = zero == 1 ? (_Float16)0 : mask_low((_Float16)(rnd(t * 64 + 32 + i * 8 + k) * 0.02f), zero); }
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[u]), "v"(b[i & 1]));
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[t] = s;
}
__global__ __launch_bounds__(256) void k_f16_32(float* out, int iters, int zero) {
    const unsigned t = blockIdx.x * 256 + threadIdx.x;
    half8 a[2], b[2];
    for (int i = 0; i < 2; ++i) for (int k = 0; k < 8; ++k) { a[i][k] = zero ? (_Float16)0 : (_Float16)rnd(t * 64 + i * 8 + k); b[i][k] = zero ? (_Float16)0 : (_Float16)(rnd(t * 64 + 32 + i * 8 + k) * 0.02f); }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[u & 1]), "v"(b[i & 1]));
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) out[t] = s;
}
__global__ __launch_bounds__(256) void k_f32_32(float* out, int iters, int zero) {
    const unsigned t = blockIdx.x * 256 + threadIdx.x;
    float a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = zero ? 0.f : rnd(t * 16 + i); b[i] = zero ? 0.f : rnd(t * 16 + 8 + i) * 0.02f; }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[u]), "v"(b[i]));
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) out[t] = s;
}

// v_mfma_f32_16x16x32_f16 fed from LDS at a chosen ratio: RD ds_read_b128 per 16 MFMAs (the 256x256 conv tile reads 4 per 16:
// 0.25 fragment reads per MFMA; a 128x128-per-wave tile would read 0.167).  The fragments read are the MFMA operands.
template <int RD>
__global__ __launch_bounds__(256) void k_f16_lds(float* out, int iters, int zero) {
    __shared__ __attribute__((aligned(16))) _Float16 lds[32768];           // 64 KB
    const unsigned t = blockIdx.x * 256 + threadIdx.x;
    for (int i = threadIdx.x; i < 32768; i += 256) lds[i] = zero ? (_Float16)0 : (_Float16)(rnd(t * 977 + i) * (i & 1 ? 0.02f : 1.f));
    __syncthreads();
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    half8 f[2][8];                               // two fragment sets: the MFMAs of a phase read one while the LDS reads fill the other
    unsigned off = (threadIdx.x * 16u) & 0xffffu;
    for (int i = 0; i < 8; ++i) f[0][i] = f[1][i] = *reinterpret_cast<const half8*>(reinterpret_cast<const char*>(lds) + ((off + i * 4096u) & 0xfff0u));
    constexpr int SUB = 8 / RD;                  // sub-iterations of 16 MFMAs per phase: every index is static
    for (int it = 0; it < iters / (2 * SUB); ++it) {
#pragma unroll
        for (int ph = 0; ph < 2; ++ph)
#pragma unroll
            for (int sub = 0; sub < SUB; ++sub) {
                off = (off + 4112u) & 0xfff0u;
#pragma unroll
                for (int r = 0; r < RD; ++r)
                    f[ph ^ 1][sub * RD + r] = *reinterpret_cast<const half8*>(reinterpret_cast<const char*>(lds) + ((off + r * 8208u) & 0xfff0u));
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[ph][(i + u) & 7], f[ph][(i + 3 * u + 1) & 7], acc[i], 0, 0, 0);
            }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[t] = s;
}

struct Hwmon { std::string power, freq; };
static std::vector<Hwmon> find_hwmon() {
    std::vector<Hwmon> v;
    for (int card = 0; card < 64; ++card) {
        const std::string base = "/sys/class/drm/card" + std::to_string(card) + "/device/hwmon";
        DIR* d = opendir(base.c_str());
        if (!d) continue;
        while (dirent* e = readdir(d)) {
            if (e->d_name[0] == '.') continue;
            const std::string h = base + "/" + e->d_name;
            for (const char* pf : {"/power1_average", "/power1_input"}) {
                FILE* f = fopen((h + pf).c_str(), "r");
                if (f) { fclose(f); v.push_back({h + pf, h + "/freq1_input"}); break; }
            }
        }
        closedir(d);
    }
    return v;
}
static double read_num(const std::string& p) { FILE* f = fopen(p.c_str(), "r"); if (!f) return -1; double x = -1; if (fscanf(f, "%lf", &x) != 1) x = -1; fclose(f); return x; }

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 1.5;
    int ncu = 0;
    CHECK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
    float* out;
    CHECK(hipMalloc(&out, 1 << 24));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const std::vector<Hwmon> hw = find_hwmon();
    printf("CUs %d, %zu hwmon cards (the loaded card = the one with the highest mean power)\n", ncu, hw.size());
    printf("%-26s %-7s %5s %10s %8s %8s\n", "instruction", "data", "w/SIMD", "TFLOP/s", "W", "MHz");
    const char* names[6] = {"v_mfma_f32_16x16x32_f16", "v_mfma_f32_32x32x16_f16", "v_mfma_f32_32x32x2_f32", "16x16x32 + 2 LDS rd/16", "16x16x32 + 4 LDS rd/16", "16x16x32 + 8 LDS rd/16"};
    const double flop_iter[6] = {16.0 * 2 * 16 * 16 * 32, 16.0 * 2 * 32 * 32 * 16, 16.0 * 2 * 32 * 32 * 2, 16.0 * 2 * 16 * 16 * 32, 16.0 * 2 * 16 * 16 * 32, 16.0 * 2 * 16 * 16 * 32};
    const int iters[6] = {20000, 10000, 5000, 20000, 20000, 20000};
    for (int kind = 0; kind < 6; ++kind)
        for (int zero = 0; zero < 5; ++zero)
            for (int wps : {2}) {
                if (kind >= 1 && zero >= 2) continue;
                if (kind >= 3 && zero) continue;
                const int blocks = ncu * wps;
                const double flops = flop_iter[kind] * iters[kind] * 4.0 * blocks;
                std::atomic<bool> stop{false};
                std::vector<double> psum(hw.size(), 0.0), fsum(hw.size(), 0.0);
                long nsamp = 0;
                std::thread sampler([&] {
                    while (!stop.load()) {
                        for (size_t i = 0; i < hw.size(); ++i) { psum[i] += read_num(hw[i].power) / 1e6; fsum[i] += read_num(hw[i].freq) / 1e6; }
                        ++nsamp;
                        std::this_thread::sleep_for(std::chrono::milliseconds(10));
                    }
                });
                auto launch = [&] {
                    if (kind == 0) hipLaunchKernelGGL(k_f16_16, dim3(blocks), dim3(256), 0, 0, out, iters[kind], zero);
                    else if (kind == 1) hipLaunchKernelGGL(k_f16_32, dim3(blocks), dim3(256), 0, 0, out, iters[kind], zero);
                    else if (kind == 2) hipLaunchKernelGGL(k_f32_32, dim3(blocks), dim3(256), 0, 0, out, iters[kind], zero);
                    else if (kind == 3) hipLaunchKernelGGL(k_f16_lds<2>, dim3(blocks), dim3(256), 0, 0, out, iters[kind], zero);
                    else if (kind == 4) hipLaunchKernelGGL(k_f16_lds<4>, dim3(blocks), dim3(256), 0, 0, out, iters[kind], zero);
                    else hipLaunchKernelGGL(k_f16_lds<8>, dim3(blocks), dim3(256), 0, 0, out, iters[kind], zero);
                };
                // run for `seconds`; the rate is that of the second half (the clock has settled under the cap by then)
                const auto t0 = std::chrono::steady_clock::now();
                double ms_late = 0; int n_late = 0;
                while (true) {
                    CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
                    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                    if (el > seconds / 2) { ms_late += ms; ++n_late; }
                    if (el > seconds) break;
                }
                stop.store(true); sampler.join();
                size_t best = 0;
                for (size_t i = 1; i < hw.size(); ++i) if (psum[i] > psum[best]) best = i;
                printf("%-26s %-7s %5d %10.0f %8.0f %8.0f\n", names[kind], zero == 0 ? "random" : zero == 1 ? "zero" : zero == 2 ? "rnd-4b" : zero == 3 ? "rnd-7b" : "rnd-10b", wps, flops * n_late / (ms_late * 1e-3) * 1e-12,
                       hw.empty() || !nsamp ? -1.0 : psum[best] / nsamp, hw.empty() || !nsamp ? -1.0 : fsum[best] / nsamp);
                fflush(stdout);
            }
    return 0;
}
