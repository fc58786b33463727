// pk_fma_probe.hip — does the packed-float32 sequence hipcc formed in the fused pose-head epilogue ("A flaky sum")
// give launch-to-launch different results on its own?  Every lane runs the exact instruction sequence (fixed registers, inline asm)
// on its own data, many times, next to the same arithmetic in scalar instructions, and counts disagreements per output.
//   hipcc --offload-arch=gfx950 -O2 -o pk_fma_probe tools/exp/pk_fma_probe.hip && ./pk_fma_probe [waves per SIMD 1..8] [nops between]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// mode 1: every other workgroup keeps the matrix pipe of its SIMDs busy (as the neighbouring workgroups' K loops do in the real launch)
__global__ __launch_bounds__(256) void probe(const float* __restrict__ in, unsigned* __restrict__ bad, int iters, float scale, int mode, float* sink) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (mode == 1 && (blockIdx.x & 1)) {
        half8 a, b;
        for (int k = 0; k < 8; ++k) { a[k] = (_Float16)in[(t * 8 + k) & 1023]; b[k] = (_Float16)in[(t * 8 + k + 5) & 1023]; }
        f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        for (int it = 0; it < iters * 2; ++it) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
        }
        sink[t] = c0[0] + c1[1] + c2[2] + c3[3];
        return;
    }
    float s0 = in[t * 8 + 0], s1 = in[t * 8 + 1], w0 = in[t * 8 + 2], w1 = in[t * 8 + 3], w2 = in[t * 8 + 4];
    unsigned long long sc = __float_as_uint(scale);
    unsigned nbad[6] = {0, 0, 0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        float q0, q1, q2, q3, q4, q5;
        asm volatile(
            "v_mov_b32 v14, %[s0]\n\tv_mov_b32 v15, %[s1]\n\tv_mov_b32 v18, %[w0]\n\tv_mov_b32 v19, %[w1]\n\tv_mov_b32 v20, %[w2]\n\t"
            "s_nop 7\n\t"
            "v_pk_mul_f32 v[4:5], %[sc], v[14:15] op_sel_hi:[0,1]\n\t"
            "v_mov_b32 v21, v18\n\t"
            "v_mov_b32 v14, v19\n\t"
            "v_mov_b32 v15, v20\n\t"
            "v_mbcnt_hi_u32_b32 v13, -1, v13\n\t"
            "v_pk_fma_f32 v[6:7], v[18:19], v[4:5], 0 op_sel_hi:[1,0,0]\n\t"
            "v_pk_fma_f32 v[8:9], v[20:21], v[4:5], 0 op_sel_hi:[1,1,0]\n\t"
            "v_pk_fma_f32 v[4:5], v[14:15], v[4:5], 0 op_sel:[0,1,0] op_sel_hi:[1,1,0]\n\t"
            "v_bitop3_b32 v14, v13, 32, 63 bitop3:8\n\t"      /* as in the kernel: the next VALU instruction overwrites src0's low register */
            "s_nop 7\n\t"
            "v_mov_b32 %[q0], v6\n\tv_mov_b32 %[q1], v7\n\tv_mov_b32 %[q2], v8\n\tv_mov_b32 %[q3], v9\n\tv_mov_b32 %[q4], v4\n\tv_mov_b32 %[q5], v5\n\t"
            : [q0] "=v"(q0), [q1] "=v"(q1), [q2] "=v"(q2), [q3] "=v"(q3), [q4] "=v"(q4), [q5] "=v"(q5)
            : [s0] "v"(s0), [s1] "v"(s1), [w0] "v"(w0), [w1] "v"(w1), [w2] "v"(w2), [sc] "s"(sc)
            : "v4", "v5", "v6", "v7", "v8", "v9", "v13", "v14", "v15", "v18", "v19", "v20", "v21");
        const float a = __fmul_rn(s0, scale), b = __fmul_rn(s1, scale);
        const float r[6] = {__fmul_rn(a, w0), __fmul_rn(a, w1), __fmul_rn(a, w2), __fmul_rn(b, w0), __fmul_rn(b, w1), __fmul_rn(b, w2)};
        const float q[6] = {q0, q1, q2, q3, q4, q5};
#pragma unroll
        for (int k = 0; k < 6; ++k) nbad[k] += __float_as_uint(q[k]) != __float_as_uint(r[k]);
        s0 = s0 * 1.0001f + 0.001f; s1 = s1 * 0.9999f - 0.002f;            // new data every round
        asm volatile("" : "+v"(s0), "+v"(s1));
    }
    for (int k = 0; k < 6; ++k)
        if (nbad[k]) atomicAdd(bad + k, nbad[k]);
}

int main(int argc, char** argv) {
    const int wps = argc > 1 ? atoi(argv[1]) : 4;                          // waves per SIMD: blocks of 4 waves, wps blocks per CU
    const int mode = argc > 2 ? atoi(argv[2]) : 0;
    const int nblk = 256 * wps, n = nblk * 256, iters = 20000;
    float* h = (float*)malloc((size_t)n * 8 * sizeof(float));
    srand(7);
    for (int i = 0; i < n * 8; ++i) h[i] = (float)rand() / RAND_MAX * 4.f - 2.f;
    float* d; unsigned* bad; float* sink;
    hipMalloc(&d, (size_t)n * 8 * sizeof(float)); hipMalloc(&bad, 6 * sizeof(unsigned)); hipMalloc(&sink, (size_t)n * sizeof(float));
    hipMemcpy(d, h, (size_t)n * 8 * sizeof(float), hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(bad, 0, 6 * sizeof(unsigned));
        hipLaunchKernelGGL(probe, dim3(nblk), dim3(256), 0, 0, d, bad, iters, 0.0078125f, mode, sink);
        unsigned out[6];
        hipMemcpy(out, bad, sizeof out, hipMemcpyDeviceToHost);
        printf("mode %d, waves/SIMD %d, %d lanes x %d rounds: disagreements per output q0..q5 = %u %u %u %u %u %u\n", mode, wps, n, iters, out[0], out[1], out[2], out[3], out[4], out[5]);
    }
    return 0;
}
