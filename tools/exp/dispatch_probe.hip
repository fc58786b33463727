// dispatch_probe.hip — in which order does the hardware hand the workgroups of one launch to the compute units?
// (experiment behind MEASURED_AND_REJECTED.md "merged main + remainder grid").  One workgroup per CU at a time (140 KB of LDS), a grid of
// n_long "long" and n_short "short" workgroups in the id order of that experiment; every workgroup records where and when it
// ran.  Build + run:  hipcc --offload-arch=gfx950 -O2 -o /tmp/dispatch_probe tools/exp/dispatch_probe.hip && /tmp/dispatch_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

struct Rec { unsigned b, hwid, xcc, cls; unsigned long long t0, t1; };

__global__ __launch_bounds__(512) void probe(Rec* out, int n_long, int n_short, int us_long, int us_short, int order) {
    extern __shared__ char lds[];
    const int b = blockIdx.x, h = n_short >> 1;
    bool shrt;
    if (order == 0) shrt = b >= n_long;                                   // all long, then all short (two launches' order)
    else if (b < 2 * h) shrt = order == 1 ? ((b >> 3) & 1) == 0 : ((b >> 5) & 1) == 0;   // merged order: half of the short ones first,
    else shrt = b >= h + n_long;                                          // alternating per workgroup of an XCD (1) or per 4 of them (2)
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();       // 100 MHz
    const unsigned long long want = (unsigned long long)(shrt ? us_short : us_long) * 100ull;
    while (__builtin_amdgcn_s_memrealtime() - t0 < want) __builtin_amdgcn_s_sleep(32);
    if (threadIdx.x == 0) {
        lds[0] = 1;
        Rec r;
        r.b = b; r.cls = shrt;
        r.hwid = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);    // HW_REG_HW_ID
        r.xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);     // HW_REG_XCC_ID
        r.t0 = t0; r.t1 = __builtin_amdgcn_s_memrealtime();
        out[b] = r;
    }
}

int main() {
    const int n_long = 768, n_short = 256, n = n_long + n_short;
    Rec* d;
    hipMalloc(&d, n * sizeof(Rec));
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    for (int order = 0; order < 3; ++order) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(probe, dim3(n), dim3(512), 140 * 1024, 0, d, n_long, n_short, 150, 55, order);
            hipDeviceSynchronize();
        }
        std::vector<Rec> r(n);
        hipMemcpy(r.data(), d, n * sizeof(Rec), hipMemcpyDeviceToHost);
        unsigned long long tmin = ~0ull, tmax = 0;
        for (auto& x : r) { tmin = std::min(tmin, x.t0); tmax = std::max(tmax, x.t1); }
        std::map<unsigned, std::vector<Rec>> cu;
        for (auto& x : r) cu[(x.xcc << 16) | (x.hwid & 0xff00)].push_back(x);       // xcc, se_id / sh_id / cu_id bits
        std::map<int, int> hist;                                                    // long workgroups per CU -> CUs
        for (auto& kv : cu) {
            int nl = 0;
            for (auto& x : kv.second) nl += !x.cls;
            hist[nl * 10 + (int)kv.second.size() - nl]++;
        }
        printf("order %d: %zu distinct (xcc, se, cu); makespan %.1f us; CUs by (long, short) workgroups:", order, cu.size(), (tmax - tmin) / 100.0);
        for (auto& kv : hist) printf("  (%d,%d) x%d", kv.first / 10, kv.first % 10, kv.second);
        printf("\n");
        if (getenv("PROBE_DUMP")) {
            for (auto& x : r) if (x.xcc == 0) printf("D %d %u %u %u %u %.1f %.1f\n", order, x.b, (x.hwid >> 13) & 7, (x.hwid >> 8) & 15, x.cls, (x.t0 - tmin) / 100.0, (x.t1 - tmin) / 100.0);
        }
        // the first CUs' sequences
        int shown = 0;
        for (auto& kv : cu) {
            if (shown++ >= 6) break;
            auto v = kv.second;
            std::sort(v.begin(), v.end(), [](const Rec& a, const Rec& b) { return a.t0 < b.t0; });
            printf("  xcc %u hw %04x:", kv.first >> 16, kv.first & 0xffff);
            for (auto& x : v) printf(" [%u %c @%.0f]", x.b, x.cls ? 's' : 'L', (x.t0 - tmin) / 100.0);
            printf("\n");
        }
    }
    hipFree(d);
    return 0;
}
