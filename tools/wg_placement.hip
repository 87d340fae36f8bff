// Where the dispatcher places the workgroups of a launch shaped like the matrix-core screening kernel (512
// threads, 45 KB of LDS, two workgroups per CU): for every workgroup the XCC / SE / CU it ran on and when it
// started, then how the first 512 pair up on CUs.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/wg_placement tools/wg_placement.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>

struct Rec {
    unsigned hwid, xcc;
    unsigned long long t0, t1;
};

__global__ __launch_bounds__(512, 4) void probe(Rec *out, int spinTicks)
{
    extern __shared__ float lds[];
    lds[threadIdx.x] = 0.0f;
    const unsigned long long t0 = wall_clock64();
    unsigned hwid = 0, xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    while (wall_clock64() - t0 < (unsigned long long)spinTicks) __builtin_amdgcn_s_sleep(16);
    if (threadIdx.x == 0) out[blockIdx.x] = Rec{hwid, xcc, t0, (unsigned long long)wall_clock64()};
}

int main()
{
    const int n = 1440;
    Rec *d;
    hipMalloc(&d, sizeof(Rec) * n);
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 46080);
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(probe, dim3(n), dim3(512), 46080, 0, d, 2000);
        hipDeviceSynchronize();
    }
    std::vector<Rec> h(n);
    hipMemcpy(h.data(), d, sizeof(Rec) * n, hipMemcpyDeviceToHost);
    unsigned long long tmin = ~0ull;
    for (auto &r : h) tmin = r.t0 < tmin ? r.t0 : tmin;
    auto cuOf = [](const Rec &r) {  // xcc, se (bits 15:13), sh (12), cu (11:8)
        return ((r.xcc & 0xf) << 12) | (((r.hwid >> 13) & 7) << 8) | (((r.hwid >> 12) & 1) << 4) | ((r.hwid >> 8) & 0xf);
    };
    std::map<unsigned, std::vector<int>> byCu;
    for (int i = 0; i < n; i++) byCu[cuOf(h[i])].push_back(i);
    printf("%zu distinct CUs\n", byCu.size());
    int shown = 0;
    for (auto &kv : byCu) {
        if (shown++ >= 6) break;
        printf("cu %05x:", kv.first);
        for (int i : kv.second) printf(" %d(@%.1fus)", i, (h[i].t0 - tmin) / 100.0);
        printf("\n");
    }
    // among the workgroups that start within the first 5 us: which block indices share a CU
    int pairsFar = 0, pairsOther = 0, pairsNear = 0;
    for (auto &kv : byCu) {
        std::vector<int> first;
        for (int i : kv.second)
            if (h[i].t0 - tmin < 500) first.push_back(i);
        if (first.size() == 2) {
            const int dlt = first[1] > first[0] ? first[1] - first[0] : first[0] - first[1];
            if (dlt == 256) pairsFar++;
            else if (dlt <= 8) pairsNear++;
            else pairsOther++;
        }
    }
    printf("first round: CU pairs 256 apart %d, at most 8 apart %d, otherwise %d\n", pairsFar, pairsNear, pairsOther);
    for (int i : {0, 1, 2, 7, 8, 9, 255, 256, 257, 511, 512}) printf("wg %d: cu %05x start %.1f us\n", i, cuOf(h[i]), (h[i].t0 - tmin) / 100.0);
    return 0;
}
