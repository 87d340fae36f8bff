// Which CUs a stream created with hipExtStreamCreateWithCUMask uses: for a few masks, the set of (XCC, SE, SH, CU) the
// workgroups of a launch ran on.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/cu_mask_probe tools/cu_mask_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>

struct Rec {
    unsigned hwid, xcc;
};

__global__ __launch_bounds__(256) void probe(Rec *out, int spinTicks)
{
    const unsigned long long t0 = wall_clock64();
    unsigned hwid = 0, xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    while (wall_clock64() - t0 < (unsigned long long)spinTicks) __builtin_amdgcn_s_sleep(16);
    if (threadIdx.x == 0) out[blockIdx.x] = Rec{hwid, xcc};
}

static void run(const char *name, const std::vector<uint32_t> &mask, Rec *d, int n)
{
    printf("%s ...\n", name);
    hipStream_t s;
    hipError_t e = mask.empty() ? hipStreamCreate(&s) : hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
    if (e != hipSuccess) {
        printf("%s: stream creation failed: %s\n", name, hipGetErrorString(e));
        return;
    }
    hipLaunchKernelGGL(probe, dim3(n), dim3(256), 0, s, d, 500);
    e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
        printf("%s: %s\n", name, hipGetErrorString(e));
        return;
    }
    std::vector<Rec> h(n);
    hipMemcpy(h.data(), d, sizeof(Rec) * n, hipMemcpyDeviceToHost);
    std::map<unsigned, int> perXcc;
    std::map<unsigned, int> byCu;
    for (auto &r : h) {
        const unsigned cu = ((r.xcc & 0xf) << 12) | (((r.hwid >> 13) & 7) << 8) | (((r.hwid >> 12) & 1) << 4) | ((r.hwid >> 8) & 0xf);
        byCu[cu]++;
    }
    for (auto &kv : byCu) perXcc[kv.first >> 12]++;
    printf("%s: %zu distinct CUs; per XCC:", name, byCu.size());
    for (auto &kv : perXcc) printf(" %u:%d", kv.first, kv.second);
    printf("\n");
    if (byCu.size() <= 40) {
        printf("   ");
        for (auto &kv : byCu) printf(" %05x", kv.first);
        printf("\n");
    }
    hipStreamDestroy(s);
}

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int n = 4096;
    Rec *d;
    hipMalloc(&d, sizeof(Rec) * n);
    run("no mask", {}, d, n);
    run("bits 0-15", {0x0000ffffu, 0, 0, 0, 0, 0, 0, 0}, d, n);
    run("bits 16-255", {0xffff0000u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u}, d, n);
    // several streams of either mask alive at once, as a library with a few contexts would hold them
    {
        const std::vector<uint32_t> small{0x0000ffffu, 0, 0, 0, 0, 0, 0, 0}, big{0xffff0000u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u};
        hipStream_t st[8];
        for (int i = 0; i < 8; i++) {
            const auto &m = i & 1 ? big : small;
            const hipError_t e = hipExtStreamCreateWithCUMask(&st[i], (uint32_t)m.size(), m.data());
            printf("stream %d: %s\n", i, hipGetErrorString(e));
        }
        for (int rep = 0; rep < 3; rep++)
            for (int i = 0; i < 8; i++) hipLaunchKernelGGL(probe, dim3(512), dim3(256), 0, st[i], d, 200);
        printf("sync: %s\n", hipGetErrorString(hipDeviceSynchronize()));
        for (int i = 0; i < 8; i++) printf("destroy %d: %s\n", i, hipGetErrorString(hipStreamDestroy(st[i])));
    }
    run("bits 0-15 again", {0x0000ffffu, 0, 0, 0, 0, 0, 0, 0}, d, n);
    return 0;
}
