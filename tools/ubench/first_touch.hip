// How long does a wave's FIRST vector load take at the start of a launch, as a function of its rank among the waves of its CU and of the
// distance between the addresses the waves touch?  (tools/roi_lean_stamps.py: in the lean ROIAlign backward the row indices - 256 contiguous
// bytes per wave, one 4-KiB plan node apart - come back 1.5 us after issue for the first 12 waves of a CU and up to 6 us for the 28th.)
// Grid: 1792 blocks x 256 threads = 28 waves on each of 256 CUs, all resident.  Build: hipcc -O3 --offload-arch=gfx950 first_touch.hip -o first_touch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <map>

__global__ __launch_bounds__(256, 8) void k(const int *buf, size_t stride_ints, unsigned long long *out, int spin) {
    const int lane = threadIdx.x & 63;
    const size_t wid = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    const int v = buf[wid * stride_ints + lane];
    asm volatile("s_waitcnt vmcnt(0)" ::"v"(v) : "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    // stay resident like the real kernel's waves do (so that every wave of the launch is on the chip at once)
    unsigned long long t2 = t1;
    while (t2 - t1 < (unsigned long long)spin) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2)::"memory");
    if (lane == 0) {
        out[wid * 3 + 0] = t0; out[wid * 3 + 1] = t1 - t0 + (v == 123456789 ? 1 : 0);
        out[wid * 3 + 2] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
    }
}

int main() {
    const int blocks = 1792, waves = blocks * 4;
    const size_t max_stride = 1 << 21;                      // bytes
    int *buf; unsigned long long *out;
    hipMalloc(&buf, (size_t)waves * max_stride + 4096);
    hipMalloc(&out, (size_t)waves * 3 * 8);
    hipMemset(buf, 0, (size_t)waves * max_stride + 4096);
    std::vector<unsigned long long> h(waves * 3);
    for (size_t stride : {(size_t)256, (size_t)1024, (size_t)4096, (size_t)16384, (size_t)65536, (size_t)(1 << 21)}) {
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, buf, stride / 4, out, 24000);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
        // rank of a wave's start among the waves of its CU
        std::map<unsigned long long, std::vector<std::pair<unsigned long long, unsigned long long>>> cus;
        for (int w = 0; w < waves; ++w) {
            const unsigned long long hw = h[w * 3 + 2];
            const unsigned long long key = ((hw >> 32) & 15) << 16 | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15);
            cus[key].push_back({h[w * 3], h[w * 3 + 1]});
        }
        std::vector<std::vector<double>> by_rank(32);
        for (auto &kv : cus) {
            std::sort(kv.second.begin(), kv.second.end());
            for (size_t i = 0; i < kv.second.size() && i < 32; ++i) by_rank[i].push_back(kv.second[i].second / 2400.0);
        }
        printf("stride %8zu B (%zu CUs):", stride, cus.size());
        for (int r = 0; r < 28; r += 3) {
            auto &v = by_rank[r];
            if (v.empty()) continue;
            std::sort(v.begin(), v.end());
            printf("  rank %2d: %.2f us", r, v[v.size() / 2]);
        }
        printf("\n");
    }
    return 0;
}
