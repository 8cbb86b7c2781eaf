// Issue rate of the packed float32 VALU instructions the lean ROIAlign backward is made of (gfx950): cycles per wave64 instruction at
// 1 .. 8 waves per SIMD, for  v_pk_fma_f32 with an SGPR-pair multiplier (the kernel's form),  with a VGPR multiplier,  and plain v_fma_f32.
// Build: hipcc -O3 --offload-arch=gfx950 pk_rate.hip -o pk_rate ; run: ./pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(512) void k(unsigned long long *out, float s, int iters) {
    f2 a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = f2{(float)threadIdx.x + i, 1.0f};
    f2 t = f2{(float)threadIdx.x, 2.0f};
    f2 sv = f2{s, s + 1.0f};            // uniform: lives in an SGPR pair for KIND 0
    f2 vv = f2{s + (float)(threadIdx.x & 1), s};   // per-lane: a VGPR pair
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (KIND == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(a[i]) : "s"(sv), "v"(t));
            else if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(vv), "v"(t));
            else if (KIND == 2) { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "s"(s), "v"(t.x)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].y) : "s"(s), "v"(t.y)); }
            else asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(a[i]) : "v"(vv), "v"(t));
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    f2 sum = f2{0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 16; ++i) sum += a[i];
    if (sum.x == 123.456f) out[1] = 1;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

int main() {
    unsigned long long *d, h[2];
    hipMalloc(&d, 16);
    const int iters = 2000;
    const char *names[4] = {"v_pk_fma_f32, SGPR-pair multiplier (op_sel_hi 0: the kernel's form)", "v_pk_fma_f32, VGPR multiplier", "2 x v_fma_f32, SGPR multiplier", "v_pk_fma_f32, VGPR multiplier with op_sel_hi 0"};
    for (int kind = 0; kind < 4; ++kind)
        for (int threads : {64, 256, 512}) {       // one block on one CU: 64 threads = 1 wave on one SIMD, 256 = 1 per SIMD, 512 = 2 per SIMD
            for (int blocks_per_cu : {1, 4}) {
                if (threads != 512 && blocks_per_cu == 4) continue;
                hipMemset(d, 0, 16);
                const int grid = 256 * blocks_per_cu;      // every CU: blocks_per_cu blocks -> 2 or 8 waves per SIMD at 512 threads
                auto launch = [&](auto kern) { hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, 0, d, 1.5f, iters); };
                hipEvent_t a0, a1; hipEventCreate(&a0); hipEventCreate(&a1);
                for (int rep = 0; rep < 2; ++rep) {
                    if (rep == 1) hipEventRecord(a0);
                    if (kind == 0) launch(k<0>); else if (kind == 1) launch(k<1>); else if (kind == 2) launch(k<2>); else launch(k<3>);
                }
                hipEventRecord(a1);
                hipDeviceSynchronize();
                float kms; hipEventElapsedTime(&kms, a0, a1);
                hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
                const double ninstr = iters * 16.0 * (kind == 2 ? 2 : 1);
                const double per = (double)h[0] / ninstr;
                const double waves_per_simd = (double)grid * threads / 64 / 1024;
                printf("%-72s %3d threads x %d blocks/CU: %.2f ticks per instruction of one wave; kernel %.1f us = %.2f ticks per wave-instruction of a SIMD (%.2f waves per SIMD)\n",
                       names[kind], threads, blocks_per_cu, per, kms * 1e3, kms * 1e3 * 2401.9 / (ninstr * waves_per_simd), waves_per_simd);
            }
        }
    // s_memtime ticks per microsecond
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, d, 1.5f, 200000); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("s_memtime: %.1f ticks per microsecond (one long launch: %llu ticks in %.1f us)\n", h[0] / (ms * 1e3), h[0], ms * 1e3);
    return 0;
}
