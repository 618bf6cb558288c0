// Developer tool: HBM copy bandwidth between regions of big allocations, to see how much
// buffer placement matters on this machine.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void copy_k(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
// R = W = 4 streams, each `n` float4 long, stream k at base + k * stride (float4 units)
__global__ __launch_bounds__(256) void copy4_k(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n,
                                               size_t stride)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    }
}

static hipEvent_t ea, eb;
template <typename F> float timeit(F launch)
{
    launch();
    (void)hipEventRecord(ea);
    for (int r = 0; r < 3; ++r) launch();
    (void)hipEventRecord(eb);
    (void)hipEventSynchronize(eb);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, ea, eb);
    return ms / 3;
}

int main(int argc, char **argv)
{
    const size_t GB = 1ull << 30;
    const size_t half = (argc > 1 ? atoll(argv[1]) : 17) * GB;
    const int trials = argc > 2 ? atoi(argv[2]) : 6;
    CHK(hipEventCreate(&ea));
    CHK(hipEventCreate(&eb));
    const int blocks = 256 * 8;
    for (int t = 0; t < trials; ++t) {
        char *buf;
        CHK(hipMalloc(&buf, 2 * half));
        CHK(hipMemset(buf, 0, 2 * half));
        const size_t n = half / 16;
        float f = timeit([&] { hipLaunchKernelGGL(copy_k, dim3(blocks), dim3(256), 0, 0, (const float4 *)buf, (float4 *)(buf + half), n); });
        float b = timeit([&] { hipLaunchKernelGGL(copy_k, dim3(blocks), dim3(256), 0, 0, (const float4 *)(buf + half), (float4 *)buf, n); });
        float f4 = timeit([&] { hipLaunchKernelGGL(copy4_k, dim3(blocks), dim3(256), 0, 0, (const float4 *)buf, (float4 *)(buf + half), n / 4, n / 4); });
        float b4 = timeit([&] { hipLaunchKernelGGL(copy4_k, dim3(blocks), dim3(256), 0, 0, (const float4 *)(buf + half), (float4 *)buf, n / 4, n / 4); });
        printf("alloc %d at %p: copy lo->hi %.0f GB/s, hi->lo %.0f GB/s; 4-stream lo->hi %.0f, hi->lo %.0f\n", t, buf,
               2.0 * half / f / 1e6, 2.0 * half / b / 1e6, 2.0 * half / f4 / 1e6, 2.0 * half / b4 / 1e6);
        CHK(hipFree(buf));
    }
    return 0;
}
