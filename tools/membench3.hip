// Developer tool: do physical regions of HBM differ?  Allocates many chunks and times
// read / write / copy per chunk and between neighbouring chunks.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
using f4 = __attribute__((ext_vector_type(4))) float;

__global__ __launch_bounds__(256) void copy_k(const f4 *__restrict__ src, f4 *__restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void read_k(const f4 *__restrict__ src, f4 *__restrict__ dst, size_t n)
{
    f4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += src[i];
    if (acc.x == 123.456f) dst[0] = acc;
}
__global__ __launch_bounds__(256) void write_k(f4 *__restrict__ dst, size_t n)
{
    f4 v = {1, 2, 3, 4};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = v;
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
    const size_t GB = 1ull << 30, chunk = (argc > 1 ? atoll(argv[1]) : 4) * GB;
    const int nchunks = argc > 2 ? atoi(argv[2]) : 64;
    CHK(hipEventCreate(&ea));
    CHK(hipEventCreate(&eb));
    std::vector<char *> c;
    for (int k = 0; k < nchunks; ++k) {
        char *p = nullptr;
        if (hipMalloc(&p, chunk) != hipSuccess) break;
        (void)hipMemset(p, 0, chunk);
        c.push_back(p);
    }
    printf("%zu chunks of %zu GB\n", c.size(), chunk / GB);
    const int blocks = 2048;
    const size_t n = chunk / 16;
    for (size_t k = 0; k < c.size(); ++k) {
        float r = timeit([&] { hipLaunchKernelGGL(read_k, dim3(blocks), dim3(256), 0, 0, (const f4 *)c[k], (f4 *)c[k], n); });
        float w = timeit([&] { hipLaunchKernelGGL(write_k, dim3(blocks), dim3(256), 0, 0, (f4 *)c[k], n); });
        float s = timeit([&] { hipLaunchKernelGGL(copy_k, dim3(blocks), dim3(256), 0, 0, (const f4 *)c[k], (f4 *)(c[k] + chunk / 2), n / 2); });
        const size_t o = (k + 1) % c.size();
        float x = timeit([&] { hipLaunchKernelGGL(copy_k, dim3(blocks), dim3(256), 0, 0, (const f4 *)c[k], (f4 *)c[o], n); });
        float y = timeit([&] { hipLaunchKernelGGL(copy_k, dim3(blocks), dim3(256), 0, 0, (const f4 *)c[o], (f4 *)c[k], n); });
        printf("chunk %2zu %p: read %.0f write %.0f copy-within %.0f copy->next %.0f copy<-next %.0f GB/s\n", k, c[k],
               chunk / r / 1e6, chunk / w / 1e6, chunk / s / 1e6, 2.0 * chunk / x / 1e6, 2.0 * chunk / y / 1e6);
    }
    return 0;
}
