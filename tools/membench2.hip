// Developer tool: write / copy bandwidth by store cache policy.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
using f4 = __attribute__((ext_vector_type(4))) float;

template <int MODE> __device__ __forceinline__ void st(f4 *p, f4 v)
{
    if (MODE == 0) *p = v;
    else if (MODE == 1) __builtin_nontemporal_store(v, p);
    else if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else if (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    else if (MODE == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
}
template <int MODE> __device__ __forceinline__ f4 ld(const f4 *p)
{
    if (MODE == 1) return __builtin_nontemporal_load(p);
    return *p;
}
template <int MODE> __global__ __launch_bounds__(256) void write_k(f4 *dst, size_t n)
{
    f4 v = {1, 2, 3, 4};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) st<MODE>(dst + i, v);
}
template <int MODE, int LMODE> __global__ __launch_bounds__(256) void copy_k(const f4 *src, f4 *dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        st<MODE>(dst + i, ld<LMODE>(src + i));
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
    const size_t GB = 1ull << 30, half = (argc > 1 ? atoll(argv[1]) : 8) * GB, n = half / 16;
    const int blocks = 256 * (argc > 2 ? atoi(argv[2]) : 8);
    CHK(hipEventCreate(&ea));
    CHK(hipEventCreate(&eb));
    char *buf;
    CHK(hipMalloc(&buf, 2 * half));
    CHK(hipMemset(buf, 0, 2 * half));
    f4 *lo = (f4 *)buf, *hi = (f4 *)(buf + half);
    const char *names[] = {"plain", "nt", "sc1", "sc0sc1", "sc0", "sc1nt"};
#define W(M) printf("write %-7s %.0f GB/s\n", names[M], half / timeit([&] { hipLaunchKernelGGL(write_k<M>, dim3(blocks), dim3(256), 0, 0, hi, n); }) / 1e6);
    W(0) W(1) W(2) W(3) W(4) W(5)
#define C(M, L) printf("copy st=%-7s ld=%s %.0f GB/s\n", names[M], L ? "nt" : "plain", 2.0 * half / timeit([&] { hipLaunchKernelGGL((copy_k<M, L>), dim3(blocks), dim3(256), 0, 0, lo, hi, n); }) / 1e6);
    C(0, 0) C(1, 0) C(2, 0) C(3, 0) C(5, 0) C(0, 1) C(1, 1) C(2, 1) C(5, 1)
    return 0;
}
