// Developer tool: 4-read + 4-write stream kernel with the streams placed in chosen 4 GiB
// chunks, to test whether spreading streams over the three 96 GiB regions of HBM matters.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

using f4 = __attribute__((ext_vector_type(4))) float;
struct Ptrs { const f4 *r[4]; f4 *w[4]; };

__global__ __launch_bounds__(256) void copy_k(const f4 *__restrict__ src, f4 *__restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
// block-contiguous 64 KiB chunks per stream, like the sweep: each wave moves 1 KiB per stream per step
__global__ __launch_bounds__(256) void stream8_k(Ptrs p, size_t n)
{
    const size_t per_block = 4096;  // float4 per block-chunk = 64 KiB
    for (size_t c = blockIdx.x; c * per_block < n; c += gridDim.x) {
        for (size_t k = threadIdx.x; k < per_block; k += 256) {
            const size_t i = c * per_block + k;
            f4 a = p.r[0][i], b = p.r[1][i], cc = p.r[2][i], d = p.r[3][i];
            f4 s = a + b + cc + d;
            p.w[0][i] = s - a; p.w[1][i] = s - b; p.w[2][i] = s - cc; p.w[3][i] = s - d;
        }
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
    const size_t GB = 1ull << 30, chunk = 4 * GB;
    (void)hipEventCreate(&ea);
    (void)hipEventCreate(&eb);
    std::vector<char *> c;
    for (int k = 0; k < 68; ++k) {
        char *p = nullptr;
        if (hipMalloc(&p, chunk) != hipSuccess) break;
        (void)hipMemset(p, 0, chunk);
        c.push_back(p);
    }
    const size_t n = chunk / 16;
    printf("%zu chunks; adjacent copy GB/s:", c.size());
    std::vector<int> region(c.size(), 0);
    std::vector<double> rate(c.size(), 0.0);
    for (size_t k = 0; k + 1 < c.size(); ++k) {
        float x = timeit([&] { hipLaunchKernelGGL(copy_k, dim3(2048), dim3(256), 0, 0, (const f4 *)c[k], (f4 *)c[k + 1], n); });
        rate[k] = 2.0 * chunk / x / 1e6;
        printf(" %.0f", rate[k]);
    }
    std::vector<double> sorted(rate.begin(), rate.end() - 1);
    std::sort(sorted.begin(), sorted.end());
    const double median = sorted[sorted.size() / 2];
    for (size_t k = 0; k + 1 < c.size(); ++k) region[k + 1] = region[k] + (rate[k] > 1.025 * median ? 1 : 0);
    printf("\nregion of each chunk:");
    for (size_t k = 0; k < c.size(); ++k) printf(" %d", region[k]);
    printf("\n");
    auto pick = [&](int reg, int idx) -> char * {  // idx-th chunk of a region, skipping the boundary chunks
        int seen = 0;
        for (size_t k = 0; k < c.size(); ++k)
            if (region[k] == reg && (k == 0 || region[k - 1] == reg) && (k + 1 == c.size() || region[k + 1] == reg))
                if (seen++ == idx) return c[k];
        return c[0];
    };
    auto run = [&](const char *name, int rr[4], int ri[4], int wr[4], int wi[4]) {
        Ptrs p;
        for (int k = 0; k < 4; ++k) { p.r[k] = (const f4 *)pick(rr[k], ri[k]); p.w[k] = (f4 *)pick(wr[k], wi[k]); }
        float ms = timeit([&] { hipLaunchKernelGGL(stream8_k, dim3(2048), dim3(256), 0, 0, p, n); });
        printf("%-44s %.0f GB/s\n", name, 8.0 * chunk / ms / 1e6);
    };
    { int rr[4] = {0, 0, 0, 0}, ri[4] = {0, 1, 2, 3}, wr[4] = {0, 0, 0, 0}, wi[4] = {4, 5, 6, 7}; run("all in region 0", rr, ri, wr, wi); }
    { int rr[4] = {1, 1, 1, 1}, ri[4] = {0, 1, 2, 3}, wr[4] = {1, 1, 1, 1}, wi[4] = {4, 5, 6, 7}; run("all in region 1", rr, ri, wr, wi); }
    { int rr[4] = {0, 0, 0, 0}, ri[4] = {0, 1, 2, 3}, wr[4] = {1, 1, 1, 1}, wi[4] = {0, 1, 2, 3}; run("reads region 0, writes region 1", rr, ri, wr, wi); }
    { int rr[4] = {1, 1, 1, 1}, ri[4] = {0, 1, 2, 3}, wr[4] = {0, 0, 0, 0}, wi[4] = {0, 1, 2, 3}; run("reads region 1, writes region 0", rr, ri, wr, wi); }
    { int rr[4] = {0, 0, 0, 0}, ri[4] = {0, 1, 2, 3}, wr[4] = {2, 2, 2, 2}, wi[4] = {0, 1, 2, 3}; run("reads region 0, writes region 2", rr, ri, wr, wi); }
    { int rr[4] = {0, 0, 1, 1}, ri[4] = {0, 1, 0, 1}, wr[4] = {0, 0, 1, 1}, wi[4] = {2, 3, 2, 3}; run("reads 0,0,1,1 writes 0,0,1,1", rr, ri, wr, wi); }
    { int rr[4] = {0, 1, 2, 0}, ri[4] = {0, 0, 0, 1}, wr[4] = {1, 2, 0, 1}, wi[4] = {1, 1, 2, 2}; run("reads 0,1,2,0 writes 1,2,0,1", rr, ri, wr, wi); }
    { int rr[4] = {0, 1, 2, 0}, ri[4] = {0, 0, 0, 1}, wr[4] = {0, 1, 2, 1}, wi[4] = {2, 1, 1, 2}; run("reads 0,1,2,0 writes 0,1,2,1", rr, ri, wr, wi); }
    return 0;
}
