// Microbenchmark: 4 read streams + 4 write streams, 32 B per lane per stream, as the f64
// sweep moves them.  Pattern A: a lane's 32 B are contiguous (two 16 B instructions, each
// touching every other 16 B of the line).  Pattern B: each 16 B instruction is contiguous
// across the wavefront (lane l takes bytes [16l,16l+16) and [1024+16l, 1024+16l+16) of a 2 KB row).
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/f64_pattern tools/f64_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

struct Ptrs { const double2 *in[4]; double2 *out[4]; };

template <int PATTERN>
__global__ __launch_bounds__(256) void mover(Ptrs p, long long rows)  // a row = 64 lanes * 32 B = 2 KB
{
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6;
    if (wave >= rows) return;
    const long long base = wave * 128;  // double2 units per row
    const long long o0 = PATTERN == 0 ? base + 2 * lane : base + lane;
    const long long o1 = PATTERN == 0 ? base + 2 * lane + 1 : base + 64 + lane;
    double2 a[4], b[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { a[k] = p.in[k][o0]; b[k] = p.in[k][o1]; }
    double2 sa = {a[0].x + a[1].x + a[2].x + a[3].x, a[0].y + a[1].y + a[2].y + a[3].y};
    double2 sb = {b[0].x + b[1].x + b[2].x + b[3].x, b[0].y + b[1].y + b[2].y + b[3].y};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        p.out[k][o0] = double2{sa.x - a[k].x, sa.y - a[k].y};
        p.out[k][o1] = double2{sb.x - b[k].x, sb.y - b[k].y};
    }
}

int main()
{
    const long long plane = 4ll << 30;  // bytes per stream: 4 GiB -> 32 GiB moved per launch
    const long long rows = plane / 2048;
    char *rd, *spacer, *wr;
    CK(hipMalloc(&rd, 4 * plane));
    CK(hipMalloc(&spacer, 80ll << 30));
    CK(hipMalloc(&wr, 4 * plane));
    CK(hipMemset(rd, 0, 4 * plane));
    CK(hipMemset(wr, 0, 4 * plane));
    // interleave like the sweep: every launch reads and writes in both allocations
    Ptrs p;
    for (int k = 0; k < 4; ++k) {
        p.in[k] = (const double2 *)((k & 1 ? wr : rd) + (k >> 1) * plane);
        p.out[k] = (double2 *)((k & 1 ? wr : rd) + (2 + (k >> 1)) * plane);
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int grid = (int)((rows * 64 + 255) / 256);
    for (int rep = 0; rep < 3; ++rep)
        for (int pat = 0; pat < 2; ++pat) {
            CK(hipEventRecord(e0));
            for (int it = 0; it < 5; ++it) {
                if (pat == 0) hipLaunchKernelGGL(mover<0>, dim3(grid), dim3(256), 0, 0, p, rows);
                else hipLaunchKernelGGL(mover<1>, dim3(grid), dim3(256), 0, 0, p, rows);
            }
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            std::printf("pattern %c: %.3f ms per launch, %.2f TB/s\n", pat ? 'B' : 'A', ms / 5,
                        8.0 * plane / (ms / 5 * 1e-3) / 1e12);
        }
    return 0;
}
