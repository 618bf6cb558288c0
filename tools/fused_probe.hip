// Feasibility probe (kept as the record of the experiment that preceded the library's fused
// kernel, and as a harness to time that kernel outside the library): one whole iteration (red half-sweep + black half-sweep) in ONE kernel
// that never writes the red potentials to memory.  phi of the red pixels is a pure function
// of the black pixels' phi, so a block marches down a strip of columns, keeps the red rows
// it needs in LDS, and reads phi_black / writes phi_black' only: 34 instead of 66 bytes per
// (pixel pair, label).  Interior of the image only (all four neighbours, all labels valid),
// 4K x 256 f32; checked bit for bit against the two ordinary half-sweeps on the same region.
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Xarch_device -fno-honor-nans \
//        -I stereo-vision_amd/csrc -o tools/fused_probe tools/fused_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "kernels.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

namespace {

constexpr int FR = 15;        // red pixels of a strip row that lie inside the strip
constexpr int FW = FR + 1;    // wavefronts per block: one red pixel each (one of them is the halo red)
constexpr int FC = 2 * FR;    // strip width in columns
constexpr int DL = 256;       // labels (floats per slot row)

struct Lds {
    float slot1[FW][DL];        // red row i+1, upward messages: consumed in the step that made them
    float slot0[2][FW][DL];     // red rows i, i+1 (by row parity), messages to the left
    float slot2[2][FW][DL];     // ... to the right
    float slot3[3][FW][DL];     // red rows i-1, i, i+1 (by row % 3), downward messages
};
static_assert(sizeof(Lds) == 8 * FW * DL * 4, "layout");

template <bool PREFETCH>
__global__ __launch_bounds__(64 * FW) void fused_iteration_kernel(ds::Geometry g, const float *__restrict__ phi,
                                                                  float *__restrict__ phi_new,
                                                                  long long new_plane_stride,
                                                                  const uint8_t *__restrict__ cost, float lambda,
                                                                  int c_first, int row_first, int rows_per_seg,
                                                                  int row_end)
{
    using namespace ds;
    __shared__ Lds lds;
    const int lane = threadIdx.x & 63, k = threadIdx.x >> 6;
    const int d0 = 4 * lane;
    float lam[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) lam[c] = lambda * (float)(d0 + c);
    const int c0 = c_first + blockIdx.x * FC;
    const int r0 = row_first + blockIdx.y * rows_per_seg;
    const int r1 = r0 + rows_per_seg < row_end ? r0 + rows_per_seg : row_end;

    V4<float> pre_psi[4];
    uint32_t pre_cost = 0;
    auto fetch = [&](int r) {  // the loads of red pixel k of row r, into registers
        const int p = r & 1;
        const int j = p == 0 ? c0 + 2 * k : c0 - 1 + 2 * k;
        const int jc = j >> 1, il = r - g.row0 + 1;
        const long long own = (long long)il * g.row_stride + (long long)jc * g.Dpad + d0;
        const long long nb[4] = {own + (long long)(p - 1) * g.Dpad, own - g.row_stride,
                                 own + (long long)p * g.Dpad, own + g.row_stride};
#pragma unroll
        for (int n = 0; n < 4; ++n) pre_psi[n] = ld4_nt(phi + g.plane_off[1][n ^ 2] + nb[n]);
        pre_cost = __builtin_nontemporal_load(
            reinterpret_cast<const uint32_t *>(cost + ((long long)(il - 1) * g.Wh + jc) * g.Dpad + d0));
    };
    auto produce = [&](int r) {  // red pixel k of row r, from the fetched registers
        const int p = r & 1;
        V4<float> psi[4], M[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) psi[n] = pre_psi[n];
        const V4<float> qc = unpack_cost<float>(pre_cost);
        messages_fused<64>(psi, lam, lambda, lane, lane, M);
        V4<float> out[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float avg = (qc.v[c] + (((M[0].v[c] + M[1].v[c]) + M[2].v[c]) + M[3].v[c])) * 0.25f;
#pragma unroll
            for (int n = 0; n < 4; ++n) out[n].v[c] = avg - M[n].v[c];
        }
        store4(&lds.slot0[p][k][d0], out[0]);
        store4(&lds.slot1[k][d0], out[1]);
        store4(&lds.slot2[p][k][d0], out[2]);
        store4(&lds.slot3[r % 3][k][d0], out[3]);
    };
    auto consume = [&](int i) {  // black pixel k of row i (k < FR)
        const int p = i & 1;
        const int j = p == 0 ? c0 + 2 * k + 1 : c0 + 2 * k;
        const int jc = j >> 1, il = i - g.row0 + 1;
        const int kv = k + (p == 0 ? 1 : 0);
        V4<float> psi[4], M[4];
        psi[0] = load4(&lds.slot2[p][k][d0]);                 // left red's message to the right
        psi[1] = load4(&lds.slot3[(i + 2) % 3][kv][d0]);      // upper red's (row i-1) message down
        psi[2] = load4(&lds.slot0[p][k + 1][d0]);             // right red's message to the left
        psi[3] = load4(&lds.slot1[kv][d0]);                   // lower red's (row i+1) message up
        const V4<float> qc = unpack_cost<float>(__builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(
            cost + g.cost_colour_stride + ((long long)(il - 1) * g.Wh + jc) * g.Dpad + d0)));
        messages_fused<64>(psi, lam, lambda, lane, lane, M);
        const long long own = (long long)il * g.row_stride + (long long)jc * g.Dpad + d0;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            V4<float> out;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float avg = (qc.v[c] + (((M[0].v[c] + M[1].v[c]) + M[2].v[c]) + M[3].v[c])) * 0.25f;
                out.v[c] = avg - M[n].v[c];
            }
            st4_nt(phi_new + n * new_plane_stride + own, out);
        }
    };

    fetch(r0 - 1);
    produce(r0 - 1);
    fetch(r0);
    produce(r0);
    if (PREFETCH) {
        fetch(r0 + 1);
        for (int i = r0; i < r1; ++i) {
            produce(i + 1);
            __syncthreads();
            if (i + 1 < r1) fetch(i + 2);  // in flight while the black row is computed
            if (k < FR) consume(i);
            __syncthreads();
        }
    } else {
        for (int i = r0; i < r1; ++i) {
            fetch(i + 1);
            produce(i + 1);
            __syncthreads();
            if (k < FR) consume(i);
            __syncthreads();
        }
    }
}

__global__ void fill_kernel(float *p, long long n, unsigned seed)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)(i * 2654435761u) ^ seed;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = (float)(x & 0xffff) * (1.0f / 1024.0f) - 32.0f;
    }
}
__global__ void fill_bytes_kernel(uint32_t *p, long long n)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)(i * 2246822519u);
        x ^= x >> 13; x *= 2654435761u;
        p[i] = x;
    }
}

// mismatching dwords between phi (colour 1 planes) and phi_new over rows [ra, rb), columns [ca, cb)
__global__ void compare_kernel(ds::Geometry g, const float *phi, const float *phi_new, long long new_plane_stride,
                               int ra, int rb, int ca, int cb, unsigned long long *bad)
{
    const long long per_row = (long long)g.Wh * g.Dpad;
    const long long total = (long long)(rb - ra) * per_row * 4;
    unsigned long long mine = 0;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(idx / ((long long)(rb - ra) * per_row));
        const long long rem = idx % ((long long)(rb - ra) * per_row);
        const int i = ra + (int)(rem / per_row);
        const long long in_row = rem % per_row;
        const int jc = (int)(in_row / g.Dpad);
        const int j = 2 * jc + ((i + 1) & 1);
        if (j < ca || j >= cb) continue;
        const long long off = (long long)(i - g.row0 + 1) * g.row_stride + in_row;
        const unsigned a = __float_as_uint(phi[g.plane_off[1][n] + off]);
        const unsigned b = __float_as_uint(phi_new[n * new_plane_stride + off]);
        mine += a != b;
    }
    if (mine) atomicAdd(bad, mine);
}

}  // namespace

int main(int argc, char **argv)
{
    const int spacer_gib = argc > 1 ? std::atoi(argv[1]) : 0;
    const int rows_per_seg = argc > 2 ? std::atoi(argv[2]) : 270;
    ds::Geometry g{};
    g.H = 2160; g.W = 3840; g.D = DL; g.Dpad = DL; g.Wh = g.W / 2; g.row0 = 0; g.Hb = g.H;
    g.row_stride = (long long)g.Wh * g.Dpad;
    const long long plane = (long long)(g.Hb + 2) * g.row_stride;
    for (int c = 0; c < 2; ++c)
        for (int s = 0; s < 4; ++s) g.plane_off[c][s] = (long long)(c * 4 + s) * plane;
    g.cost_colour_stride = (long long)g.Hb * g.Wh * g.Dpad;
    float *phi, *phi_new;
    uint8_t *cost;
    void *spacer = nullptr;
    unsigned long long *bad;
    CK(hipMalloc(&phi, 8 * plane * sizeof(float)));
    if (spacer_gib > 0) CK(hipMalloc(&spacer, (size_t)spacer_gib << 30));
    CK(hipMalloc(&phi_new, 4 * plane * sizeof(float)));
    CK(hipMalloc(&cost, 2 * g.cost_colour_stride));
    CK(hipMalloc(&bad, 8));
    hipLaunchKernelGGL(fill_kernel, dim3(8192), dim3(256), 0, 0, phi, 8 * plane, 12345u);
    hipLaunchKernelGGL(fill_bytes_kernel, dim3(8192), dim3(256), 0, 0, (uint32_t *)cost, 2 * g.cost_colour_stride / 4);
    CK(hipMemset(phi_new, 0, 4 * plane * sizeof(float)));
    CK(hipDeviceSynchronize());

    const int c_first = 258, row_first = 2, row_end = g.H - 2;
    const int strips = (g.W - 2 - c_first) / FC;                  // last strip's halo red stays inside the image
    const int segs = (row_end - row_first + rows_per_seg - 1) / rows_per_seg;
    const int c_end = c_first + strips * FC;
    std::printf("strips %d x segments %d = %d blocks of %d threads, LDS %zu bytes\n", strips, segs, strips * segs,
                64 * FW, sizeof(Lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const float lambda = 1.2f;
    float ms_fused = 0, ms_plain = 0;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        if (rep & 1)
            hipLaunchKernelGGL(fused_iteration_kernel<true>, dim3(strips, segs), dim3(64 * FW), 0, 0, g, phi, phi_new, plane,
                               cost, lambda, c_first, row_first, rows_per_seg, row_end);
        else
            hipLaunchKernelGGL(fused_iteration_kernel<false>, dim3(strips, segs), dim3(64 * FW), 0, 0, g, phi, phi_new, plane,
                               cost, lambda, c_first, row_first, rows_per_seg, row_end);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        CK(hipEventElapsedTime(&ms_fused, e0, e1));
        std::printf("fused iteration over the interior (%s): %.3f ms\n", (rep & 1) ? "prefetch" : "plain", ms_fused);
    }
    // the library's kernel (borders and all) over the whole image, for several segment counts
    {
        ds::FusedPlanes np;
        for (int n = 0; n < 4; ++n) np.out_off[n] = (phi_new - phi) + n * plane;
        const int strips_all = (g.W + FC - 1) / FC;
        for (int nseg : {2, 4, 6, 8, 12, 16}) {
            const int rps = (g.H + nseg - 1) / nseg;
            float ms = 0, best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL((ds::fused_iteration_kernel<float, 64, 16, false>), dim3(strips_all, (g.H + rps - 1) / rps),
                                   dim3(64 * FW), 0, 0, g, phi, np, cost, lambda, 0, g.H, rps, 0, 0);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipGetLastError());
                CK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
            }
            std::printf("library kernel, whole image, %2d segments (%d blocks): %.3f ms\n", nseg,
                        strips_all * ((g.H + rps - 1) / rps), best);
        }
        // restore what the compare below expects in phi_new
        hipLaunchKernelGGL(fused_iteration_kernel<true>, dim3(strips, segs), dim3(64 * FW), 0, 0, g, phi, phi_new, plane, cost,
                           lambda, c_first, row_first, rows_per_seg, row_end);
        CK(hipDeviceSynchronize());
    }
    // the two ordinary half-sweeps, in place (phi_new keeps the fused result of the same input)
    const int seg_pixels = 4;
    const dim3 grid(g.Wh / seg_pixels, g.H), block(ds::kBlock);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((ds::sweep_kernel<float, 64, false>), grid, block, 0, 0, g, phi, cost, lambda, 0, 0, seg_pixels, g.H, 0);
    hipLaunchKernelGGL((ds::sweep_kernel<float, 64, false>), grid, block, 0, 0, g, phi, cost, lambda, 1, 0, seg_pixels, g.H, 0);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    CK(hipEventElapsedTime(&ms_plain, e0, e1));
    CK(hipMemset(bad, 0, 8));
    hipLaunchKernelGGL(compare_kernel, dim3(4096), dim3(256), 0, 0, g, phi, phi_new, plane, row_first, row_end, c_first,
                       c_end, bad);
    unsigned long long h_bad = 0;
    CK(hipMemcpy(&h_bad, bad, 8, hipMemcpyDeviceToHost));
    const double frac = (double)(row_end - row_first) * (c_end - c_first) / ((double)g.H * g.W);
    std::printf("two half-sweeps over the whole image: %.3f ms (%.3f ms for the interior's share, %.1f %% of the image)\n",
                ms_plain, ms_plain * frac, 100 * frac);
    std::printf("fused / plain on the same pixels: %.3f   mismatching dwords in the interior: %llu\n",
                ms_fused / (ms_plain * frac), h_bad);
    return h_bad ? 1 : 0;
}
