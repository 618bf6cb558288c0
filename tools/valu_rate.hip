// What one SIMD of this GPU issues per cycle, by instruction kind and by wavefronts per SIMD:
// every block holds one CU (100 KB of LDS), W = 4 / 8 / 12 / 16 wavefronts = 1 / 2 / 3 / 4 per
// SIMD, each running the same unrolled stream of independent vector instructions.  Prints
// shader cycles per wavefront-instruction as one wavefront sees them, and per SIMD.
//   hipcc --offload-arch=gfx950 -O2 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#define REP8(op) op(0) op(1) op(2) op(3) op(4) op(5) op(6) op(7)
#define OPS_FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %" #i "\n\t"
#define OPS_MIN(i) "v_min_f32 %" #i ", %" #i ", %8\n\t"
#define OPS_ADD(i) "v_add_f32 %" #i ", %" #i ", %8\n\t"
#define OPS_MIN3(i) "v_min3_f32 %" #i ", %" #i ", %8, %8\n\t"
#define OPS_DPP_SHR(i) "v_min_f32_dpp %" #i ", %" #i ", %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define OPS_DPP_BC(i) "v_min_f32_dpp %" #i ", %" #i ", %" #i " row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
#define OPS_MOV_WSHR(i) "v_mov_b32_dpp %" #i ", %" #i " wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define OPS_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n\t"
#define OPS_CVT(i) "v_cvt_f32_ubyte1 %" #i ", %8\n\t"

enum Kind { FMA, MIN, ADD, MIN3, DPP_SHR, DPP_BC, MOV_WSHR, CNDMASK, CVT, PK_ADD, MIN64, ADD64, MIXED, NKINDS };
const char *kKindName[NKINDS] = {"v_fma_f32", "v_min_f32", "v_add_f32", "v_min3_f32", "v_min_f32_dpp row_shr:1",
                                 "v_min_f32_dpp row_bcast:15", "v_mov_b32_dpp wave_shr:1", "v_cndmask_b32", "v_cvt_f32_ubyte1",
                                 "v_pk_add_f32", "v_min_f64", "v_add_f64", "min/add/dpp mix"};

template <int KIND> __global__ __launch_bounds__(1024) void rate_kernel(int iters, float seed, long long *cycles, float *sink)
{
    extern __shared__ float lds[];
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = seed + threadIdx.x * 0.001f + i;
    float y = seed * 0.5f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[8];
    double d[8];
    for (int i = 0; i < 8; ++i) { p[i] = f2{x[i], x[i] + 1.f}; d[i] = x[i]; }
    const f2 py = f2{y, y};
    const double dy = y;
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (KIND == FMA) asm volatile(REP8(OPS_FMA) REP8(OPS_FMA) : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(y));
        if (KIND == MIN) asm volatile(REP8(OPS_MIN) REP8(OPS_MIN) : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(y));
        if (KIND == ADD) asm volatile(REP8(OPS_ADD) REP8(OPS_ADD) : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(y));
        if (KIND == MIN3) asm volatile(REP8(OPS_MIN3) REP8(OPS_MIN3) : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(y));
        if (KIND == DPP_SHR) asm volatile("s_nop 1\n\t" REP8(OPS_DPP_SHR) REP8(OPS_DPP_SHR) : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(y));
        if (KIND == DPP_BC) asm volatile("s_nop 1\n\t" REP8(OPS_DPP_BC) REP8(OPS_DPP_BC) : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(y));
        if (KIND == MOV_WSHR) asm volatile("s_nop 1\n\t" REP8(OPS_MOV_WSHR) REP8(OPS_MOV_WSHR) : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(y));
        if (KIND == CNDMASK) asm volatile(REP8(OPS_CNDMASK) REP8(OPS_CNDMASK) : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(y) : "vcc");
        if (KIND == CVT) asm volatile(REP8(OPS_CVT) REP8(OPS_CVT) : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(y));
        if (KIND == PK_ADD) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(py));
        }
        if (KIND == MIN64) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_min_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dy));
        }
        if (KIND == ADD64) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dy));
        }
        if (KIND == MIXED)  // the proportions of the diffusion update: 1 dpp min in 6, the rest min / add alternating
            asm volatile("s_nop 1\n\t" OPS_DPP_SHR(0) OPS_MIN(1) OPS_ADD(2) OPS_MIN(3) OPS_ADD(4) OPS_MIN(5) OPS_ADD(6) OPS_MIN(7)
                         OPS_DPP_SHR(1) OPS_ADD(0) OPS_MIN(2) OPS_ADD(3) OPS_MIN(4) OPS_ADD(5) OPS_MIN(6) OPS_ADD(7)
                         : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(y));
    }
    const long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += x[i] + p[i].x + p[i].y + (float)d[i];
    if (s == 12345.678f) sink[0] = s + lds[threadIdx.x];
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND> int run(int waves, long long *dcycles, float *dsink)
{
    const int iters = 4096, blocks = 256;
    std::vector<long long> h(blocks * 16);
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(rate_kernel<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(64 * waves), 100 * 1024, 0, iters, 1.5f, dcycles, dsink);
        CHECK(hipGetLastError());
        CHECK(hipDeviceSynchronize());
    }
    CHECK(hipMemcpy(h.data(), dcycles, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
    double sum = 0, mx = 0;
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < waves; ++w) {
            sum += (double)h[b * 16 + w];
            if ((double)h[b * 16 + w] > mx) mx = (double)h[b * 16 + w];
        }
    const double per_wave = sum / (blocks * waves) / (iters * 16.0);
    std::printf("%-28s %2d wavefronts per block (%d per SIMD): %6.2f cycles per instruction as a wavefront sees it, %5.2f per SIMD (slowest wavefront %6.2f)\n",
                kKindName[KIND], waves, waves / 4, per_wave, per_wave / (waves / 4), mx / (iters * 16.0));
    return 0;
}

template <int KIND> int sweep(long long *c, float *s)
{
    for (int w = 4; w <= 16; w += 4)
        if (run<KIND>(w, c, s)) return 1;
    return 0;
}

int main()
{
    long long *dcycles;
    float *dsink;
    CHECK(hipMalloc(&dcycles, 256 * 16 * sizeof(long long)));
    CHECK(hipMalloc(&dsink, 64));
    if (sweep<FMA>(dcycles, dsink) || sweep<MIN>(dcycles, dsink) || sweep<ADD>(dcycles, dsink) || sweep<MIN3>(dcycles, dsink) ||
        sweep<DPP_SHR>(dcycles, dsink) || sweep<DPP_BC>(dcycles, dsink) || sweep<MOV_WSHR>(dcycles, dsink) ||
        sweep<CNDMASK>(dcycles, dsink) || sweep<CVT>(dcycles, dsink) || sweep<PK_ADD>(dcycles, dsink) ||
        sweep<MIN64>(dcycles, dsink) || sweep<ADD64>(dcycles, dsink) || sweep<MIXED>(dcycles, dsink))
        return 1;
    CHECK(hipFree(dcycles));
    CHECK(hipFree(dsink));
    return 0;
}
