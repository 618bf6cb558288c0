// Lane-level building blocks shared by the pixel-grid kernels (kernels.hpp) and the superpixel
// kernels (superpixel.hpp) for gfx950 (wave64): 4-label vectors, DPP cross-lane moves, and
// prefix / suffix / whole-segment minima over the LPP lanes that hold one node's labels
// (4 consecutive labels per lane, LPP = 16, 32 or 64 for up to 64, 128, 256 labels).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ds {

constexpr int kBlock = 256;  // 4 wavefronts

template <typename T> struct Real;
template <> struct Real<float> {
    static __device__ __forceinline__ float inf() { return __builtin_huge_valf(); }
    static __device__ __forceinline__ float min(float a, float b) { return __builtin_fminf(a, b); }
    static __device__ __forceinline__ float abs(float a) { return __builtin_fabsf(a); }
};
template <> struct Real<double> {
    static __device__ __forceinline__ double inf() { return __builtin_huge_val(); }
    static __device__ __forceinline__ double min(double a, double b) { return __builtin_fmin(a, b); }
    static __device__ __forceinline__ double abs(double a) { return __builtin_fabs(a); }
};

// ---- 4-label vectors ----------------------------------------------------------
template <typename T> struct V4 { T v[4]; };

__device__ __forceinline__ V4<float> load4(const float *p)
{
    float4 x = *reinterpret_cast<const float4 *>(p);
    return {{x.x, x.y, x.z, x.w}};
}
__device__ __forceinline__ V4<double> load4(const double *p)
{
    double2 a = *reinterpret_cast<const double2 *>(p);
    double2 b = *reinterpret_cast<const double2 *>(p + 2);
    return {{a.x, a.y, b.x, b.y}};
}
__device__ __forceinline__ void store4(float *p, const V4<float> &x)
{
    *reinterpret_cast<float4 *>(p) = make_float4(x.v[0], x.v[1], x.v[2], x.v[3]);
}
__device__ __forceinline__ void store4(double *p, const V4<double> &x)
{
    *reinterpret_cast<double2 *>(p) = make_double2(x.v[0], x.v[1]);
    *reinterpret_cast<double2 *>(p + 2) = make_double2(x.v[2], x.v[3]);
}
template <typename T> __device__ __forceinline__ V4<T> splat(T x) { return {{x, x, x, x}}; }

// ---- DPP cross-lane moves (gfx9 encodings) ----------------------------------------
// Lanes whose source lane does not exist (shifted off the row / wave) or that the
// row mask disables keep `old`.
constexpr int DPP_ROW_SHL = 0x100;   // + n: lane l reads lane l+n of its 16-lane row
constexpr int DPP_ROW_SHR = 0x110;   // + n: lane l reads lane l-n of its 16-lane row
constexpr int DPP_WAVE_SHL1 = 0x130; // lane l reads lane l+1 of the wavefront
constexpr int DPP_WAVE_SHR1 = 0x138; // lane l reads lane l-1 of the wavefront
constexpr int DPP_ROW_BCAST15 = 0x142; // lane 15 of each row to every lane of the next row
constexpr int DPP_ROW_BCAST31 = 0x143; // lane 31 to every lane of rows 2 and 3

template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp(float old, float src)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old),
                                                                 __builtin_bit_cast(int, src), CTRL,
                                                                 ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double dpp(double old, double src)
{
    int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), CTRL, ROW_MASK, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float readlane(float x, int lane)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), lane));
}
__device__ __forceinline__ double readlane(double x, int lane)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}

// min over the lanes strictly below this one inside its LPP-lane segment (+inf for the first).
//
// Each step is e = min(e, t) with t = e moved across lanes.  A lane without a source keeps
// what t held before; t starts at +inf and afterwards only ever holds an earlier e of a
// lower lane of the same segment, i.e. a min over labels that belong to this lane's prefix
// anyway, so the stale value is harmless (min is idempotent) and t never needs to be reset.
template <typename T, int LPP>
__device__ __forceinline__ T excl_prefix_min(T v, int lseg)
{
    using R = Real<T>;
    const T inf = R::inf();
    T e = dpp<DPP_WAVE_SHR1>(inf, v);
    if (LPP < 64) e = (lseg == 0) ? inf : e;
    T t = inf;
    t = dpp<DPP_ROW_SHR + 1>(t, e); e = R::min(e, t);
    t = dpp<DPP_ROW_SHR + 2>(t, e); e = R::min(e, t);
    t = dpp<DPP_ROW_SHR + 4>(t, e); e = R::min(e, t);
    t = dpp<DPP_ROW_SHR + 8>(t, e); e = R::min(e, t);
    if (LPP >= 32) { t = dpp<DPP_ROW_BCAST15, 0xa>(t, e); e = R::min(e, t); }
    if (LPP >= 64) { t = dpp<DPP_ROW_BCAST31, 0xc>(t, e); e = R::min(e, t); }
    return e;
}

// min over the lanes strictly above this one inside its LPP-lane segment (+inf for the last).
template <typename T, int LPP>
__device__ __forceinline__ T excl_suffix_min(T v, int lseg, int lane)
{
    using R = Real<T>;
    const T inf = R::inf();
    T e = dpp<DPP_WAVE_SHL1>(inf, v);
    if (LPP < 64) e = (lseg == LPP - 1) ? inf : e;
    T t = inf;
    t = dpp<DPP_ROW_SHL + 1>(t, e); e = R::min(e, t);
    t = dpp<DPP_ROW_SHL + 2>(t, e); e = R::min(e, t);
    t = dpp<DPP_ROW_SHL + 4>(t, e); e = R::min(e, t);
    t = dpp<DPP_ROW_SHL + 8>(t, e); e = R::min(e, t);
    // e(l) now covers the rest of the lane's 16-lane row; add the rows above it.
    if (LPP >= 32) {
        const int row = lane >> 4;
        const T t1 = readlane(e, 16), t3 = readlane(e, 48);
        T add;
        if (LPP >= 64) {
            const T t2 = readlane(e, 32);
            const T c23 = R::min(t2, t3), c123 = R::min(t1, c23);
            add = row == 0 ? c123 : row == 1 ? c23 : row == 2 ? t3 : inf;
        } else {
            add = row == 0 ? t1 : row == 2 ? t3 : inf;
        }
        e = R::min(e, add);
    }
    return e;
}

// min over all lanes of the LPP-lane segment
template <typename T, int LPP>
__device__ __forceinline__ T segment_min(T v)
{
#pragma unroll
    for (int m = LPP / 2; m >= 1; m >>= 1) v = Real<T>::min(v, __shfl_xor(v, m));
    return v;
}

}  // namespace ds
