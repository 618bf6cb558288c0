// Device code of the diffusion_stereo hot path for gfx950 (MI355X, wave64).
//
// Work decomposition (all kernels): labels d run across lanes, 4 consecutive
// labels per lane, so one pixel occupies a segment of LPP lanes (LPP = 16, 32 or
// 64 for D <= 64, 128, 256) and a wavefront holds 64/LPP pixels at a time.  All
// phi rows are stored d-fastest, so every global access is a 16-byte (f32) or
// 32-byte (f64) per-lane vector and a wavefront instruction moves one contiguous
// kilobyte.  The min-plus products over d are prefix / suffix minima evaluated
// with DPP cross-lane moves; they are HBM-bound scans, not contractions, so no
// MFMA is involved.
//
// Device layout of phi ("colour split"): phi[colour][slot][band row + halo][jc][dp]
//   colour = (i + j) & 1, jc = j >> 1 (pixels of one colour are contiguous),
//   dp in [0, Dpad), Dpad = D rounded up to a multiple of 4.
// Entries the reference never touches (d > j, and the padding dp >= D) hold +inf,
// which makes every masked min a plain min: +inf only ever meets finite addends
// (x - lambda*d, x + lambda*d), never another infinity, on paths that are stored.
//
// Update rule (reference: diffusion_stereo/src/diffusion_graph.rs:135-141,152-169,
// 212-251,302-341), per pixel t, slot n with neighbour t' and reverse slot n':
//   psi_n    = phi[t'][n'][.]
//   M_n(d)   = min over allowed d' of ( lambda*|d-d'| + psi_n(d') )
//   phi[t][n][d] <- ( q_t(d) + sum_m M_m(d) ) / |N(t)| - M_n(d)
// "allowed" = edge_exists: d' <= d+1 for the right neighbour (slot 2), d' >= d-1
// for the left one (slot 0), no constraint vertically (slots 1, 3).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lanes.hpp"

namespace ds {

struct Geometry {
    int H, W, D;      // whole image, number of labels
    int Dpad;         // D rounded up to a multiple of 4
    int Wh;           // (W + 1) / 2: pixel slots per colour per row
    int row0, Hb;     // band: image rows [row0, row0 + Hb)
    // element offset of phi(colour, slot, buffer row il, jc, dp) relative to the base pointer =
    //   plane_off[colour][slot] + il * row_stride + jc * Dpad + dp
    // The eight planes need not be adjacent (or even in one allocation): the host places
    // them so that every launch's reads and writes spread over several regions of HBM.
    long long plane_off[2][4];
    long long row_stride;
    long long cost_colour_stride;  // bytes per colour of the packed cost: Hb * Wh * Dpad
};

// ---- the four messages of one pixel --------------------------------------------
// psi[n] = the neighbour's reverse row (4 labels of this lane; +inf where invalid or
// where the neighbour does not exist), lam[k] = fl(lambda * d_k), lam1 = lambda.
// PARTIAL: the pixel's labels fill only the lower lanes of its segment and lane_on says whether
// this lane holds labels; a lane that holds none keeps out of the minima of the lanes below it
// (the suffix minima and the label the lane below takes as its next one), and what it computes
// for itself is never used.
template <typename T, int LPP, bool PARTIAL = false>
__device__ __forceinline__ void messages(const V4<T> psi[4], const T lam[4], T lam1, int lseg, int lane,
                                         V4<T> M[4], bool lane_on = true)
{
    using R = Real<T>;
    const T inf = R::inf();
    T fwd[4], bwd[4];

    // forward part min_{d'<=d}: slots 1, 2, 3
    auto forward = [&](const V4<T> &x) {
        T p[4];
        p[0] = x.v[0] - lam[0];
#pragma unroll
        for (int k = 1; k < 4; ++k) p[k] = R::min(p[k - 1], x.v[k] - lam[k]);
        const T xf = excl_prefix_min<T, LPP>(p[3], lseg);
#pragma unroll
        for (int k = 0; k < 4; ++k) fwd[k] = R::min(p[k], xf) + lam[k];
    };
    // backward part min_{d'>=d}: slots 0, 1, 3
    auto backward = [&](const V4<T> &x) {
        T s[4];
        s[3] = x.v[3] + lam[3];
#pragma unroll
        for (int k = 2; k >= 0; --k) s[k] = R::min(s[k + 1], x.v[k] + lam[k]);
        const T xb = excl_suffix_min<T, LPP>(PARTIAL ? (lane_on ? s[0] : inf) : s[0], lseg, lane);
#pragma unroll
        for (int k = 0; k < 4; ++k) bwd[k] = R::min(s[k], xb) - lam[k];
    };

    // slot 0 (left neighbour): d' >= d-1
    {
        backward(psi[0]);
        T prev = dpp<DPP_WAVE_SHR1>(inf, psi[0].v[3]);
        if (LPP < 64) prev = (lseg == 0) ? inf : prev;
        M[0].v[0] = R::min(bwd[0], lam1 + prev);
#pragma unroll
        for (int k = 1; k < 4; ++k) M[0].v[k] = R::min(bwd[k], lam1 + psi[0].v[k - 1]);
    }
    // slots 1 and 3 (vertical): unconstrained
#pragma unroll
    for (int n = 1; n < 4; n += 2) {
        forward(psi[n]);
        backward(psi[n]);
#pragma unroll
        for (int k = 0; k < 4; ++k) M[n].v[k] = R::min(fwd[k], bwd[k]);
    }
    // slot 2 (right neighbour): d' <= d+1
    {
        forward(psi[2]);
        T next = dpp<DPP_WAVE_SHL1>(inf, PARTIAL ? (lane_on ? psi[2].v[0] : inf) : psi[2].v[0]);
        if (LPP < 64) next = (lseg == LPP - 1) ? inf : next;
#pragma unroll
        for (int k = 0; k < 3; ++k) M[2].v[k] = R::min(fwd[k], lam1 + psi[2].v[k + 1]);
        M[2].v[3] = R::min(fwd[3], lam1 + next);
    }
}

// Everything a lane needs to know about "its" pixel for one step.
struct Pixel {
    bool active;       // pixel exists and the lane's 4 labels lie inside Dpad
    bool has[4];       // neighbour slot exists (utils.rs:42-67)
    int nn;            // |N(t)|
    int j;             // column
    int d0;            // first of the lane's 4 labels
    long long own;     // element offset of (il, jc, d0) inside one (colour, slot) plane
    long long nb[4];   // same, for the neighbour read through slot n
    long long cost;    // byte offset into the packed cost of this colour
};

// i = image row, il = row inside the band's buffer (halo row above = 0), s = column of
// the row's first pixel of this colour, jc = pixel slot, d0 = first label of the lane.
__device__ __forceinline__ Pixel locate(const Geometry &g, int i, int s, int jc, int d0)
{
    Pixel p;
    const int il = i - g.row0 + 1;
    p.j = 2 * jc + s;
    p.d0 = d0;
    p.active = (p.j < g.W) && (d0 < g.Dpad);
    p.has[0] = p.j > 0;
    p.has[1] = i > 0;
    p.has[2] = p.j + 1 < g.W;
    p.has[3] = i + 1 < g.H;
    p.nn = (int)p.has[0] + (int)p.has[1] + (int)p.has[2] + (int)p.has[3];
    p.own = (long long)il * g.row_stride + (long long)jc * g.Dpad + d0;
    p.nb[0] = p.own + (long long)(s - 1) * g.Dpad;       // (i, j-1): slot jc + s - 1
    p.nb[1] = p.own - g.row_stride;                      // (i-1, j): slot jc
    p.nb[2] = p.own + (long long)s * g.Dpad;             // (i, j+1): slot jc + s
    p.nb[3] = p.own + g.row_stride;                      // (i+1, j): slot jc
    p.cost = ((long long)(il - 1) * g.Wh + jc) * g.Dpad + d0;
    return p;
}

__device__ __forceinline__ int reverse_slot(int n) { return n ^ 2; }  // utils.rs:75-83

template <typename T>
__device__ __forceinline__ void load_psi(const Geometry &g, const T *phi, int other_colour, const Pixel &p,
                                         V4<T> psi[4])
{
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        // the neighbour's labels above its own column index hold +inf: no need to fetch them
        const int jn = n == 0 ? p.j - 1 : n == 2 ? p.j + 1 : p.j;
        psi[n] = (p.active && p.has[n] && p.d0 <= jn)
                     ? load4(phi + g.plane_off[other_colour][reverse_slot(n)] + p.nb[n])
                     : splat(Real<T>::inf());
    }
}

template <typename T>
__device__ __forceinline__ V4<T> load_cost(const uint8_t *cost, const Pixel &p)
{
    V4<T> q = splat(T(0));
    if (p.active && p.d0 <= p.j) {
        const uint32_t w = *reinterpret_cast<const uint32_t *>(cost + p.cost);
#pragma unroll
        for (int k = 0; k < 4; ++k) q.v[k] = (T)((w >> (8 * k)) & 0xffu);
    }
    return q;
}

template <typename T> __device__ __forceinline__ V4<T> unpack_cost(uint32_t w)
{
    V4<T> q;
#pragma unroll
    for (int k = 0; k < 4; ++k) q.v[k] = (T)((w >> (8 * k)) & 0xffu);
    return q;
}

// f / |N(t)| as the reference divides (diffusion_graph.rs:249-250); powers of two exactly.
template <typename T> __device__ __forceinline__ T divide_by_neighbours(T f, int nn)
{
    if (nn == 4) return f * T(0.25);
    if (nn == 2) return f * T(0.5);
    if (nn == 3) return f / T(3);
    return f;
}

// ---- f32 scans with the min fused into the DPP instruction -------------------------
// v_min_f32_dpp d, d, d <ctrl>: lanes with a valid source take min(own, source), lanes
// without one (shifted off the row, or masked rows) keep their value -- exactly a min
// with a +inf fill, in one instruction.  The three prefix scans and the three suffix
// scans of a pixel advance in lock step inside one asm block, so consecutive DPP reads
// of a register are always separated by the other scans' instructions (the 2 wait
// states a DPP read needs after a VALU write of its source); the leading s_nop covers
// whatever instruction preceded the block.
#define DS_DPP6(ctrl_up, ctrl_down)                                \
    "v_min_f32_dpp %0, %0, %0 " ctrl_up " row_mask:0xf bank_mask:0xf\n\t"   \
    "v_min_f32_dpp %1, %1, %1 " ctrl_up " row_mask:0xf bank_mask:0xf\n\t"   \
    "v_min_f32_dpp %2, %2, %2 " ctrl_up " row_mask:0xf bank_mask:0xf\n\t"   \
    "v_min_f32_dpp %3, %3, %3 " ctrl_down " row_mask:0xf bank_mask:0xf\n\t" \
    "v_min_f32_dpp %4, %4, %4 " ctrl_down " row_mask:0xf bank_mask:0xf\n\t" \
    "v_min_f32_dpp %5, %5, %5 " ctrl_down " row_mask:0xf bank_mask:0xf\n\t"
#define DS_DPP3(ctrl, rows)                                         \
    "v_min_f32_dpp %0, %0, %0 " ctrl " row_mask:" rows " bank_mask:0xf\n\t" \
    "v_min_f32_dpp %1, %1, %1 " ctrl " row_mask:" rows " bank_mask:0xf\n\t" \
    "v_min_f32_dpp %2, %2, %2 " ctrl " row_mask:" rows " bank_mask:0xf\n\t"

// up[0..2]: values shifted one lane up already (exclusive prefix inputs);
// down[0..2]: values shifted one lane down (exclusive suffix inputs).  On return up[]
// holds the exclusive prefix minima over the LPP-lane segment and down[] the suffix
// minima over the lane's own 16-lane row (the caller adds the rows above).
template <int LPP>
__device__ __forceinline__ void fused_scans6(float up[3], float down[3])
{
    if (LPP >= 64) {
        asm("s_nop 1\n\t" DS_DPP6("row_shr:1", "row_shl:1") DS_DPP6("row_shr:2", "row_shl:2")
                DS_DPP6("row_shr:4", "row_shl:4") DS_DPP6("row_shr:8", "row_shl:8")
                    DS_DPP3("row_bcast:15", "0xa") DS_DPP3("row_bcast:31", "0xc")
            : "+v"(up[0]), "+v"(up[1]), "+v"(up[2]), "+v"(down[0]), "+v"(down[1]), "+v"(down[2]));
    } else if (LPP >= 32) {
        asm("s_nop 1\n\t" DS_DPP6("row_shr:1", "row_shl:1") DS_DPP6("row_shr:2", "row_shl:2")
                DS_DPP6("row_shr:4", "row_shl:4") DS_DPP6("row_shr:8", "row_shl:8")
                    DS_DPP3("row_bcast:15", "0xa")
            : "+v"(up[0]), "+v"(up[1]), "+v"(up[2]), "+v"(down[0]), "+v"(down[1]), "+v"(down[2]));
    } else {
        asm("s_nop 1\n\t" DS_DPP6("row_shr:1", "row_shl:1") DS_DPP6("row_shr:2", "row_shl:2")
                DS_DPP6("row_shr:4", "row_shl:4") DS_DPP6("row_shr:8", "row_shl:8")
            : "+v"(up[0]), "+v"(up[1]), "+v"(up[2]), "+v"(down[0]), "+v"(down[1]), "+v"(down[2]));
    }
}

// messages() for f32 with the six scans fused as above.  Same values, bit for bit.
template <int LPP>
__device__ __forceinline__ void messages_fused(const V4<float> psi[4], const float lam[4], float lam1,
                                               int lseg, int lane, V4<float> M[4])
{
    using R = Real<float>;
    const float inf = R::inf();
    // in-lane prefix minima of a = psi - lam for slots 1, 2, 3; suffix minima of
    // b = psi + lam for slots 0, 1, 3
    float p[3][4], sfx[3][4];
    const int pslot[3] = {1, 2, 3}, sslot[3] = {0, 1, 3};
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const V4<float> &x = psi[pslot[t]];
        p[t][0] = x.v[0] - lam[0];
#pragma unroll
        for (int k = 1; k < 4; ++k) p[t][k] = R::min(p[t][k - 1], x.v[k] - lam[k]);
        const V4<float> &y = psi[sslot[t]];
        sfx[t][3] = y.v[3] + lam[3];
#pragma unroll
        for (int k = 2; k >= 0; --k) sfx[t][k] = R::min(sfx[t][k + 1], y.v[k] + lam[k]);
    }
    float up[3], down[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        up[t] = dpp<DPP_WAVE_SHR1>(inf, p[t][3]);
        down[t] = dpp<DPP_WAVE_SHL1>(inf, sfx[t][0]);
        if (LPP < 64) {
            up[t] = (lseg == 0) ? inf : up[t];
            down[t] = (lseg == LPP - 1) ? inf : down[t];
        }
    }
    fused_scans6<LPP>(up, down);
    if (LPP >= 32) {
        const int row = lane >> 4;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const float t1 = readlane(down[t], 16), t3 = readlane(down[t], 48);
            float add;
            if (LPP >= 64) {
                const float t2 = readlane(down[t], 32);
                const float c23 = R::min(t2, t3), c123 = R::min(t1, c23);
                add = row == 0 ? c123 : row == 1 ? c23 : row == 2 ? t3 : inf;
            } else {
                add = row == 0 ? t1 : row == 2 ? t3 : inf;
            }
            down[t] = R::min(down[t], add);
        }
    }
    float fwd[3][4], bwd[3][4];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            fwd[t][k] = R::min(p[t][k], up[t]) + lam[k];
            bwd[t][k] = R::min(sfx[t][k], down[t]) - lam[k];
        }
    float prev = dpp<DPP_WAVE_SHR1>(inf, psi[0].v[3]);
    float next = dpp<DPP_WAVE_SHL1>(inf, psi[2].v[0]);
    if (LPP < 64) {
        prev = (lseg == 0) ? inf : prev;
        next = (lseg == LPP - 1) ? inf : next;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        M[0].v[k] = R::min(bwd[0][k], lam1 + (k == 0 ? prev : psi[0].v[k > 0 ? k - 1 : 0]));
        M[1].v[k] = R::min(fwd[0][k], bwd[1][k]);
        M[2].v[k] = R::min(fwd[1][k], lam1 + (k == 3 ? next : psi[2].v[k < 3 ? k + 1 : 3]));
        M[3].v[k] = R::min(fwd[2][k], bwd[2][k]);
    }
}

// f32 takes the fused scans, f64 the generic ones (v_min_f64 has no DPP form).
template <typename T, int LPP>
__device__ __forceinline__ void messages_any(const V4<T> psi[4], const T lam[4], T lam1, int lseg, int lane,
                                             V4<T> M[4])
{
    if constexpr (sizeof(T) == 4) messages_fused<LPP>(psi, lam, lam1, lseg, lane, M);
    else messages<T, LPP>(psi, lam, lam1, lseg, lane, M);
}

// ---- stage 2: one colour's half-sweep ----------------------------------------------
// grid.y = rows, grid.x = row segments of seg_pixels same-colour pixels; the 4
// wavefronts of a block take the segment's pixel groups (64/LPP pixels) round-robin.
//
// A group is "interior" when every pixel in it has four neighbours and all D labels
// valid (and D fills the lanes exactly): then nothing needs masking and all addresses
// are a wavefront-uniform row pointer plus a 32-bit lane offset.  Other groups (image
// border, the first D columns, padded D) take the general path, which also skips the
// vectors that hold nothing but +inf (labels d > j: 3 % of a 4K x 256 volume).
//
// Every byte is touched exactly once per launch, so f32 loads and stores are non-temporal
// (measured 3 % faster than default-policy accesses at 4K x 256).
__device__ __forceinline__ V4<float> ld4_nt(const float *p)
{
    using f4 = __attribute__((ext_vector_type(4))) float;
    const f4 x = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(p));
    return {{x.x, x.y, x.z, x.w}};
}
// f64: a lane's 32 bytes arrive as two 16-byte loads whose lanes are 32 bytes apart, so each
// instruction uses half of every 128-byte line and relies on the cache to serve the other
// half to the second instruction; non-temporal accesses give that up (measured 12 % slower).
__device__ __forceinline__ V4<double> ld4_nt(const double *p) { return load4(p); }
__device__ __forceinline__ void st4_nt(float *p, const V4<float> &x)
{
    using f4 = __attribute__((ext_vector_type(4))) float;
    const f4 v = {x.v[0], x.v[1], x.v[2], x.v[3]};
    __builtin_nontemporal_store(v, reinterpret_cast<f4 *>(p));
}
__device__ __forceinline__ void st4_nt(double *p, const V4<double> &x) { store4(p, x); }

// One pixel's update on the general path: any border, any number of valid labels.
template <typename T, int LPP>
__device__ __forceinline__ void update_pixel(const Geometry &g, T *phi, const uint8_t *cost_c, int colour,
                                             const Pixel &p, const T lam[4], T lambda, int lseg, int lane)
{
    using R = Real<T>;
    V4<T> psi[4], M[4];
    load_psi(g, phi, 1 - colour, p, psi);
    const V4<T> qc = load_cost<T>(cost_c, p);
    messages_any<T, LPP>(psi, lam, lambda, lseg, lane, M);
    const int K = p.j < g.D - 1 ? p.j : g.D - 1;
    V4<T> avg;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        T acc = T(0);
#pragma unroll
        for (int n = 0; n < 4; ++n) acc += p.has[n] ? M[n].v[k] : T(0);
        avg.v[k] = divide_by_neighbours(qc.v[k] + acc, p.nn);
    }
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        if (p.active && p.has[n] && p.d0 <= K) {  // d0 > K: the vector is +inf and stays so
            V4<T> out;
#pragma unroll
            for (int k = 0; k < 4; ++k) out.v[k] = (p.d0 + k <= K) ? avg.v[k] - M[n].v[k] : R::inf();
            store4(phi + g.plane_off[colour][n] + p.own, out);
        }
    }
}

// PROBE only changes the kernel's name: launches made while ds_create times candidate
// placements of phi show up apart from the real sweeps in a profile.
template <typename T, int LPP, bool PROBE = false>
__global__ __launch_bounds__(kBlock) void sweep_kernel(Geometry g, T *__restrict__ phi,
                                                       const uint8_t *__restrict__ cost, T lambda,
                                                       int colour, int row_begin, int seg_pixels,
                                                       int split, int row_begin2)
{
    constexpr int PPW = 64 / LPP;
    constexpr int WAVES = kBlock / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane / LPP, lseg = lane % LPP, d0 = 4 * lseg;
    // grid rows [0, split) are image rows from row_begin on, the rest image rows from row_begin2
    // on (a band's first and last rows in one launch)
    const int i = (int)blockIdx.y < split ? row_begin + blockIdx.y : row_begin2 + (blockIdx.y - split);
    const int s = (i + colour) & 1;
    T lam[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) lam[k] = lambda * (T)(d0 + k);

    const uint8_t *cost_c = cost + colour * g.cost_colour_stride;

    // wavefront-uniform row pointers for the interior path
    const int il = i - g.row0 + 1;
    const long long rowbase = (long long)il * g.row_stride;
    const long long *po = g.plane_off[1 - colour], *pm = g.plane_off[colour];
    const T *src[4] = {phi + po[2] + rowbase + (long long)(s - 1) * g.Dpad,  // (i, j-1)
                       phi + po[3] + rowbase - g.row_stride,                // (i-1, j)
                       phi + po[0] + rowbase + (long long)s * g.Dpad,       // (i, j+1)
                       phi + po[1] + rowbase + g.row_stride};               // (i+1, j)
    T *dst[4] = {phi + pm[0] + rowbase, phi + pm[1] + rowbase, phi + pm[2] + rowbase, phi + pm[3] + rowbase};
    const uint8_t *cost_row = cost_c + (long long)(il - 1) * g.Wh * g.Dpad;
    const unsigned lane_off = (unsigned)(q * g.Dpad + d0);
    const bool rows_interior = i > 0 && i + 1 < g.H && g.D == 4 * LPP;

    const int seg0 = blockIdx.x * seg_pixels;
    const int groups = seg_pixels / PPW;  // host guarantees divisibility
    for (int grp = wave; grp < groups; grp += WAVES) {
        const int jc0 = seg0 + grp * PPW;
        const int jmin = 2 * jc0 + s, jmax = jmin + 2 * (PPW - 1);
        V4<T> psi[4], M[4];
        if (rows_interior && jmin >= g.D && jmax + 1 < g.W) {
            const unsigned off = (unsigned)(jc0 * g.Dpad) + lane_off;
#pragma unroll
            for (int n = 0; n < 4; ++n) psi[n] = ld4_nt(src[n] + off);
            const V4<T> qc = unpack_cost<T>(__builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(cost_row + off)));
            messages_any<T, LPP>(psi, lam, lambda, lseg, lane, M);
            V4<T> avg;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                avg.v[k] = (qc.v[k] + (((M[0].v[k] + M[1].v[k]) + M[2].v[k]) + M[3].v[k])) * T(0.25);
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                V4<T> out;
#pragma unroll
                for (int k = 0; k < 4; ++k) out.v[k] = avg.v[k] - M[n].v[k];
                st4_nt(dst[n] + off, out);
            }
        } else {
            update_pixel<T, LPP>(g, phi, cost_c, colour, locate(g, i, s, jc0 + q, d0), lam, lambda, lseg, lane);
        }
    }
}

// The reference's raster order (diffusion_graph.rs:264-269), exactly: pixel (i, j) reads its
// left and top neighbours after their update of this iteration and its right and bottom
// neighbours before theirs.  All four lie on the anti-diagonals i + j -+ 1, so the pixels of
// one anti-diagonal k = i + j are independent of one another, and updating the diagonals in
// increasing k reproduces the sequential sweep: pixels (i_begin + m, k - i_begin - m), all of
// colour k & 1.
//
// Several raster-order iterations in flight at once.  Anti-diagonal k of iteration t needs
// diagonal k - 1 of iteration t (its left / top neighbours, already updated) and diagonal k + 1
// of iteration t - 1 (its right / bottom neighbours, not yet updated in t): both carry the
// "time step" s - 1 where s = k + 2 t.  So all diagonals with the same s are independent, and a
// call of n iterations needs K + 2 (n - 1) dependent launches (K = H + W - 1 diagonals) instead
// of n K: launch s updates, in place, diagonal s - 2 t of iteration t for every t it is live in.
// Pixels see exactly the operands of the sequential sweep, so the result is the same, bit for
// bit, as one diagonal per launch.  grid.y = the live iterations (t = t_lo + blockIdx.y),
// grid.x = pixel groups of the longest diagonal.
template <typename T, int LPP>
__global__ __launch_bounds__(kBlock) void sweep_diagonals_kernel(Geometry g, T *phi, const uint8_t *__restrict__ cost,
                                                                 T lambda, int s, int t_lo)
{
    constexpr int PPW = 64 / LPP;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane / LPP, lseg = lane % LPP, d0 = 4 * lseg;
    const int k = s - 2 * (t_lo + (int)blockIdx.y);  // host guarantees 0 <= k <= H + W - 2
    const int i_begin = k - (g.W - 1) > 0 ? k - (g.W - 1) : 0;
    const int i_end = k < g.H - 1 ? k : g.H - 1;  // inclusive
    const int count = i_end - i_begin + 1;
    const int m0 = (blockIdx.x * (kBlock / 64) + wave) * PPW;
    if (m0 >= count) return;  // whole wavefront beyond this diagonal
    T lam[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) lam[c] = lambda * (T)(d0 + c);
    const int colour = k & 1;
    const uint8_t *cost_c = cost + colour * g.cost_colour_stride;
    const int m = m0 + q;
    const bool in_range = m < count;
    const int i = in_range ? i_begin + m : i_begin;
    const int j = in_range ? k - i : 2 * g.Wh + 1;  // past the last column: locate() marks it inactive
    update_pixel<T, LPP>(g, phi, cost_c, colour, locate(g, i, j & 1, j >> 1, d0), lam, lambda, lseg, lane);
}

// ---- a whole iteration in one kernel: the first colour's potentials never reach memory ---
// Under the red-black schedule phi of the colour-0 ("red") pixels is a pure function of the
// colour-1 ("black") pixels' phi, and the black update reads nothing but the red result:
//     phi_red = F(phi_black),   phi_black' = G(phi_red).
// This kernel evaluates G(F(.)) for a strip of columns while marching down the rows: per
// (pixel pair, label) it reads 4 black rows and writes 4, instead of reading 8 and writing 8
// as the two half-sweeps do.  The black result goes to a second set of colour-1 planes
// (np.out_off): blocks run in any order, so the input must stay intact.  The arithmetic per
// pixel is the half-sweeps' own (same helpers, same expression order), so the result is
// bit-identical to sweep_kernel(colour 0) followed by sweep_kernel(colour 1).
//
// A wavefront holds PPW = 64 / LPP pixel slots, a block FW wavefronts, i.e. NP = FW * PPW
// slots.  Slot kp owns the column pair (a, a + 1), a = c0 - 1 + 2 kp (c0 even): in an odd row
// its red pixel is at a and its black pixel at a + 1, in an even row the other way round.
// Both vertical neighbours of the slot's black pixel are then the slot's own red pixels of
// the rows above and below, and one horizontal neighbour is its own red pixel of the same
// row: those three messages never leave the wavefront's registers.  Only the message a red
// pixel sends to the neighbouring slot's black pixel goes through LDS (one row of labels per
// slot, two deep), and one barrier per row step orders it.  NP slots produce NP red pixels
// per row -- one of them outside the strip, for the black pixel at the strip's edge: NP /
// (NP - 1) of the phi reads -- and NP - 1 black pixels.  Step i makes the red row i + 1 (its
// loads were issued a step earlier), then the black row i.
struct FusedPlanes {
    long long out_off[4];  // element offsets (from the phi base) of the planes that receive phi_black'
};

// component-wise select (a ?: on whole V4 objects would be done through memory)
template <typename T> __device__ __forceinline__ V4<T> pick(bool c, const V4<T> &x, const V4<T> &y)
{
    V4<T> r;
#pragma unroll
    for (int k = 0; k < 4; ++k) r.v[k] = c ? x.v[k] : y.v[k];
    return r;
}

template <typename T, int NP, int DL> struct FusedLds {
    T h[2][NP][DL];
};

// The work of one block: strip `strip` (from column col_begin on) x row segment `seg` (from
// row_begin on) of the launch's region.
template <typename T, int LPP, int FW>
__device__ __forceinline__ void fused_iteration_block(const Geometry &g, T *__restrict__ phi, const FusedPlanes &np,
                                                      const uint8_t *__restrict__ cost, T lambda, int row_begin,
                                                      int row_end, int rows_per_seg, int col_begin, int col_end,
                                                      int strip, int seg, int store_first)
{
    using R = Real<T>;
    constexpr int PPW = 64 / LPP, NP = FW * PPW, NB = NP - 1, FC = 2 * NB, DL = 4 * LPP;
    __shared__ FusedLds<T, NP, DL> lds;
    const T inf = R::inf();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane / LPP, lseg = lane % LPP, d0 = 4 * lseg;
    const int kp = wave * PPW + q;
    T lam[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) lam[c] = lambda * (T)(d0 + c);
    const int c0 = col_begin + strip * FC;  // col_begin even: strips may start anywhere (launches over a part of the image)
    const int a = c0 - 1 + 2 * kp;  // first column of the slot's pair (odd)
    const int r0 = row_begin + seg * rows_per_seg;
    const int r1 = r0 + rows_per_seg < row_end ? r0 + rows_per_seg : row_end;
    // col_end: columns from there on belong to another launch (the last strip may reach beyond)
    const bool cols_interior = g.D == DL && c0 - 1 >= g.D && c0 + FC + 1 < g.W && c0 + FC <= col_end;
    const uint8_t *cost_red = cost, *cost_black = cost + g.cost_colour_stride;

    // what fetch() keeps for produce(): the four neighbour rows and the packed cost word (the
    // pixel's geometry is cheap to recompute and would cost a dozen registers to keep)
    V4<T> pre_psi[4];
    uint32_t pre_cost = 0;
    auto red_pixel = [&](int r) {
        const int p = r & 1;
        const int j = p ? a : a + 1;
        Pixel px = locate(g, r, p, j >> 1, d0);
        px.active = px.active && j >= 0;
        return px;
    };
    auto fetch = [&](int r) {  // loads of the slot's red pixel of row r
        if (r < 0 || r >= g.H) return;
        const int p = r & 1;
        const int j = p ? a : a + 1;
        const int jc = j >> 1, il = r - g.row0 + 1;
        if (cols_interior && r >= 1 && r + 1 < g.H) {
            const long long own = (long long)il * g.row_stride + (long long)jc * g.Dpad + d0;
            const long long nb[4] = {own + (long long)(p - 1) * g.Dpad, own - g.row_stride,
                                     own + (long long)p * g.Dpad, own + g.row_stride};
#pragma unroll
            for (int n = 0; n < 4; ++n) pre_psi[n] = ld4_nt(phi + g.plane_off[1][n ^ 2] + nb[n]);
            pre_cost = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(
                cost_red + ((long long)(il - 1) * g.Wh + jc) * g.Dpad + d0));
        } else {
            const Pixel px = red_pixel(r);
            load_psi(g, phi, 1, px, pre_psi);
            pre_cost = (px.active && px.d0 <= px.j) ? *reinterpret_cast<const uint32_t *>(cost_red + px.cost) : 0u;
        }
    };
    // messages of the slot's red pixels, by the row that will use them
    V4<T> down_prev = splat(inf), down_cur = splat(inf), side_cur = splat(inf);  // rows i-1, i, i
    V4<T> down_next, side_next, up_next;                                         // row i+1
    auto produce = [&](int r) {  // red pixel of row r: registers (+ one message to LDS)
        down_next = side_next = up_next = splat(inf);
        if (r < 0 || r >= g.H) return;
        const int p = r & 1;
        V4<T> psi[4], M[4], out[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) psi[n] = pre_psi[n];
        const V4<T> qc = unpack_cost<T>(pre_cost);
        messages_any<T, LPP>(psi, lam, lambda, lseg, lane, M);
        if (cols_interior && r >= 1 && r + 1 < g.H) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const T avg = (qc.v[c] + (((M[0].v[c] + M[1].v[c]) + M[2].v[c]) + M[3].v[c])) * T(0.25);
#pragma unroll
                for (int n = 0; n < 4; ++n) out[n].v[c] = avg - M[n].v[c];
            }
        } else {
            const Pixel px = red_pixel(r);
            const int K = px.j < g.D - 1 ? px.j : g.D - 1;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                T acc = T(0);
#pragma unroll
                for (int n = 0; n < 4; ++n) acc += px.has[n] ? M[n].v[c] : T(0);
                const T avg = divide_by_neighbours(qc.v[c] + acc, px.nn);
                const bool valid = px.active && d0 + c <= K;
#pragma unroll
                for (int n = 0; n < 4; ++n) out[n].v[c] = (valid && px.has[n]) ? avg - M[n].v[c] : inf;
            }
        }
        // The last launch of a call also stores the first colour (nothing reads those planes while
        // iterations run): the block's own rows, the strip's own columns -- in an odd row the red
        // pixels of slots 1 .. NP - 1, in an even row those of slots 0 .. NP - 2 -- and, as
        // sweep_kernel does it, neither a slot towards a neighbour that does not exist nor a vector
        // without a valid label.
        if (store_first && r >= r0 && r < r1 && (p ? kp >= 1 : kp <= NB - 1)) {
            const int j = p ? a : a + 1;
            const long long own = (long long)(r - g.row0 + 1) * g.row_stride + (long long)(j >> 1) * g.Dpad + d0;
            if (cols_interior && r >= 1 && r + 1 < g.H) {
#pragma unroll
                for (int n = 0; n < 4; ++n) st4_nt(phi + g.plane_off[0][n] + own, out[n]);
            } else {
                const Pixel px = red_pixel(r);
                const int K = px.j < g.D - 1 ? px.j : g.D - 1;
                if (px.active && j < col_end && d0 <= K) {
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        if (px.has[n]) store4(phi + g.plane_off[0][n] + own, out[n]);
                }
            }
        }
        // odd row: red at a, its right neighbour is the slot's own black pixel, its left one the
        // previous slot's; even row: mirrored
        store4(&lds.h[p][kp][d0], pick(p != 0, out[0], out[2]));
        side_next = pick(p != 0, out[2], out[0]);
        up_next = out[1];
        down_next = out[3];
    };
    auto consume = [&](int i) {  // the slot's black pixel of row i
        const int p = i & 1;
        const int j = p ? a + 1 : a;
        const bool mine = p ? kp <= NB - 1 : kp >= 1;  // inside the strip
        const int jc = j >> 1, il = i - g.row0 + 1;
        const int nbr = p ? kp + 1 : kp - 1;           // slot whose red pixel is the other horizontal neighbour
        const bool wave_full = p ? (wave * PPW + PPW - 1 <= NB - 1) : (wave * PPW >= 1);
        V4<T> psi[4], M[4];
        if (cols_interior && wave_full && i >= 1 && i + 1 < g.H) {
            const V4<T> across = load4(&lds.h[p][nbr][d0]);
            psi[0] = pick(p != 0, side_cur, across);
            psi[1] = down_prev;
            psi[2] = pick(p != 0, across, side_cur);
            psi[3] = up_next;
            const V4<T> qc = unpack_cost<T>(__builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(
                cost_black + ((long long)(il - 1) * g.Wh + jc) * g.Dpad + d0)));
            messages_any<T, LPP>(psi, lam, lambda, lseg, lane, M);
            const long long own = (long long)il * g.row_stride + (long long)jc * g.Dpad + d0;
            V4<T> avg;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                avg.v[c] = (qc.v[c] + (((M[0].v[c] + M[1].v[c]) + M[2].v[c]) + M[3].v[c])) * T(0.25);
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                V4<T> out;
#pragma unroll
                for (int c = 0; c < 4; ++c) out.v[c] = avg.v[c] - M[n].v[c];
                st4_nt(phi + np.out_off[n] + own, out);
            }
        } else {
            Pixel px = locate(g, i, (i + 1) & 1, jc, d0);
            px.active = px.active && mine && j >= 0 && j < col_end;
            const V4<T> none = splat(inf);
            const bool has_across = px.active && (p ? px.has[2] : px.has[0]);
            const bool has_own = px.active && (p ? px.has[0] : px.has[2]);
            V4<T> across = none;
            if (has_across) across = load4(&lds.h[p][nbr][d0]);
            const V4<T> own_side = pick(has_own, side_cur, none);
            psi[0] = pick(p != 0, own_side, across);
            psi[1] = pick(px.active && px.has[1], down_prev, none);
            psi[2] = pick(p != 0, across, own_side);
            psi[3] = pick(px.active && px.has[3], up_next, none);
            const V4<T> qc = load_cost<T>(cost_black, px);
            messages_any<T, LPP>(psi, lam, lambda, lseg, lane, M);
            const int K = px.j < g.D - 1 ? px.j : g.D - 1;
            V4<T> avg;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                T acc = T(0);
#pragma unroll
                for (int n = 0; n < 4; ++n) acc += px.has[n] ? M[n].v[c] : T(0);
                avg.v[c] = divide_by_neighbours(qc.v[c] + acc, px.nn);
            }
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                if (px.active && px.has[n] && d0 <= K) {
                    V4<T> out;
#pragma unroll
                    for (int c = 0; c < 4; ++c) out.v[c] = (d0 + c <= K) ? avg.v[c] - M[n].v[c] : inf;
                    store4(phi + np.out_off[n] + px.own, out);
                }
            }
        }
    };
    auto rotate = [&]() {
        down_prev = down_cur;
        down_cur = down_next;
        side_cur = side_next;
    };

    fetch(r0 - 1);
    produce(r0 - 1);
    rotate();
    fetch(r0);
    produce(r0);
    rotate();
    fetch(r0 + 1);
    __syncthreads();
    for (int i = r0; i < r1; ++i) {
        produce(i + 1);                // writes this row parity's ring entry; the black row reads the other
        if (i + 1 < r1) fetch(i + 2);  // in flight while the black row is computed
        consume(i);
        rotate();
        __syncthreads();
    }
}

// One rectangular region per launch: grid (strips, row segments).
// PROBE only changes the kernel's name (placement probes at ds_create, as for sweep_kernel).
template <typename T, int LPP, int FW, bool PROBE = false>
__global__ __launch_bounds__(64 * FW) void fused_iteration_kernel(Geometry g, T *__restrict__ phi, FusedPlanes np,
                                                                  const uint8_t *__restrict__ cost, T lambda,
                                                                  int row_begin, int row_end, int rows_per_seg,
                                                                  int col_begin, int store_first)
{
    fused_iteration_block<T, LPP, FW>(g, phi, np, cost, lambda, row_begin, row_end, rows_per_seg, col_begin,
                                      0x7fffffff, (int)blockIdx.x, (int)blockIdx.y, store_first);
}

// Up to four rectangular regions in one launch (the frame around a two-iteration step's interior:
// dependent launches on a stream that shares the device with a kernel that fills it each wait
// for compute units to drain, so fewer, larger launches finish earlier).  Grid: first[n] blocks
// in one dimension; block b belongs to the region k with first[k] <= b < first[k + 1], strip
// (b - first[k]) % strips[k], row segment (b - first[k]) / strips[k].
struct FusedRegions {
    int n;
    int first[5];
    int col_begin[4], col_end[4], strips[4], row_begin[4], row_end[4], rows_per_seg[4];
};

template <typename T, int LPP, int FW>
__global__ __launch_bounds__(64 * FW) void fused_regions_kernel(Geometry g, T *__restrict__ phi, FusedPlanes np,
                                                                const uint8_t *__restrict__ cost, T lambda,
                                                                FusedRegions reg, int store_first)
{
    const int b = (int)blockIdx.x;
    int k = 0;
    while (k + 1 < reg.n && b >= reg.first[k + 1]) ++k;
    const int local = b - reg.first[k];
    fused_iteration_block<T, LPP, FW>(g, phi, np, cost, lambda, reg.row_begin[k], reg.row_end[k], reg.rows_per_seg[k],
                                      reg.col_begin[k], reg.col_end[k], local % reg.strips[k], local / reg.strips[k],
                                      store_first);
}

// ---- energy and residual ----------------------------------------------------------
// Per pixel: min_d f_t(d) + sum over slots 2,3 of min_d( phi[t][n][d] + M_n(d) )
// (diffusion_graph.rs:456-518; min_{d,d'} g evaluated as min_d of phi + M).  Each
// block writes one partial sum (f64) and one partial max |f_t(d)|.
template <typename T, int LPP>
__global__ __launch_bounds__(kBlock) void energy_kernel(Geometry g, const T *__restrict__ phi,
                                                        const uint8_t *__restrict__ cost, T lambda,
                                                        int colour, int seg_pixels,
                                                        double *__restrict__ part_sum,
                                                        double *__restrict__ part_max)
{
    using R = Real<T>;
    constexpr int PPW = 64 / LPP;
    const T inf = R::inf();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane / LPP, lseg = lane % LPP, d0 = 4 * lseg;
    const int i = g.row0 + blockIdx.y;
    const int s = (i + colour) & 1;
    T lam[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) lam[k] = lambda * (T)(d0 + k);
    const uint8_t *cost_c = cost + colour * g.cost_colour_stride;

    double sum = 0.0, worst = 0.0;
    const int seg0 = blockIdx.x * seg_pixels;
    const int groups = seg_pixels / PPW;
    for (int grp = wave; grp < groups; grp += kBlock / 64) {
        const int jc = seg0 + grp * PPW + q;
        const Pixel p = locate(g, i, s, jc, d0);
        V4<T> psi[4], M[4], own[4];
        load_psi(g, phi, 1 - colour, p, psi);
#pragma unroll
        for (int n = 0; n < 4; ++n)
            own[n] = (p.active && p.has[n]) ? load4(phi + g.plane_off[colour][n] + p.own) : splat(T(0));
        const V4<T> qc = load_cost<T>(cost_c, p);
        messages<T, LPP>(psi, lam, lambda, lseg, lane, M);
        const int K = p.j < g.D - 1 ? p.j : g.D - 1;
        T fmin = inf, e2 = inf, e3 = inf, fabsmax = T(0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool valid = p.active && (d0 + k <= K);
            T sp = T(0);
#pragma unroll
            for (int n = 0; n < 4; ++n) sp += p.has[n] ? own[n].v[k] : T(0);
            const T f = qc.v[k] - sp;
            if (valid) {
                fmin = R::min(fmin, f);
                fabsmax = p.nn > 0 ? (R::abs(f) > fabsmax ? R::abs(f) : fabsmax) : fabsmax;
                e2 = R::min(e2, own[2].v[k] + M[2].v[k]);
                e3 = R::min(e3, own[3].v[k] + M[3].v[k]);
            }
        }
        fmin = segment_min<T, LPP>(fmin);
        e2 = segment_min<T, LPP>(e2);
        e3 = segment_min<T, LPP>(e3);
        fabsmax = -segment_min<T, LPP>(-fabsmax);
        if (p.j < g.W && lseg == 0) {
            double e = (double)fmin;
            if (p.has[2]) e += (double)e2;
            if (p.has[3]) e += (double)e3;
            sum += e;
            worst = worst > (double)fabsmax ? worst : (double)fabsmax;
        }
    }
    // fixed-order reduction: lane 0 of each wavefront ends with the wavefront's total, then the
    // block's four totals are added in wavefront order
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        sum += __shfl_down(sum, m);
        const double o = __shfl_down(worst, m);
        worst = worst > o ? worst : o;
    }
    __shared__ double wave_sum[kBlock / 64], wave_max[kBlock / 64];
    if (lane == 0) {
        wave_sum[wave] = sum;
        wave_max[wave] = worst;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = wave_sum[0], m = wave_max[0];
#pragma unroll
        for (int w = 1; w < kBlock / 64; ++w) {
            s += wave_sum[w];
            m = m > wave_max[w] ? m : wave_max[w];
        }
        const long long slot = (long long)blockIdx.y * gridDim.x + blockIdx.x;
        part_sum[slot] = s;
        part_max[slot] = m;
    }
}

// Deterministic later stages: block b sums / maxes partials [b * chunk, (b + 1) * chunk) in a
// fixed order into out[b].  Run once with many blocks, then once with one block over the
// block results.
__global__ __launch_bounds__(kBlock) void reduce_partials_kernel(const double *__restrict__ part_sum,
                                                                 const double *__restrict__ part_max,
                                                                 long long n, long long chunk,
                                                                 double *__restrict__ out_sum,
                                                                 double *__restrict__ out_max)
{
    __shared__ double ssum[kBlock], smax[kBlock];
    const long long begin = (long long)blockIdx.x * chunk;
    const long long end = begin + chunk < n ? begin + chunk : n;
    double a = 0.0, b = 0.0;
    for (long long k = begin + threadIdx.x; k < end; k += kBlock) {
        a += part_sum[k];
        b = b > part_max[k] ? b : part_max[k];
    }
    ssum[threadIdx.x] = a;
    smax[threadIdx.x] = b;
    __syncthreads();
    for (int m = kBlock / 2; m >= 1; m >>= 1) {
        if ((int)threadIdx.x < m) {
            ssum[threadIdx.x] += ssum[threadIdx.x + m];
            smax[threadIdx.x] = smax[threadIdx.x] > smax[threadIdx.x + m] ? smax[threadIdx.x]
                                                                         : smax[threadIdx.x + m];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out_sum[blockIdx.x] = ssum[0];
        out_max[blockIdx.x] = smax[0];
    }
}

// ---- stage 3: label extraction -----------------------------------------------------
// mode 1: argmin_d f_t(d) (diffusion_graph.rs:456-470); mode 0: argmin_d of
// q_t(d) + sum_n M_n(d) (f_t after the pixel's vertex step, diffusion_graph.rs:302-317).
// First minimum wins (strict <), i.e. the lexicographic minimum of (value, d).
template <typename T, int LPP>
__global__ __launch_bounds__(kBlock) void labels_kernel(Geometry g, const T *__restrict__ phi,
                                                        const uint8_t *__restrict__ cost, T lambda,
                                                        int colour, int seg_pixels, int mode,
                                                        uint32_t *__restrict__ out)
{
    using R = Real<T>;
    constexpr int PPW = 64 / LPP;
    const T inf = R::inf();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane / LPP, lseg = lane % LPP, d0 = 4 * lseg;
    const int i = g.row0 + blockIdx.y;
    const int s = (i + colour) & 1;
    T lam[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) lam[k] = lambda * (T)(d0 + k);
    const uint8_t *cost_c = cost + colour * g.cost_colour_stride;

    const int seg0 = blockIdx.x * seg_pixels;
    const int groups = seg_pixels / PPW;
    for (int grp = wave; grp < groups; grp += kBlock / 64) {
        const int jc = seg0 + grp * PPW + q;
        const Pixel p = locate(g, i, s, jc, d0);
        const V4<T> qc = load_cost<T>(cost_c, p);
        V4<T> val;
        if (mode == 0) {
            V4<T> psi[4], M[4];
            load_psi(g, phi, 1 - colour, p, psi);
            messages<T, LPP>(psi, lam, lambda, lseg, lane, M);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                T acc = T(0);
#pragma unroll
                for (int n = 0; n < 4; ++n) acc += p.has[n] ? M[n].v[k] : T(0);
                val.v[k] = qc.v[k] + acc;
            }
        } else {
            V4<T> own[4];
#pragma unroll
            for (int n = 0; n < 4; ++n)
                own[n] = (p.active && p.has[n]) ? load4(phi + g.plane_off[colour][n] + p.own) : splat(T(0));
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                T sp = T(0);
#pragma unroll
                for (int n = 0; n < 4; ++n) sp += p.has[n] ? own[n].v[k] : T(0);
                val.v[k] = qc.v[k] - sp;
            }
        }
        const int K = p.j < g.D - 1 ? p.j : g.D - 1;
        T best = inf;
        int arg = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool valid = p.active && (d0 + k <= K);
            if (valid && val.v[k] < best) { best = val.v[k]; arg = d0 + k; }
        }
#pragma unroll
        for (int m = LPP / 2; m >= 1; m >>= 1) {
            const T ob = __shfl_xor(best, m);
            const int oa = __shfl_xor(arg, m);
            if (ob < best || (ob == best && oa < arg)) { best = ob; arg = oa; }
        }
        // all-infinite column cannot happen (d = 0 is always valid); reference default is 0
        if (p.j < g.W && lseg == 0)
            out[(long long)(i - g.row0) * g.W + p.j] = (arg == 0x7fffffff) ? 0u : (uint32_t)arg;
    }
}

// ---- epsilon-consistency ("crossing out") --------------------------------------------
// Reference: diffusion_stereo/src/crossing_out_graph.rs:67-183.  A vertex (t, d) starts
// alive iff d <= j and f_t(d) <= min_d f_t + eps (:78-93); an edge (t, n, d, d') is usable
// iff it exists and g_tt'(d, d') <= min_{d,d'} g_tt' + eps (:102-130).  crossing_out
// (:135-183) deletes vertices until every alive (t, d) has, towards every existing
// neighbour, a usable edge to an alive (t', d').  That greatest fixed point does not depend
// on the order of deletions, so it is reached here by repeated whole-image passes that
// update one byte mask in place (deletions are monotone: a stale read only delays one).
//
// The reference keeps an O(D^2) edge mask per pixel and slot; here "is there a usable
// edge to an alive label of t'" is one more min-plus product: with psi restricted to the
// neighbour's alive labels (+inf elsewhere),
//     exists d': usable  <=>  phi[t][n][d] + min_{d' alive} (lambda|d-d'| + psi(d')) <= gmin + eps
// evaluated with the same scans as the sweep.  Masks hold one byte per label, a dword per lane.
template <typename T> __device__ __forceinline__ V4<T> load_own(const Geometry &g, const T *phi, int colour,
                                                                const Pixel &p, int n)
{
    return (p.active && p.has[n]) ? load4(phi + g.plane_off[colour][n] + p.own) : splat(T(0));
}

// Masks are stored like the packed cost plus one halo row above and below each colour's
// rows (a band reads its neighbours' boundary rows there): alive[colour][Hb + 2][Wh][Dpad].
// mask_base() points at band row 0 of a colour, so Pixel::cost offsets apply unchanged.
__host__ __device__ __forceinline__ long long mask_colour_stride(const Geometry &g)
{
    return (long long)(g.Hb + 2) * g.Wh * g.Dpad;
}
template <typename P> __device__ __forceinline__ P *mask_base(const Geometry &g, P *alive, int colour)
{
    return alive + colour * mask_colour_stride(g) + (long long)g.Wh * g.Dpad;
}

__device__ __forceinline__ long long mask_off(const Geometry &g, const Pixel &p, int n)
{
    // byte offset, inside one colour's mask, of the lane's 4 labels at the neighbour in slot n
    // (n < 0: the pixel itself); rows are band rows (no halo)
    const long long row = (long long)g.Wh * g.Dpad;
    switch (n) {
    case 0: return p.cost + (long long)(((p.j & 1) ? 1 : 0) - 1) * g.Dpad;  // (i, j-1): slot jc + s - 1
    case 1: return p.cost - row;
    case 2: return p.cost + (long long)((p.j & 1) ? 1 : 0) * g.Dpad;        // (i, j+1): slot jc + s
    case 3: return p.cost + row;
    default: return p.cost;
    }
}

// dirty flags of the relaxation passes: one byte per (band row, row segment); `prev` holds the
// blocks that deleted something in the previous pass, `cur` those that have in this one.
__device__ __forceinline__ bool co_block_dirty(const uint8_t *prev, const uint8_t *cur, long long at, int r, int seg,
                                               int rows, int nseg)
{
    const volatile uint8_t *now = cur;
    int run = prev[at] | now[at];
    if (seg > 0) run |= prev[at - 1] | now[at - 1];
    if (seg + 1 < nseg) run |= prev[at + 1] | now[at + 1];
    if (r > 0) run |= prev[at - nseg] | now[at - nseg];
    if (r + 1 < rows) run |= prev[at + nseg] | now[at + nseg];
    return run != 0;
}

// First pass: vertex mask and gmin[t][n]; mode bit: see co_relax_kernel.
template <typename T, int LPP>
__global__ __launch_bounds__(kBlock) void co_init_kernel(Geometry g, const T *__restrict__ phi,
                                                         const uint8_t *__restrict__ cost, T lambda, T eps,
                                                         int colour, int seg_pixels,
                                                         uint8_t *__restrict__ alive, T *__restrict__ gmin)
{
    using R = Real<T>;
    constexpr int PPW = 64 / LPP;
    const T inf = R::inf();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane / LPP, lseg = lane % LPP, d0 = 4 * lseg;
    const int i = g.row0 + blockIdx.y;
    const int s = (i + colour) & 1;
    T lam[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) lam[k] = lambda * (T)(d0 + k);
    const uint8_t *cost_c = cost + colour * g.cost_colour_stride;
    uint8_t *alive_c = mask_base(g, alive, colour);
    T *gmin_c = gmin + (long long)colour * g.Hb * g.Wh * 4;

    const int seg0 = blockIdx.x * seg_pixels;
    const int groups = seg_pixels / PPW;
    for (int grp = wave; grp < groups; grp += kBlock / 64) {
        const int jc = seg0 + grp * PPW + q;
        const Pixel p = locate(g, i, s, jc, d0);
        V4<T> psi[4], M[4], own[4];
        load_psi(g, phi, 1 - colour, p, psi);
#pragma unroll
        for (int n = 0; n < 4; ++n) own[n] = load_own(g, phi, colour, p, n);
        const V4<T> qc = load_cost<T>(cost_c, p);
        messages<T, LPP>(psi, lam, lambda, lseg, lane, M);
        const int K = p.j < g.D - 1 ? p.j : g.D - 1;
        T f[4], fmin = inf, gm[4] = {inf, inf, inf, inf};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool valid = p.active && (d0 + k <= K);
            T sp = T(0);
#pragma unroll
            for (int n = 0; n < 4; ++n) sp += p.has[n] ? own[n].v[k] : T(0);
            f[k] = qc.v[k] - sp;
            if (valid) {
                fmin = R::min(fmin, f[k]);
#pragma unroll
                for (int n = 0; n < 4; ++n) gm[n] = R::min(gm[n], own[n].v[k] + M[n].v[k]);
            }
        }
        fmin = segment_min<T, LPP>(fmin);
#pragma unroll
        for (int n = 0; n < 4; ++n) gm[n] = segment_min<T, LPP>(gm[n]);
        uint32_t bits = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool valid = p.active && (d0 + k <= K);
            if (valid && f[k] <= fmin + eps) bits |= 1u << (8 * k);
        }
        if (p.active) *reinterpret_cast<uint32_t *>(alive_c + p.cost) = bits;
        if (p.j < g.W && lseg == 0) {
            T *dst = gmin_c + ((long long)(i - g.row0) * g.Wh + jc) * 4;
#pragma unroll
            for (int n = 0; n < 4; ++n) dst[n] = gm[n];
        }
    }
}

// One relaxation pass over one colour; changed[0] is set if any vertex died, changed[1] if
// some pixel has no alive vertex left.
template <typename T, int LPP>
__global__ __launch_bounds__(kBlock) void co_relax_kernel(Geometry g, const T *__restrict__ phi, T lambda,
                                                          T eps, int colour, int seg_pixels,
                                                          uint8_t *alive, const T *__restrict__ gmin,
                                                          int *__restrict__ changed, int row_first, int seg_first,
                                                          int col_lo, int col_hi, const uint8_t *dirty_prev,
                                                          uint8_t *dirty_cur, int nseg, int segs_per_block, int seg_end)
{
    using R = Real<T>;
    constexpr int PPW = 64 / LPP;
    const T inf = R::inf();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane / LPP, lseg = lane % LPP, d0 = 4 * lseg;
    const int i = row_first + blockIdx.y;  // the grid may cover a window of the band only
    const int s = (i + colour) & 1;
    // A pixel's answer can only change after a neighbour's mask has: a block whose own pixels and
    // whose four neighbouring blocks (same rows of the segments beside it, same segment of the rows
    // above and below) changed nothing in the previous pass, nor yet in this one, has nothing to do.
    // Deletions travel one or two pixels per pass, so late passes are thin fronts.
    // A block takes segs_per_block (<= 64) consecutive segments of its row.  Which of them are due is
    // looked up by as many lanes at once (one round trip to the flags, not one per segment: in a late
    // pass nearly every block returns here, and the pass costs what those round trips cost).
    __shared__ unsigned long long due_lds;
    const int seg_block = seg_first + (int)blockIdx.x * segs_per_block;
    unsigned long long due;
    if (dirty_prev) {
        if (wave == 0) {
            const int seg = seg_block + lane;
            const bool mine_due = lane < segs_per_block && seg < seg_end &&
                                  co_block_dirty(dirty_prev, dirty_cur, (long long)(i - g.row0) * nseg + seg, i - g.row0, seg, g.Hb, nseg);
            const unsigned long long ballot = __ballot(mine_due);
            if (lane == 0) due_lds = ballot;
        }
        __syncthreads();
        due = due_lds;
    } else {
        const int count = seg_end - seg_block < segs_per_block ? seg_end - seg_block : segs_per_block;
        due = count >= 64 ? ~0ull : ((1ull << (count > 0 ? count : 0)) - 1ull);
    }
    for (; due; due &= due - 1ull) {
    const int seg = seg_block + __builtin_ctzll(due);
    const long long flag_at = (long long)(i - g.row0) * nseg + seg;
    T lam[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) lam[k] = lambda * (T)(d0 + k);
    uint8_t *alive_c = mask_base(g, alive, colour);
    const uint8_t *alive_o = mask_base(g, alive, 1 - colour);
    const T *gmin_c = gmin + (long long)colour * g.Hb * g.Wh * 4;

    const int seg0 = seg * seg_pixels;
    const int groups = seg_pixels / PPW;
    bool any = false, lost = false;
    for (int grp = wave; grp < groups; grp += kBlock / 64) {
        const int jc = seg0 + grp * PPW + q;
        const Pixel p = locate(g, i, s, jc, d0);
        V4<T> psi[4], M[4], own[4];
        load_psi(g, phi, 1 - colour, p, psi);
        uint32_t mine = p.active ? *reinterpret_cast<const volatile uint32_t *>(alive_c + p.cost) : 0u;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            own[n] = load_own(g, phi, colour, p, n);
            uint32_t nb = 0;
            if (p.active && p.has[n]) nb = *reinterpret_cast<const volatile uint32_t *>(alive_o + mask_off(g, p, n));
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (!((nb >> (8 * k)) & 1u)) psi[n].v[k] = inf;  // dead labels of t' offer no edge
        }
        messages<T, LPP>(psi, lam, lambda, lseg, lane, M);
        T gm[4] = {T(0), T(0), T(0), T(0)};
        if (p.j < g.W) {
            const T *src = gmin_c + ((long long)(i - g.row0) * g.Wh + jc) * 4;
#pragma unroll
            for (int n = 0; n < 4; ++n) gm[n] = src[n];
        }
        uint32_t next = mine;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!((mine >> (8 * k)) & 1u)) continue;
            bool ok = true;
#pragma unroll
            for (int n = 0; n < 4; ++n)
                if (p.has[n]) ok = ok && (own[n].v[k] + M[n].v[k] <= gm[n] + eps);
            if (!ok) next &= ~(1u << (8 * k));
        }
        // a windowed pass changes nothing outside its window, however far its blocks reach: the
        // caller's bookkeeping (which pixels can still be affected) counts on it
        const bool inside = p.j >= col_lo && p.j <= col_hi;
        if (!inside) next = mine;
        if (p.active && next != mine) {
            *reinterpret_cast<volatile uint32_t *>(alive_c + p.cost) = next;
            any = true;
        }
        uint32_t left_alive = p.active ? next : 0u;
#pragma unroll
        for (int m = LPP / 2; m >= 1; m >>= 1) left_alive |= __shfl_xor(left_alive, m);
        if (p.j < g.W && inside && lseg == 0 && left_alive == 0u) lost = true;
    }
    if (__any(any) && lane == 0) {
        atomicOr(changed, 1);
        if (dirty_cur) dirty_cur[flag_at] = 1;
    }
    if (__any(lost) && lane == 0) atomicOr(changed + 1, 1);
    }
}

// Smallest alive label per pixel (crossing_out_graph.rs:307-327); 0xffffffff and
// *empty = 1 where a pixel has none (vertices_exist false, :196-211).
template <int LPP>
__global__ __launch_bounds__(kBlock) void co_labels_kernel(Geometry g, const uint8_t *__restrict__ alive, int colour,
                                                           int seg_pixels, uint32_t *__restrict__ out,
                                                           int *__restrict__ empty)
{
    constexpr int PPW = 64 / LPP;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane / LPP, lseg = lane % LPP, d0 = 4 * lseg;
    const int i = g.row0 + blockIdx.y;
    const int s = (i + colour) & 1;
    const uint8_t *alive_c = mask_base(g, alive, colour);
    const int seg0 = blockIdx.x * seg_pixels;
    const int groups = seg_pixels / PPW;
    for (int grp = wave; grp < groups; grp += kBlock / 64) {
        const int jc = seg0 + grp * PPW + q;
        const Pixel p = locate(g, i, s, jc, d0);
        const uint32_t bits = p.active ? *reinterpret_cast<const uint32_t *>(alive_c + p.cost) : 0u;
        int first = 0x7fffffff;
#pragma unroll
        for (int k = 3; k >= 0; --k)
            if ((bits >> (8 * k)) & 1u) first = d0 + k;
#pragma unroll
        for (int m = LPP / 2; m >= 1; m >>= 1) {
            const int o = __shfl_xor(first, m);
            first = o < first ? o : first;
        }
        if (p.j < g.W && lseg == 0) {
            out[(long long)(i - g.row0) * g.W + p.j] = first == 0x7fffffff ? 0xffffffffu : (uint32_t)first;
            if (first == 0x7fffffff) atomicOr(empty, 1);
        }
    }
}

// Alive labels per pixel: out[row][j] = how many labels the pixel has left.
template <int LPP>
__global__ __launch_bounds__(kBlock) void co_count_kernel(Geometry g, const uint8_t *__restrict__ alive, int colour,
                                                          int seg_pixels, uint32_t *__restrict__ out)
{
    constexpr int PPW = 64 / LPP;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane / LPP, lseg = lane % LPP, d0 = 4 * lseg;
    const int i = g.row0 + blockIdx.y;
    const int s = (i + colour) & 1;
    const uint8_t *alive_c = mask_base(g, alive, colour);
    const int seg0 = blockIdx.x * seg_pixels;
    const int groups = seg_pixels / PPW;
    for (int grp = wave; grp < groups; grp += kBlock / 64) {
        const int jc = seg0 + grp * PPW + q;
        const Pixel p = locate(g, i, s, jc, d0);
        const uint32_t bits = p.active ? *reinterpret_cast<const uint32_t *>(alive_c + p.cost) & 0x01010101u : 0u;
        int count = __popc(bits);
#pragma unroll
        for (int m = LPP / 2; m >= 1; m >>= 1) count += __shfl_xor(count, m);
        if (p.j < g.W && lseg == 0) out[(long long)(i - g.row0) * g.W + p.j] = (uint32_t)count;
    }
}

// alive[colour][row][jc][dp] -> host-order bytes out[row][j][d] (reference: vertices[i][j][d]).
__global__ __launch_bounds__(kBlock) void co_unpack_kernel(Geometry g, const uint8_t *__restrict__ alive,
                                                           uint8_t *__restrict__ out)
{
    const long long total = (long long)g.Hb * g.W * g.D;
    for (long long idx = (long long)blockIdx.x * kBlock + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * kBlock) {
        const int d = (int)(idx % g.D);
        const long long px = idx / g.D;
        const int j = (int)(px % g.W), r = (int)(px / g.W);
        const int colour = (g.row0 + r + j) & 1;
        out[idx] = mask_base(g, alive, colour)[((long long)r * g.Wh + (j >> 1)) * g.Dpad + d] & 1u;
    }
}

// ---- find_best_labeling: fix one label per pixel, in raster order ------------------------
// Reference: diffusion_stereo/src/crossing_out_graph.rs:284-300 (min_vertex_between_existing),
// :336-345 (cross_vertex), :349-364 (find_best_labeling).  One wavefront walks the pixels from
// `first` on.  A pixel with a single alive label keeps it (crossing out its other labels is a
// no-op, and the crossing-out that follows changes nothing).  At the first pixel with several
// alive labels it chooses the one with the smallest f_t(d) (f64 in the reference's expression
// order, the first of equals), deletes the others and stops: the caller relaxes the masks to
// their fixed point again and relaunches from the next pixel.
// state[0] = 0: walked to the end of the image (or through `budget` pixels), 1: deleted labels
// at pixel state[1] - 1, 2: pixel state[1] has no alive label; state[1] = next pixel to look at.
template <typename T>
__global__ __launch_bounds__(64) void co_fix_kernel(Geometry g, const T *__restrict__ phi,
                                                    const uint8_t *__restrict__ cost, uint8_t *alive,
                                                    long long first, long long budget,
                                                    uint32_t *__restrict__ labels, long long *__restrict__ state)
{
    const int lane = threadIdx.x, d0 = 4 * lane;
    const long long total = (long long)g.H * g.W;
    const double inf = __builtin_huge_val();
    long long idx = first;
    for (long long step = 0; idx < total && step < budget; ++step, ++idx) {
        const int i = (int)(idx / g.W), j = (int)(idx % g.W);
        const int colour = (i + j) & 1, jc = j >> 1;
        const long long in_colour = ((long long)(i - g.row0) * g.Wh + jc) * g.Dpad + d0;
        uint8_t *mask = mask_base(g, alive, colour) + in_colour;
        const uint32_t bits = d0 < g.Dpad ? *reinterpret_cast<const volatile uint32_t *>(mask) & 0x01010101u : 0u;
        int count = __popc(bits), lowest = 0x7fffffff;
#pragma unroll
        for (int k = 3; k >= 0; --k)
            if ((bits >> (8 * k)) & 1u) lowest = d0 + k;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            count += __shfl_xor(count, m);
            const int o = __shfl_xor(lowest, m);
            lowest = o < lowest ? o : lowest;
        }
        if (count == 0) {
            if (lane == 0) { state[0] = 2; state[1] = idx; }
            return;
        }
        if (count == 1) {
            if (lane == 0) labels[idx] = (uint32_t)lowest;
            continue;
        }
        // several alive labels: smallest f_t(d) = q_t(d) - sum_n phi[t][n][d], first of equals
        const bool has[4] = {j > 0, i > 0, j + 1 < g.W, i + 1 < g.H};
        const long long own = (long long)(i - g.row0 + 1) * g.row_stride + (long long)jc * g.Dpad + d0;
        double best = inf;
        int best_d = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!((bits >> (8 * k)) & 1u)) continue;
            double sum = 0.;
#pragma unroll
            for (int n = 0; n < 4; ++n)
                if (has[n]) sum += (double)phi[g.plane_off[colour][n] + own + k];
            const double f = (double)cost[colour * g.cost_colour_stride + in_colour + k] - sum;
            if (f < best) { best = f; best_d = d0 + k; }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const double of = __shfl_xor(best, m);
            const int od = __shfl_xor(best_d, m);
            if (of < best || (of == best && od < best_d)) { best = of; best_d = od; }
        }
        if (d0 < g.Dpad) {
            const uint32_t keep = (best_d >= d0 && best_d < d0 + 4) ? 1u << (8 * (best_d - d0)) : 0u;
            *reinterpret_cast<volatile uint32_t *>(mask) = keep;
        }
        if (lane == 0) {
            labels[idx] = (uint32_t)best_d;
            state[0] = 1;
            state[1] = idx + 1;
        }
        return;
    }
    if (lane == 0) { state[0] = 0; state[1] = idx; }
}

// ---- epsilon search: every penalty gap of the graph -----------------------------------
// Reference: diffusion_stereo/src/epsilon_search.rs:39-74 (create_array_of_epsilons before
// its sort and dedup).  For every vertex (t, d), d <= j: f_t(d) - min_d f_t; for every
// existing neighbour n and every existing edge (d, d'): g(d, d') - min over the pair's edges.
// One block per pixel, f64 arithmetic in the reference's expression order whatever the
// handle's dtype (phi is widened first), so the values equal the host restatement's bit for
// bit.  Layout: out[(pixel * D + d) * (1 + 4 D) + 0] = vertex gap, [+ 1 + n * D + d'] = edge
// gaps; +inf where the vertex / neighbour / edge does not exist (they sort to the end).
// Dynamic LDS: (9 * D + kBlock) doubles.
template <typename T>
__global__ __launch_bounds__(kBlock) void eps_gaps_kernel(Geometry g, const T *__restrict__ phi,
                                                          const uint8_t *__restrict__ cost, double lambda,
                                                          double *__restrict__ out)
{
    extern __shared__ double eps_lds[];
    const int D = g.D;
    double *own = eps_lds;            // [4][D]
    double *nbr = own + 4 * D;        // [4][D]: the neighbour's reverse slot
    double *f = nbr + 4 * D;          // [D]
    double *red = f + D;              // [kBlock]
    const double inf = __builtin_huge_val();
    const int tid = threadIdx.x;
    const long long pix = blockIdx.x;
    const int i = (int)(pix / g.W), j = (int)(pix % g.W);
    const int colour = (i + j) & 1, jc = j >> 1, il = i - g.row0 + 1;
    const bool has[4] = {j > 0, i > 0, j + 1 < g.W, i + 1 < g.H};
    const int ni[4] = {i, i - 1, i, i + 1}, nj[4] = {j - 1, j, j + 1, j};
    const long long own_off = (long long)il * g.row_stride + (long long)jc * g.Dpad;

    for (int e = tid; e < 4 * D; e += kBlock) {
        const int n = e / D, d = e % D;
        double a = 0., b = 0.;
        if (has[n]) {
            a = (double)phi[g.plane_off[colour][n] + own_off + d];
            const long long off = (long long)(ni[n] - g.row0 + 1) * g.row_stride + (long long)(nj[n] >> 1) * g.Dpad;
            b = (double)phi[g.plane_off[1 - colour][n ^ 2] + off + d];
        }
        own[e] = a;
        nbr[e] = b;
    }
    __syncthreads();
    auto block_min = [&](double v) {
        red[tid] = v;
        __syncthreads();
        for (int w = kBlock / 2; w >= 1; w >>= 1) {
            if (tid < w) red[tid] = fmin(red[tid], red[tid + w]);
            __syncthreads();
        }
        const double m = red[0];
        __syncthreads();
        return m;
    };
    // vertices (diffusion_graph.rs:99-122, 456-470)
    double vmin = inf;
    for (int d = tid; d < D; d += kBlock) {
        double v = inf;
        if (d <= j) {
            double sum = 0.;
#pragma unroll
            for (int n = 0; n < 4; ++n)
                if (has[n]) sum += own[n * D + d];
            const double q = (double)cost[colour * g.cost_colour_stride + ((long long)(il - 1) * g.Wh + jc) * g.Dpad + d];
            v = q - sum;
        }
        f[d] = v;
        vmin = fmin(vmin, v);
    }
    vmin = block_min(vmin);
    const long long stride = 1 + 4LL * D;
    double *dst = out + pix * D * stride;
    for (int d = tid; d < D; d += kBlock) dst[d * stride] = d <= j ? f[d] - vmin : inf;
    // edges (diffusion_graph.rs:135-141, 152-169, 480-494)
    for (int n = 0; n < 4; ++n) {
        auto exists = [&](int d, int nd) {
            if (!has[n] || d > j || nd > nj[n]) return false;
            if (n == 2 && nd > d + 1) return false;
            if (n == 0 && d > nd + 1) return false;
            return true;
        };
        auto penalty = [&](int d, int nd) {
            const double dist = d > nd ? (double)(d - nd) : (double)(nd - d);
            return lambda * dist + own[n * D + d] + nbr[n * D + nd];
        };
        double emin = inf;
        for (int e = tid; e < D * D; e += kBlock) {
            const int d = e / D, nd = e % D;
            if (exists(d, nd)) emin = fmin(emin, penalty(d, nd));
        }
        emin = block_min(emin);
        for (int e = tid; e < D * D; e += kBlock) {
            const int d = e / D, nd = e % D;
            dst[d * stride + 1 + n * D + nd] = exists(d, nd) ? penalty(d, nd) - emin : inf;
        }
    }
}

// ---- epsilon search without the array: one question about the gaps per launch ------------
// The array of create_array_of_epsilons (epsilon_search.rs:39-74) has H W D (1 + 4 D) entries
// (2.2e12 at 4K x 256) and exists only to be searched.  This kernel walks the same gaps, in the
// same f64 expression order as eps_gaps_kernel, and keeps three numbers instead of storing them:
//   result[0]  the smallest gap that is > lo (as the bits of a non-negative double; +inf if none)
//   result[1]  the largest gap
//   result[2]  how many gaps are <= lo
// Gaps are >= +0 (each is x - min over a set that holds x, in the same arithmetic), and the bit
// patterns of non-negative doubles order like unsigned integers: one atomicMin / atomicMax /
// atomicAdd per block.  mins[pixel][5] caches the pixel's vertex minimum and its four pair
// minima; have_mins = 0 computes and stores them (the first launch for a state of phi), 1 reads them.
// A block takes one pixel; a thread keeps one neighbour label nd (the low `shift` bits of its index)
// and walks d in steps of kBlock >> shift; dynamic LDS: 9 D doubles.
template <typename T>
__global__ __launch_bounds__(kBlock) void eps_query_kernel(Geometry g, const T *__restrict__ phi,
                                                           const uint8_t *__restrict__ cost, double lambda, double lo,
                                                           int shift, double *__restrict__ mins, int have_mins,
                                                           unsigned long long *__restrict__ result)
{
    extern __shared__ double eps_lds[];
    __shared__ double red[3][kBlock / 64];
    const int D = g.D;
    double *own = eps_lds;            // [4][D]
    double *nbr = own + 4 * D;        // [4][D]: the neighbour's reverse slot
    double *f = nbr + 4 * D;          // [D]
    const double inf = __builtin_huge_val();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long pix = blockIdx.x;
    const int i = (int)(pix / g.W), j = (int)(pix % g.W);
    const int colour = (i + j) & 1, jc = j >> 1, il = i - g.row0 + 1;
    const bool has[4] = {j > 0, i > 0, j + 1 < g.W, i + 1 < g.H};
    const int ni[4] = {i, i - 1, i, i + 1}, nj[4] = {j - 1, j, j + 1, j};
    const long long own_off = (long long)il * g.row_stride + (long long)jc * g.Dpad;

    for (int e = tid; e < 4 * D; e += kBlock) {
        const int n = e / D, d = e % D;
        double a = 0., b = 0.;
        if (has[n]) {
            a = (double)phi[g.plane_off[colour][n] + own_off + d];
            const long long off = (long long)(ni[n] - g.row0 + 1) * g.row_stride + (long long)(nj[n] >> 1) * g.Dpad;
            b = (double)phi[g.plane_off[1 - colour][n ^ 2] + off + d];
        }
        own[e] = a;
        nbr[e] = b;
    }
    __syncthreads();
    // min / max / sum over the block; every thread gets the result
    auto block_reduce = [&](double v, int op) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const double o = __shfl_xor(v, m);
            v = op == 0 ? fmin(v, o) : op == 1 ? fmax(v, o) : v + o;
        }
        if (lane == 0) red[op][wave] = v;
        __syncthreads();
        double r = red[op][0];
        for (int w = 1; w < kBlock / 64; ++w) r = op == 0 ? fmin(r, red[op][w]) : op == 1 ? fmax(r, red[op][w]) : r + red[op][w];
        __syncthreads();
        return r;
    };
    double next = inf, largest = 0., count = 0.;  // counts stay far below 2^53
    auto take = [&](double gap) {
        if (gap > lo) next = fmin(next, gap);
        else count += 1.;
        largest = fmax(largest, gap);
    };
    // vertices (diffusion_graph.rs:99-122, 456-470)
    double vmin = inf;
    for (int d = tid; d < D; d += kBlock) {
        double v = inf;
        if (d <= j) {
            double sum = 0.;
#pragma unroll
            for (int n = 0; n < 4; ++n)
                if (has[n]) sum += own[n * D + d];
            const double q = (double)cost[colour * g.cost_colour_stride + ((long long)(il - 1) * g.Wh + jc) * g.Dpad + d];
            v = q - sum;
        }
        f[d] = v;
        vmin = fmin(vmin, v);
    }
    if (have_mins) vmin = mins[pix * 5];
    else {
        vmin = block_reduce(vmin, 0);
        if (tid == 0) mins[pix * 5] = vmin;
    }
    __syncthreads();
    for (int d = tid; d < D; d += kBlock)
        if (d <= j) take(f[d] - vmin);
    // edges (diffusion_graph.rs:135-141, 152-169, 480-494)
    const int nd = tid & ((1 << shift) - 1), dstep = kBlock >> shift;
    for (int n = 0; n < 4; ++n) {
        if (!has[n]) continue;  // uniform over the block
        const bool nd_ok = nd < D && nd <= nj[n];
        const double psi = nd < D ? nbr[n * D + nd] : 0.;
        const int dmax = j < D - 1 ? j : D - 1;
        auto exists = [&](int d) {  // d <= j holds by the loop's limit
            if (!nd_ok) return false;
            if (n == 2 && nd > d + 1) return false;
            if (n == 0 && d > nd + 1) return false;
            return true;
        };
        auto penalty = [&](int d) {
            const double dist = d > nd ? (double)(d - nd) : (double)(nd - d);
            return lambda * dist + own[n * D + d] + psi;
        };
        double emin;
        if (have_mins) emin = mins[pix * 5 + 1 + n];
        else {
            emin = inf;
            for (int d = tid >> shift; d <= dmax; d += dstep)
                if (exists(d)) emin = fmin(emin, penalty(d));
            emin = block_reduce(emin, 0);
            if (tid == 0) mins[pix * 5 + 1 + n] = emin;
        }
        for (int d = tid >> shift; d <= dmax; d += dstep)
            if (exists(d)) take(penalty(d) - emin);
    }
    next = block_reduce(next, 0);
    largest = block_reduce(largest, 1);
    count = block_reduce(count, 2);
    if (tid == 0) {
        if (next < inf) atomicMin(result, (unsigned long long)__double_as_longlong(next));
        atomicMax(result + 1, (unsigned long long)__double_as_longlong(largest));
        atomicAdd(result + 2, (unsigned long long)count);
    }
}

// ---- crossing-out with the reference's own comparisons -------------------------------------
// co_init / co_relax decide "is this edge within epsilon of its pair's minimum" from min-plus
// scans in the handle's precision: right except when a penalty lies within rounding of the
// threshold -- which is where an epsilon taken from the array of gaps puts one on purpose.
// This kernel evaluates every comparison as the reference does (crossing_out_graph.rs:78-130), in
// f64 and its expression order, O(D^2) per pixel and pass:
//   vertex (t, d):   f_t(d) <= min_d f_t + epsilon                                   (:84-85)
//   edge, t's copy:  A_t  = (lambda |d - d'| + phi_t(d)) + phi_t'(d') <= min_t,n + epsilon   (:113-115)
//   edge, t''s copy: A_t' = (lambda |d' - d| + phi_t'(d')) + phi_t(d) <= min_t',n' + epsilon
// The two copies of one edge are rounded in different orders and may disagree in the last bit.  The
// reference's crossing_out (:135-183) deletes t's copy when vertex (t', d') is dead only through
// t''s copy (:150-153 test edges[t'][d'][n'][d] before clearing both), so a copy whose twin was
// never set survives its neighbour's vertex: (t, d) keeps its support towards n iff some d' has
//   edge_exists and A_t and (vertex (t', d') alive or not A_t').
// That condition is monotone in the vertex masks, so in-place passes in any order reach the
// reference's fixed point.  mins[pixel][5]: vertex minimum and pair minima (eps_query_kernel).
// phase 0 writes the vertex masks, phase 1 is one relaxation pass.  A block takes a row segment (the
// O(D) passes' grid, so their dirty flags apply) pixel by pixel; for a pixel, a thread keeps one d'
// (the low `shift` bits of its index) and walks d in steps of kBlock >> shift.
// Dynamic LDS: 8 D doubles + D ints + 5 D bytes.
template <typename T>
__global__ __launch_bounds__(kBlock) void co_exact_kernel(Geometry g, const T *__restrict__ phi,
                                                          const uint8_t *__restrict__ cost, double lambda, double eps,
                                                          int shift, const double *__restrict__ mins, uint8_t *alive,
                                                          int *__restrict__ changed, int phase,
                                                          const uint8_t *dirty_prev, uint8_t *dirty_cur, int seg_pixels,
                                                          int nseg, int *version, int *seen, int segs_per_block,
                                                          int pass_no)
{
    extern __shared__ double eps_lds[];
    __shared__ int holders, pixel_died;
    __shared__ unsigned long long due_lds, due_segments_lds;
    const int D = g.D;
    double *own = eps_lds;            // [4][D]
    double *nbr = own + 4 * D;        // [4][D]: the neighbour's reverse slot
    int *support = reinterpret_cast<int *>(nbr + 4 * D);                 // [D]: bit n = an edge towards n holds
    uint8_t *mine = reinterpret_cast<uint8_t *>(support + D);            // [D]
    uint8_t *theirs = mine + D;                                          // [4][D]
    const int tid = threadIdx.x, lane = tid & 63;
    // grid = (row segments / segs_per_block, band rows): the O(D) passes' segments and their dirty
    // flags (co_block_dirty).  Which of the block's segments are due is looked up by as many lanes
    // at once (co_relax_kernel does the same); a due segment's pixels are taken one after the other.
    const int i = g.row0 + (int)blockIdx.y;
    const int seg_block = (int)blockIdx.x * segs_per_block;
    unsigned long long due_segments;
    if (phase == 1 && dirty_prev) {
        if (tid < 64) {
            const int sg = seg_block + tid;
            const bool mine_due = tid < segs_per_block && sg < nseg &&
                                  co_block_dirty(dirty_prev, dirty_cur, (long long)blockIdx.y * nseg + sg, (int)blockIdx.y, sg, g.Hb, nseg);
            const unsigned long long ballot = __ballot(mine_due);
            if (tid == 0) due_segments_lds = ballot;
        }
        __syncthreads();
        due_segments = due_segments_lds;
    } else {
        const int count = nseg - seg_block < segs_per_block ? nseg - seg_block : segs_per_block;
        due_segments = count >= 64 ? ~0ull : ((1ull << (count > 0 ? count : 0)) - 1ull);
    }
    const int nd = tid & ((1 << shift) - 1), dstep = kBlock >> shift;
    const int group_lanes = shift >= 6 ? 64 : (1 << shift);  // lanes of a wavefront that work on the same d
    const unsigned long long group_mask =
        group_lanes == 64 ? ~0ull : (((1ull << group_lanes) - 1ull) << (lane & ~(group_lanes - 1)));
    const bool group_leader = (lane & (group_lanes - 1)) == 0;
    for (; due_segments; due_segments &= due_segments - 1ull) {
    const int seg = seg_block + __builtin_ctzll(due_segments);
    const long long flag_at = (long long)blockIdx.y * nseg + seg;
    bool block_changed = false;
    __syncthreads();  // the previous segment's shared words are done with
    // Which of the segment's pixels are due: version[pixel] is the last pass in which the pixel lost
    // labels (0: none since the masks were made), seen[pixel] the last pass in which it was looked at.
    // A pixel is due if a neighbour lost labels in that pass or later.  "Or later" would do if a
    // visit were sure to see what its neighbours delete in the same pass, but it is not: another
    // block's stores become visible to this one in no particular order before the kernel ends, its
    // pass number possibly before its mask bytes.  Only what earlier launches wrote is relied on, so a
    // pixel next to a deletion is looked at once more in the pass after it.
    // One lane per pixel looks that up (segments of up to 64 pixels; larger ones look at every pixel).
    const bool tracked = phase == 1 && version && 2 * seg_pixels <= 64;
    unsigned long long due = ~0ull;
    if (tracked) {
        if (tid < 64) {
            const int colour = tid / seg_pixels, q = tid % seg_pixels;
            const int j = 2 * (seg * seg_pixels + q) + ((i + colour) & 1);
            bool mine_due = false;
            if (tid < 2 * seg_pixels && j < g.W) {
                const volatile int *v = version;
                const long long pix = (long long)i * g.W + j;
                int latest = -1;
                if (j > 0) latest = max(latest, v[pix - 1]);
                if (i > 0) latest = max(latest, v[pix - g.W]);
                if (j + 1 < g.W) latest = max(latest, v[pix + 1]);
                if (i + 1 < g.H) latest = max(latest, v[pix + g.W]);
                mine_due = latest >= seen[pix];
            }
            const unsigned long long ballot = __ballot(mine_due);
            if (tid == 0) due_lds = ballot;
        }
        __syncthreads();
        due = due_lds;
        if (due == 0ull) continue;
    }

    auto pixel = [&](int j) {
        const long long pix = (long long)i * g.W + j;
        const int colour = (i + j) & 1, jc = j >> 1, il = i - g.row0 + 1;
        const bool has[4] = {j > 0, i > 0, j + 1 < g.W, i + 1 < g.H};
        const int ni[4] = {i, i - 1, i, i + 1}, nj[4] = {j - 1, j, j + 1, j};
        const long long own_off = (long long)il * g.row_stride + (long long)jc * g.Dpad;
        uint8_t *alive_t = mask_base(g, alive, colour) + ((long long)(i - g.row0) * g.Wh + jc) * g.Dpad;
        __syncthreads();  // the previous pixel's LDS is done with
        for (int e = tid; e < 4 * D; e += kBlock) {
            const int n = e / D, d = e % D;
            double a = 0., b = 0.;
            uint8_t m = 0;
            if (has[n]) {
                a = (double)phi[g.plane_off[colour][n] + own_off + d];
                const long long off = (long long)(ni[n] - g.row0 + 1) * g.row_stride + (long long)(nj[n] >> 1) * g.Dpad;
                b = (double)phi[g.plane_off[1 - colour][n ^ 2] + off + d];
                if (phase == 1)
                    m = *reinterpret_cast<const volatile uint8_t *>(
                            mask_base(g, alive, 1 - colour) + ((long long)(ni[n] - g.row0) * g.Wh + (nj[n] >> 1)) * g.Dpad + d) & 1u;
            }
            own[e] = a;
            nbr[e] = b;
            theirs[e] = m;
        }
        if (tid == 0) {
            holders = 0;
            pixel_died = 0;
        }
        __syncthreads();
        if (phase == 0) {
            const double limit = mins[pix * 5] + eps;
            for (int d = tid; d < g.Dpad; d += kBlock) {
                bool keep = false;
                if (d < D && d <= j) {
                    double sum = 0.;
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        if (has[n]) sum += own[n * D + d];
                    const double v = (double)cost[colour * g.cost_colour_stride + ((long long)(il - 1) * g.Wh + jc) * g.Dpad + d] - sum;
                    keep = v <= limit;
                }
                alive_t[d] = keep ? 1 : 0;
            }
            return;
        }
        for (int d = tid; d < D; d += kBlock) {
            mine[d] = *reinterpret_cast<const volatile uint8_t *>(alive_t + d) & 1u;
            support[d] = 0;
        }
        __syncthreads();
        const int dmax = j < D - 1 ? j : D - 1;
        for (int n = 0; n < 4; ++n) {
            if (!has[n]) continue;  // uniform over the block
            const bool nd_ok = nd < D && nd <= nj[n];
            const double psi = nd < D ? nbr[n * D + nd] : 0.;
            const bool their_vertex = nd < D && theirs[n * D + nd];
            const double limit_t = mins[pix * 5 + 1 + n] + eps;
            const double limit_n = mins[((long long)ni[n] * g.W + nj[n]) * 5 + 1 + (n ^ 2)] + eps;
            for (int d = tid >> shift; d <= dmax; d += dstep) {
                if (!mine[d]) continue;  // uniform over the lanes that share d
                bool ok = nd_ok;
                if (n == 2 && nd > d + 1) ok = false;
                if (n == 0 && d > nd + 1) ok = false;
                if (ok) {
                    const double dist = d > nd ? (double)(d - nd) : (double)(nd - d);
                    const double phi_d = own[n * D + d];
                    const bool a_t = lambda * dist + phi_d + psi <= limit_t;
                    const bool a_n = lambda * dist + psi + phi_d <= limit_n;
                    ok = a_t && (their_vertex || !a_n);
                }
                // one LDS atomic per group of lanes that share d, not one per lane
                const unsigned long long found = __ballot(ok) & group_mask;
                if (found && group_leader) atomicOr(&support[d], 1 << n);
            }
        }
        __syncthreads();
        const int need = (has[0] ? 1 : 0) | (has[1] ? 2 : 0) | (has[2] ? 4 : 0) | (has[3] ? 8 : 0);
        bool died = false, left = false;
        for (int d = tid; d < D; d += kBlock) {
            if (!mine[d]) continue;
            if ((support[d] & need) != need) {
                alive_t[d] = 0;
                died = true;
            } else {
                left = true;
            }
        }
        const bool wave_died = __any(died);
        if (wave_died) block_changed = true;
        // "the pixel has no vertex left" needs all wavefronts: count those that still hold one
        const bool wave_holds = __any(left);
        if (lane == 0 && wave_holds) atomicAdd(&holders, 1);
        if (lane == 0 && wave_died) pixel_died = 1;
        __syncthreads();
        if (tid == 0) {
            if (holders == 0) atomicOr(changed + 1, 1);
            if (tracked) {
                seen[pix] = pass_no;
                if (pixel_died) version[pix] = pass_no;
            }
        }
    };

    for (int colour = 0; colour < 2; ++colour)
        for (int q = 0; q < seg_pixels; ++q) {
            const int slot = colour * seg_pixels + q;
            if (tracked && !((due >> slot) & 1ull)) continue;
            const int j = 2 * (seg * seg_pixels + q) + ((i + colour) & 1);
            if (j < g.W) pixel(j);  // uniform over the block
        }
    if (block_changed && lane == 0) {
        atomicOr(changed, 1);
        if (dirty_cur) dirty_cur[flag_at] = 1;
    }
    }
}

// ---- stage 1: cost volume --------------------------------------------------------------
// cost[d][r][j] = |L[i][j] - R[i][j-d]| for d <= j, else 0; r = band row, i = row0 + r.
// One thread per 4 consecutive bytes of the label-major volume: lanes run along j, so
// a wavefront writes 256 contiguous bytes.
__global__ __launch_bounds__(kBlock) void cost_volume_kernel(const uint8_t *__restrict__ L,
                                                             const uint8_t *__restrict__ Rimg, int W,
                                                             int D, int row0, int Hb,
                                                             uint8_t *__restrict__ out)
{
    const long long total = (long long)D * Hb * W;
    const long long base = 4ll * ((long long)blockIdx.x * kBlock + threadIdx.x);
    if (base >= total) return;
    uint32_t word = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const long long idx = base + b;
        if (idx >= total) break;
        const int j = (int)(idx % W);
        const long long t = idx / W;
        const int r = (int)(t % Hb), d = (int)(t / Hb);
        uint32_t v = 0;
        if (j >= d) {
            const long long rowoff = (long long)(row0 + r) * W;
            const int a = L[rowoff + j], c = Rimg[rowoff + j - d];
            v = (uint32_t)(a > c ? a - c : c - a);
        }
        word |= v << (8 * b);
    }
    if (base + 3 < total) {
        *reinterpret_cast<uint32_t *>(out + base) = word;  // buffer is 4-byte aligned and padded
    } else {
        for (int b = 0; base + b < total; ++b) out[base + b] = (uint8_t)(word >> (8 * b));
    }
}

// Repack label-major cost[d][r][j] into the sweep's layout packed[colour][r][jc][dp]
// (dp fastest, 0 for dp >= D) through an LDS transpose: a block takes 64 consecutive
// columns of one row and all labels.
constexpr int kPackCols = 64;
__global__ __launch_bounds__(kBlock) void pack_cost_kernel(Geometry g, const uint8_t *__restrict__ lm,
                                                           uint8_t *__restrict__ packed)
{
    extern __shared__ uint8_t tile[];  // [kPackCols][Dpad + 4]
    const int pitch = g.Dpad + 4;
    const int r = blockIdx.y, j0 = blockIdx.x * kPackCols;
    const int x = threadIdx.x & 63, dy = threadIdx.x >> 6;
    for (int d = dy; d < g.Dpad; d += kBlock / 64) {
        uint8_t v = 0;
        if (d < g.D && j0 + x < g.W) v = lm[((long long)d * g.Hb + r) * g.W + j0 + x];
        tile[x * pitch + d] = v;
    }
    __syncthreads();
    const int i = g.row0 + r;
    const int words = g.Dpad / 4;
    for (int w = threadIdx.x; w < kPackCols * words; w += kBlock) {
        const int col = w / words, dw = w % words;
        const int j = j0 + col;
        if (j >= g.W) continue;
        const int colour = (i + j) & 1, jc = j >> 1;
        const uint8_t *src = tile + col * pitch + 4 * dw;
        const uint32_t word = src[0] | (src[1] << 8) | (src[2] << 16) | ((uint32_t)src[3] << 24);
        *reinterpret_cast<uint32_t *>(packed + colour * g.cost_colour_stride +
                                      ((long long)r * g.Wh + jc) * g.Dpad + 4 * dw) = word;
    }
}

// ---- post-filter of a disparity map -------------------------------------------------------
// Reference: diffusion_stereo/src/diffusion_graph.rs:399-424 (box_blur), with the wrapping
// usize arithmetic of a release build: the window is rows [i - r, i + r) x columns [j - r,
// j + r) and is EMPTY when i < r or j < r (the range's start wraps past its end); a neighbour
// smaller than the centre gives a wrapped, huge difference and never counts; up starts at the
// centre's value and down at 1; out = up * 255 / (down * norm) in integer arithmetic.
// One thread per pixel, lanes along j (the window's rows are read as coalesced runs).
__global__ __launch_bounds__(kBlock) void box_blur_kernel(const uint32_t *__restrict__ map, int H, int W,
                                                          unsigned long long threshold, unsigned long long radius,
                                                          unsigned long long norm,
                                                          unsigned long long *__restrict__ out)
{
    const long long idx = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (idx >= (long long)H * W) return;
    const int i = (int)(idx / W), j = (int)(idx % W);
    const unsigned long long centre = map[idx];
    unsigned long long up = centre, down = 1;
    if ((unsigned long long)i >= radius && (unsigned long long)j >= radius) {
        const unsigned long long i_hi = (unsigned long long)i + radius < (unsigned long long)H ? i + radius : H;
        const unsigned long long j_hi = (unsigned long long)j + radius < (unsigned long long)W ? j + radius : W;
        for (unsigned long long n_i = i - radius; n_i < i_hi; ++n_i)
            for (unsigned long long n_j = j - radius; n_j < j_hi; ++n_j) {
                const unsigned long long v = map[n_i * W + n_j];
                // (v - centre) as f64 > threshold as f64, v - centre wrapping when v < centre
                const bool keep = !((double)(v - centre) > (double)threshold);
                up += keep ? v : 0ull;
                down += keep ? 1ull : 0ull;
            }
    }
    out[idx] = up * 255ull / (down * norm);
}

// ---- state initialisation / conversion -----------------------------------------------
// phi <- 0 on labels d <= j, +inf elsewhere, for every row of the buffer (halo included).
template <typename T>
__global__ __launch_bounds__(kBlock) void init_phi_kernel(Geometry g, T *__restrict__ phi)
{
    // one thread per 4 consecutive labels (Dpad is a multiple of 4): one 16 / 32 B store
    const int quads = g.Dpad / 4;
    const long long per_plane = (long long)(g.Hb + 2) * g.Wh * quads;
    const long long total = 8 * per_plane;
    const T inf = Real<T>::inf();
    for (long long idx = (long long)blockIdx.x * kBlock + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * kBlock) {
        const int plane = (int)(idx / per_plane);
        const long long rem = idx - plane * per_plane;
        const int colour = plane / 4, slot = plane % 4;
        const long long px = rem / quads;
        const int dp = 4 * (int)(rem - px * quads);
        const int il = (int)(px / g.Wh), jc = (int)(px - (long long)il * g.Wh);
        const int i = g.row0 + il - 1;
        const int j = 2 * jc + ((i + colour) & 1);
        V4<T> v;
#pragma unroll
        for (int k = 0; k < 4; ++k) v.v[k] = (dp + k < g.D && dp + k <= j && j < g.W) ? T(0) : inf;
        store4(phi + g.plane_off[colour][slot] + (long long)il * g.row_stride + (long long)jc * g.Dpad + dp, v);
    }
}

// Deep halo rows (bands.hpp): four consecutive rows of each of four planes to / from one
// contiguous staging buffer per neighbour, so that a neighbour gets one transfer.  blockIdx.y =
// 4 * side + slot; rows[side] = first of the four rows as a row of the stored planes (-1: no
// neighbour on that side); stage + side * 16 rows; pack = 1: planes -> stage, 0: stage -> planes.
struct HaloRows {
    long long plane[4];
    int row[2];
};
template <typename T>
__global__ __launch_bounds__(kBlock) void halo_rows_kernel(T *__restrict__ phi, HaloRows hr, long long row_stride,
                                                           T *__restrict__ stage, int pack)
{
    const int side = blockIdx.y >> 2, slot = blockIdx.y & 3;
    if (hr.row[side] < 0) return;
    const long long n4 = row_stride;  // elements of four rows, in vectors of 4
    T *rows = phi + hr.plane[slot] + (long long)hr.row[side] * row_stride;
    T *buf = stage + ((long long)side * 16 + slot * 4) * row_stride;
    for (long long k = (long long)blockIdx.x * kBlock + threadIdx.x; k < n4; k += (long long)gridDim.x * kBlock) {
        if (pack) store4(buf + 4 * k, load4(rows + 4 * k));
        else store4(rows + 4 * k, load4(buf + 4 * k));
    }
}

// Order-independent checksum: sum over valid (existing slot, d <= j) entries of the
// band's own rows of the entry's bit pattern (as u64, wrapping).
template <typename T>
__global__ __launch_bounds__(kBlock) void checksum_kernel(Geometry g, const T *__restrict__ phi,
                                                          unsigned long long *__restrict__ out)
{
    const long long per_plane = (long long)(g.Hb + 2) * g.Wh * g.Dpad;
    const long long total = 8 * per_plane;
    unsigned long long acc = 0;
    for (long long idx = (long long)blockIdx.x * kBlock + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * kBlock) {
        const long long plane = idx / per_plane, rem = idx % per_plane;
        const int colour = (int)(plane / 4), n = (int)(plane % 4);
        const int dp = (int)(rem % g.Dpad);
        const long long px = rem / g.Dpad;
        const int jc = (int)(px % g.Wh), il = (int)(px / g.Wh);
        if (il < 1 || il > g.Hb) continue;
        const int i = g.row0 + il - 1;
        const int j = 2 * jc + ((i + colour) & 1);
        if (j >= g.W || dp >= g.D || dp > j) continue;
        const bool has = n == 0 ? j > 0 : n == 1 ? i > 0 : n == 2 ? j + 1 < g.W : i + 1 < g.H;
        if (!has) continue;
        T v = phi[g.plane_off[colour][n] + (long long)il * g.row_stride + (long long)jc * g.Dpad + dp];
        if (v == T(0)) v = T(0);  // -0 and +0 count alike
        if constexpr (sizeof(T) == 4) acc += (unsigned long long)__builtin_bit_cast(uint32_t, v);
        else acc += __builtin_bit_cast(unsigned long long, v);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_down(acc, m);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}

}  // namespace ds

#include <type_traits>
#include "fused2.hpp"
