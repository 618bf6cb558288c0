// Feasibility probe: TWO whole iterations in one kernel.  Same column-pair design as the
// library's fused_iteration_kernel (a wavefront owns two columns; vertical and own-side
// messages stay in registers, the message to the neighbouring wavefront goes through LDS,
// one barrier per row step), with a second stage trailing the first by two rows: the black
// potentials the first iteration would store are consumed on chip by the second iteration's
// red pixels, whose messages feed the second iteration's black pixels -- only those are stored.
// HBM traffic per (pixel pair, label) and per TWO iterations: 16 B read (x 16/13: a strip of 16
// column pairs yields 13) + 16 B written + cost bytes, against 2 x 34 B for two launches of the
// one-iteration kernel.  Interior of the image only (all four neighbours, all labels valid),
// 4K x 256 f32; checked bit for bit against two launches of the library's kernel.
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Xarch_device -fno-honor-nans \
//        -I stereo-vision_amd/csrc -o tools/fused2_probe tools/fused2_probe.hip
// run:   tools/fused2_probe [spacer GiB between phi and the output planes] [rows per segment]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "kernels.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

namespace {

constexpr int FW = 16;            // wavefronts per block = column pairs per strip row
constexpr int OUT_PAIRS = FW - 3; // column pairs whose second-iteration black pixels are complete
constexpr int FC = 2 * OUT_PAIRS; // output columns per strip
constexpr int DL = 256;

struct Lds {
    float ring1[2][FW][DL];  // first iteration, red -> neighbouring pair's black
    float ring2[2][FW][DL];  // first iteration's black -> neighbouring pair's second-iteration red
    float ring3[2][FW][DL];  // second iteration, red -> neighbouring pair's black
};

using V = ds::V4<float>;
typedef float F2 __attribute__((ext_vector_type(2)));

// ds::messages_fused<64> with its additions two labels at a time (v_pk_add_f32): the same
// IEEE additions, fewer instructions.  -DPACKED_ADDS selects it: a measurement of what trimming
// vector instructions is worth once the kernel is no longer bound by HBM.
__device__ __forceinline__ void messages_packed(const V psi[4], const float lam[4], float lam1, int lane, V M[4])
{
    using R = ds::Real<float>;
    const float inf = R::inf();
    const F2 lam_lo = {lam[0], lam[1]}, lam_hi = {lam[2], lam[3]};
    float p[3][4], sfx[3][4];
    const int pslot[3] = {1, 2, 3}, sslot[3] = {0, 1, 3};
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const V &x = psi[pslot[t]];
        const F2 a_lo = F2{x.v[0], x.v[1]} - lam_lo, a_hi = F2{x.v[2], x.v[3]} - lam_hi;
        p[t][0] = a_lo.x;
        p[t][1] = R::min(p[t][0], a_lo.y);
        p[t][2] = R::min(p[t][1], a_hi.x);
        p[t][3] = R::min(p[t][2], a_hi.y);
        const V &y = psi[sslot[t]];
        const F2 b_lo = F2{y.v[0], y.v[1]} + lam_lo, b_hi = F2{y.v[2], y.v[3]} + lam_hi;
        sfx[t][3] = b_hi.y;
        sfx[t][2] = R::min(sfx[t][3], b_hi.x);
        sfx[t][1] = R::min(sfx[t][2], b_lo.y);
        sfx[t][0] = R::min(sfx[t][1], b_lo.x);
    }
    float up[3], down[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        up[t] = ds::dpp<ds::DPP_WAVE_SHR1>(inf, p[t][3]);
        down[t] = ds::dpp<ds::DPP_WAVE_SHL1>(inf, sfx[t][0]);
    }
    ds::fused_scans6<64>(up, down);
    const int row = lane >> 4;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const float t1 = ds::readlane(down[t], 16), t2 = ds::readlane(down[t], 32), t3 = ds::readlane(down[t], 48);
        const float c23 = R::min(t2, t3), c123 = R::min(t1, c23);
        const float add = row == 0 ? c123 : row == 1 ? c23 : row == 2 ? t3 : inf;
        down[t] = R::min(down[t], add);
    }
    float fwd[3][4], bwd[3][4];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const F2 f_lo = F2{R::min(p[t][0], up[t]), R::min(p[t][1], up[t])} + lam_lo;
        const F2 f_hi = F2{R::min(p[t][2], up[t]), R::min(p[t][3], up[t])} + lam_hi;
        const F2 b_lo = F2{R::min(sfx[t][0], down[t]), R::min(sfx[t][1], down[t])} - lam_lo;
        const F2 b_hi = F2{R::min(sfx[t][2], down[t]), R::min(sfx[t][3], down[t])} - lam_hi;
        fwd[t][0] = f_lo.x; fwd[t][1] = f_lo.y; fwd[t][2] = f_hi.x; fwd[t][3] = f_hi.y;
        bwd[t][0] = b_lo.x; bwd[t][1] = b_lo.y; bwd[t][2] = b_hi.x; bwd[t][3] = b_hi.y;
    }
    const float prev = ds::dpp<ds::DPP_WAVE_SHR1>(inf, psi[0].v[3]);
    const float next = ds::dpp<ds::DPP_WAVE_SHL1>(inf, psi[2].v[0]);
    const F2 l1 = {lam1, lam1};
    const F2 left_lo = l1 + F2{prev, psi[0].v[0]}, left_hi = l1 + F2{psi[0].v[1], psi[0].v[2]};
    const F2 right_lo = l1 + F2{psi[2].v[1], psi[2].v[2]}, right_hi = l1 + F2{psi[2].v[3], next};
    const float left[4] = {left_lo.x, left_lo.y, left_hi.x, left_hi.y};
    const float right[4] = {right_lo.x, right_lo.y, right_hi.x, right_hi.y};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        M[0].v[k] = R::min(bwd[0][k], left[k]);
        M[1].v[k] = R::min(fwd[0][k], bwd[1][k]);
        M[2].v[k] = R::min(fwd[1][k], right[k]);
        M[3].v[k] = R::min(fwd[2][k], bwd[2][k]);
    }
}

// A block (strip starting at c0, black rows [r0, r1)) in which every pixel any stage touches has four
// neighbours and all D labels: the mask-free kernel can take it.
__device__ __forceinline__ bool block_is_interior(const ds::Geometry &g, int c0, int r0, int r1)
{
    return g.D == DL && c0 - 1 >= g.D && c0 + 32 < g.W && r0 - 3 >= 1 && r1 + 2 + 1 < g.H;
}

__global__ __launch_bounds__(64 * FW) void fused2_kernel(ds::Geometry g, const float *__restrict__ phi,
                                                         float *__restrict__ phi_new, long long new_plane_stride,
                                                         const uint8_t *__restrict__ cost, float lambda,
                                                         int c_first, int row_first, int rows_per_seg, int row_end,
                                                         bool only_where_safe)
{
    using namespace ds;
    __shared__ Lds lds;
    const float inf = __builtin_huge_valf();
    const int lane = threadIdx.x & 63, kp = threadIdx.x >> 6;
    const int d0 = 4 * lane;
    float lam[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) lam[c] = lambda * (float)(d0 + c);
    const int c0 = c_first + blockIdx.x * FC;   // first output column is c0 + 2
    const int a = c0 - 1 + 2 * kp;              // the pair's first column (odd)
    const int r0 = row_first + blockIdx.y * rows_per_seg;
    const int r1 = r0 + rows_per_seg < row_end ? r0 + rows_per_seg : row_end;
    // Launched over the same grid as fused2_general_kernel: take only the blocks whose every pixel
    // has four neighbours and all labels (the other kernel takes the rest).
    if (only_where_safe && !block_is_interior(g, c0, r0, r1)) return;
    const uint8_t *cost_red = cost, *cost_black = cost + g.cost_colour_stride;

    auto update = [&](const V psi[4], uint32_t cost_word, V out[4]) {
        V M[4];
        const V qc = unpack_cost<float>(cost_word);
#ifdef PACKED_ADDS
        messages_packed(psi, lam, lambda, lane, M);
#else
        messages_fused<64>(psi, lam, lambda, lane, lane, M);
#endif
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float avg = (qc.v[c] + (((M[0].v[c] + M[1].v[c]) + M[2].v[c]) + M[3].v[c])) * 0.25f;
#pragma unroll
            for (int n = 0; n < 4; ++n) out[n].v[c] = avg - M[n].v[c];
        }
    };
    auto cost_word = [&](const uint8_t *base, int row, int j) {
        return __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(
            base + ((long long)(row - g.row0) * g.Wh + (j >> 1)) * g.Dpad + d0));
    };
    auto clampk = [&](int k) { return k < 0 ? 0 : k > FW - 1 ? FW - 1 : k; };

    // ---- first iteration, red pixel of row r: from memory
    V pre_psi[4];
    uint32_t pre_cost = 0;
    auto fetch = [&](int r) {
        const int p = r & 1;
        const int j = p ? a : a + 1;
        const int jc = j >> 1, il = r - g.row0 + 1;
        const long long own = (long long)il * g.row_stride + (long long)jc * g.Dpad + d0;
        const long long nb[4] = {own + (long long)(p - 1) * g.Dpad, own - g.row_stride,
                                 own + (long long)p * g.Dpad, own + g.row_stride};
#pragma unroll
        for (int n = 0; n < 4; ++n) pre_psi[n] = ld4_nt(phi + g.plane_off[1][n ^ 2] + nb[n]);
        pre_cost = cost_word(cost_red, r, j);
    };
    // messages kept in registers, named by the stage that made them (A: red, first iteration;
    // B: black, first iteration; C: red, second iteration) and the row that will use them
    V a_down_prev = splat(inf), a_down_cur = splat(inf), a_side_cur = splat(inf), a_down_next, a_side_next, a_up_next;
    V b_down_prev = splat(inf), b_down_cur = splat(inf), b_side_cur = splat(inf), b_down_next, b_side_next, b_up_next;
    V c_down_prev = splat(inf), c_down_cur = splat(inf), c_side_cur = splat(inf), c_down_next, c_side_next, c_up_next;

    auto stage_a = [&](int r) {  // red(r), first iteration
        const int p = r & 1;
        V psi[4], out[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) psi[n] = pre_psi[n];
        update(psi, pre_cost, out);
        store4(&lds.ring1[p][kp][d0], pick(p != 0, out[0], out[2]));
        a_side_next = pick(p != 0, out[2], out[0]);
        a_up_next = out[1];
        a_down_next = out[3];
    };
    auto stage_b = [&](int i) {  // black(i), first iteration: stays on chip
        const int p = i & 1;
        const int j = p ? a + 1 : a;
        const int nbr = clampk(p ? kp + 1 : kp - 1);
        V psi[4], out[4];
        const V across = load4(&lds.ring1[p][nbr][d0]);
        psi[0] = pick(p != 0, a_side_cur, across);
        psi[1] = a_down_prev;
        psi[2] = pick(p != 0, across, a_side_cur);
        psi[3] = a_up_next;
        update(psi, cost_word(cost_black, i, j), out);
        // odd row: black at a+1, its left neighbour is the pair's own red pixel of the second
        // iteration, its right one the next pair's; even row: mirrored
        store4(&lds.ring2[p][kp][d0], pick(p != 0, out[2], out[0]));
        b_side_next = pick(p != 0, out[0], out[2]);
        b_up_next = out[1];
        b_down_next = out[3];
    };
    auto stage_c = [&](int r) {  // red(r), second iteration
        const int p = r & 1;
        const int j = p ? a : a + 1;
        const int nbr = clampk(p ? kp - 1 : kp + 1);
        V psi[4], out[4];
        const V across = load4(&lds.ring2[p][nbr][d0]);
        psi[0] = pick(p != 0, across, b_side_cur);
        psi[1] = b_down_prev;
        psi[2] = pick(p != 0, b_side_cur, across);
        psi[3] = b_up_next;
        update(psi, cost_word(cost_red, r, j), out);
        store4(&lds.ring3[p][kp][d0], pick(p != 0, out[0], out[2]));
        c_side_next = pick(p != 0, out[2], out[0]);
        c_up_next = out[1];
        c_down_next = out[3];
    };
    auto stage_d = [&](int i) {  // black(i), second iteration: to memory
        const int p = i & 1;
        const int j = p ? a + 1 : a;
        const int nbr = clampk(p ? kp + 1 : kp - 1);
        const bool mine = p ? (kp >= 1 && kp <= OUT_PAIRS) : (kp >= 2 && kp <= OUT_PAIRS + 1);
        V psi[4], out[4];
        const V across = load4(&lds.ring3[p][nbr][d0]);
        psi[0] = pick(p != 0, c_side_cur, across);
        psi[1] = c_down_prev;
        psi[2] = pick(p != 0, across, c_side_cur);
        psi[3] = c_up_next;
        update(psi, cost_word(cost_black, i, j), out);
        if (mine) {
            const long long own = (long long)(i - g.row0 + 1) * g.row_stride + (long long)(j >> 1) * g.Dpad + d0;
#pragma unroll
            for (int n = 0; n < 4; ++n) st4_nt(phi_new + n * new_plane_stride + own, out[n]);
        }
    };

    // Step i: red(i+1) and black(i) of the first iteration, red(i-1) and black(i-2) of the second.
    fetch(r0 - 3);
    stage_a(r0 - 3);
    a_down_prev = a_down_cur; a_down_cur = a_down_next; a_side_cur = a_side_next;
    fetch(r0 - 2);
    stage_a(r0 - 2);
    a_down_prev = a_down_cur; a_down_cur = a_down_next; a_side_cur = a_side_next;
    fetch(r0 - 1);
    __syncthreads();
    for (int i = r0 - 2; i < r1 + 2; ++i) {
        stage_a(i + 1);
        if (i + 2 < r1 + 3) fetch(i + 2);
        stage_b(i);
        if (i - 1 >= r0 - 1) stage_c(i - 1);
        if (i - 2 >= r0) stage_d(i - 2);
        a_down_prev = a_down_cur; a_down_cur = a_down_next; a_side_cur = a_side_next;
        b_down_prev = b_down_cur; b_down_cur = b_down_next; b_side_cur = b_side_next;
        c_down_prev = c_down_cur; c_down_cur = c_down_next; c_side_cur = c_side_next;
        __syncthreads();
    }
}

// ---- the same with the image's borders: every stage has the masked path of the library's kernel
// (pixels without four neighbours, labels d > j, strips that hang over the image's edges), so the
// launch covers the whole image.  Strips start two columns left of their first output column.
__global__ __launch_bounds__(64 * FW) void fused2_general_kernel(ds::Geometry g, const float *__restrict__ phi,
                                                                 float *__restrict__ phi_new, long long new_plane_stride,
                                                                 const uint8_t *__restrict__ cost, float lambda,
                                                                 int row_begin, int row_end, int rows_per_seg,
                                                                 bool skip_where_safe)
{
    using namespace ds;
    __shared__ Lds lds;
    const float inf = __builtin_huge_valf();
    const int lane = threadIdx.x & 63, kp = threadIdx.x >> 6;
    const int d0 = 4 * lane;
    float lam[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) lam[c] = lambda * (float)(d0 + c);
    const int c0 = (int)blockIdx.x * FC - 2;    // first output column is c0 + 2
    const int a = c0 - 1 + 2 * kp;
    const int r0 = row_begin + blockIdx.y * rows_per_seg;
    const int r1 = r0 + rows_per_seg < row_end ? r0 + rows_per_seg : row_end;
    if (skip_where_safe && block_is_interior(g, c0, r0, r1)) return;  // the mask-free kernel's blocks
    const uint8_t *cost_red = cost, *cost_black = cost + g.cost_colour_stride;
    const bool cols_interior = g.D == DL && c0 - 1 >= g.D && c0 + 32 < g.W;
    const V none = splat(inf);
    auto clampk = [&](int k) { return k < 0 ? 0 : k > FW - 1 ? FW - 1 : k; };
    auto rows_interior = [&](int r) { return r >= 1 && r + 1 < g.H; };

    // out[n] of a pixel from its four incoming rows and its cost word; masked form for px
    auto update_fast = [&](const V psi[4], uint32_t cost_word, V out[4]) {
        V M[4];
        const V qc = unpack_cost<float>(cost_word);
        messages_fused<64>(psi, lam, lambda, lane, lane, M);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float avg = (qc.v[c] + (((M[0].v[c] + M[1].v[c]) + M[2].v[c]) + M[3].v[c])) * 0.25f;
#pragma unroll
            for (int n = 0; n < 4; ++n) out[n].v[c] = avg - M[n].v[c];
        }
    };
    auto update_masked = [&](const Pixel &px, const V psi[4], const V &qc, V out[4]) {
        V M[4];
        messages_fused<64>(psi, lam, lambda, lane, lane, M);
        const int K = px.j < g.D - 1 ? px.j : g.D - 1;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float acc = 0.f;
#pragma unroll
            for (int n = 0; n < 4; ++n) acc += px.has[n] ? M[n].v[c] : 0.f;
            const float avg = divide_by_neighbours(qc.v[c] + acc, px.nn);
            const bool valid = px.active && d0 + c <= K;
#pragma unroll
            for (int n = 0; n < 4; ++n) out[n].v[c] = (valid && px.has[n]) ? avg - M[n].v[c] : inf;
        }
    };
    auto red_pixel = [&](int r, bool mine) {
        const int p = r & 1;
        const int j = p ? a : a + 1;
        Pixel px = locate(g, r, p, j >> 1, d0);
        px.active = px.active && mine && j >= 0;
        return px;
    };
    auto black_pixel = [&](int i, bool mine) {
        const int p = i & 1;
        const int j = p ? a + 1 : a;
        Pixel px = locate(g, i, (i + 1) & 1, j >> 1, d0);
        px.active = px.active && mine && j >= 0;
        return px;
    };

    V pre_psi[4];
    uint32_t pre_cost = 0;
    auto fetch = [&](int r) {
        if (r < 0 || r >= g.H) return;
        const int p = r & 1;
        const int j = p ? a : a + 1;
        const int jc = j >> 1, il = r - g.row0 + 1;
        if (cols_interior && rows_interior(r)) {
            const long long own = (long long)il * g.row_stride + (long long)jc * g.Dpad + d0;
            const long long nb[4] = {own + (long long)(p - 1) * g.Dpad, own - g.row_stride,
                                     own + (long long)p * g.Dpad, own + g.row_stride};
#pragma unroll
            for (int n = 0; n < 4; ++n) pre_psi[n] = ld4_nt(phi + g.plane_off[1][n ^ 2] + nb[n]);
            pre_cost = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(
                cost_red + ((long long)(il - 1) * g.Wh + jc) * g.Dpad + d0));
        } else {
            const Pixel px = red_pixel(r, true);
            load_psi(g, phi, 1, px, pre_psi);
            pre_cost = (px.active && px.d0 <= px.j) ? *reinterpret_cast<const uint32_t *>(cost_red + px.cost) : 0u;
        }
    };
    V a_down_prev = none, a_down_cur = none, a_side_cur = none, a_down_next, a_side_next, a_up_next;
    V b_down_prev = none, b_down_cur = none, b_side_cur = none, b_down_next, b_side_next, b_up_next;
    V c_down_prev = none, c_down_cur = none, c_side_cur = none, c_down_next, c_side_next, c_up_next;

    auto stage_a = [&](int r) {  // red(r), first iteration, from memory
        a_down_next = a_side_next = a_up_next = none;
        if (r < 0 || r >= g.H) return;
        const int p = r & 1;
        V psi[4], out[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) psi[n] = pre_psi[n];
        if (cols_interior && rows_interior(r)) update_fast(psi, pre_cost, out);
        else update_masked(red_pixel(r, true), psi, unpack_cost<float>(pre_cost), out);
        store4(&lds.ring1[p][kp][d0], pick(p != 0, out[0], out[2]));
        a_side_next = pick(p != 0, out[2], out[0]);
        a_up_next = out[1];
        a_down_next = out[3];
    };
    // a black pixel's four incoming rows: own pair's red pixel of the same row, the neighbouring
    // pair's (through LDS), own pair's red pixels of the rows above and below
    auto black_inputs = [&](int i, const Pixel &px, bool fast, const float (*ring)[FW][DL], const V &side_cur,
                            const V &down_prev, const V &up_next, V psi[4]) {
        const int p = i & 1;
        const int nbr = clampk(p ? kp + 1 : kp - 1);
        if (fast) {
            const V across = load4(&ring[p][nbr][d0]);
            psi[0] = pick(p != 0, side_cur, across);
            psi[1] = down_prev;
            psi[2] = pick(p != 0, across, side_cur);
            psi[3] = up_next;
        } else {
            const bool has_across = px.active && (p ? px.has[2] : px.has[0]);
            const bool has_own = px.active && (p ? px.has[0] : px.has[2]);
            V across = none;
            if (has_across) across = load4(&ring[p][nbr][d0]);
            const V own_side = pick(has_own, side_cur, none);
            psi[0] = pick(p != 0, own_side, across);
            psi[1] = pick(px.active && px.has[1], down_prev, none);
            psi[2] = pick(p != 0, across, own_side);
            psi[3] = pick(px.active && px.has[3], up_next, none);
        }
    };
    auto stage_b = [&](int i) {  // black(i), first iteration: stays on chip
        b_down_next = b_side_next = b_up_next = none;
        if (i < 0 || i >= g.H) return;
        const int p = i & 1;
        const bool mine = p ? kp <= FW - 2 : kp >= 1;
        const bool fast = cols_interior && rows_interior(i);
        const Pixel px = black_pixel(i, mine);
        V psi[4], out[4];
        black_inputs(i, px, fast, lds.ring1, a_side_cur, a_down_prev, a_up_next, psi);
        if (fast) {
            const int j = p ? a + 1 : a;
            update_fast(psi, __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(
                                 cost_black + ((long long)(i - g.row0) * g.Wh + (j >> 1)) * g.Dpad + d0)), out);
        } else {
            update_masked(px, psi, load_cost<float>(cost_black, px), out);
        }
        store4(&lds.ring2[p][kp][d0], pick(p != 0, out[2], out[0]));
        b_side_next = pick(p != 0, out[0], out[2]);
        b_up_next = out[1];
        b_down_next = out[3];
    };
    auto stage_c = [&](int r) {  // red(r), second iteration, from the first iteration's black pixels on chip
        c_down_next = c_side_next = c_up_next = none;
        if (r < 0 || r >= g.H) return;
        const int p = r & 1;
        const bool fast = cols_interior && rows_interior(r);
        const Pixel px = red_pixel(r, kp >= 1 && kp <= FW - 2);
        const int nbr = clampk(p ? kp - 1 : kp + 1);
        V psi[4], out[4];
        if (fast) {
            const V across = load4(&lds.ring2[p][nbr][d0]);
            psi[0] = pick(p != 0, across, b_side_cur);
            psi[1] = b_down_prev;
            psi[2] = pick(p != 0, b_side_cur, across);
            psi[3] = b_up_next;
            const int j = p ? a : a + 1;
            update_fast(psi, __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(
                                 cost_red + ((long long)(r - g.row0) * g.Wh + (j >> 1)) * g.Dpad + d0)), out);
        } else {
            // odd row: red at a, left neighbour = previous pair's black, right = own pair's; even: mirrored
            const bool has_across = px.active && (p ? px.has[0] : px.has[2]);
            const bool has_own = px.active && (p ? px.has[2] : px.has[0]);
            V across = none;
            if (has_across) across = load4(&lds.ring2[p][nbr][d0]);
            const V own_side = pick(has_own, b_side_cur, none);
            psi[0] = pick(p != 0, across, own_side);
            psi[1] = pick(px.active && px.has[1], b_down_prev, none);
            psi[2] = pick(p != 0, own_side, across);
            psi[3] = pick(px.active && px.has[3], b_up_next, none);
            update_masked(px, psi, load_cost<float>(cost_red, px), out);
        }
        store4(&lds.ring3[p][kp][d0], pick(p != 0, out[0], out[2]));
        c_side_next = pick(p != 0, out[2], out[0]);
        c_up_next = out[1];
        c_down_next = out[3];
    };
    auto stage_d = [&](int i) {  // black(i), second iteration: to memory
        const int p = i & 1;
        const bool mine = p ? (kp >= 1 && kp <= OUT_PAIRS) : (kp >= 2 && kp <= OUT_PAIRS + 1);
        const bool fast = cols_interior && rows_interior(i);
        const Pixel px = black_pixel(i, mine);
        V psi[4], out[4];
        black_inputs(i, px, fast, lds.ring3, c_side_cur, c_down_prev, c_up_next, psi);
        if (fast) {
            const int j = p ? a + 1 : a;
            update_fast(psi, __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(
                                 cost_black + ((long long)(i - g.row0) * g.Wh + (j >> 1)) * g.Dpad + d0)), out);
            if (mine) {
                const long long own = (long long)(i - g.row0 + 1) * g.row_stride + (long long)(j >> 1) * g.Dpad + d0;
#pragma unroll
                for (int n = 0; n < 4; ++n) st4_nt(phi_new + n * new_plane_stride + own, out[n]);
            }
        } else {
            update_masked(px, psi, load_cost<float>(cost_black, px), out);
            const int K = px.j < g.D - 1 ? px.j : g.D - 1;
#pragma unroll
            for (int n = 0; n < 4; ++n)
                if (px.active && px.has[n] && d0 <= K) store4(phi_new + n * new_plane_stride + px.own, out[n]);
        }
    };
    auto rotate = [&]() {
        a_down_prev = a_down_cur; a_down_cur = a_down_next; a_side_cur = a_side_next;
        b_down_prev = b_down_cur; b_down_cur = b_down_next; b_side_cur = b_side_next;
        c_down_prev = c_down_cur; c_down_cur = c_down_next; c_side_cur = c_side_next;
    };

    fetch(r0 - 3);
    stage_a(r0 - 3);
    rotate();
    fetch(r0 - 2);
    stage_a(r0 - 2);
    rotate();
    fetch(r0 - 1);
    __syncthreads();
    for (int i = r0 - 2; i < r1 + 2; ++i) {
        stage_a(i + 1);
        if (i + 2 < r1 + 3) fetch(i + 2);
        stage_b(i);
        if (i - 1 >= r0 - 1) stage_c(i - 1);
        if (i - 2 >= r0) stage_d(i - 2);
        rotate();
        __syncthreads();
    }
}

// finite entries (the labels the reference uses) get random values; +inf entries stay
__global__ void perturb_kernel(float *p, long long n, unsigned seed)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)(i * 2654435761u) ^ seed;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        if (p[i] < __builtin_huge_valf()) p[i] = (float)(x & 0xffff) * (1.0f / 1024.0f) - 32.0f;
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

// mismatching dwords between two sets of four colour-1 planes over rows [ra, rb), columns [ca, cb)
__global__ void compare_kernel(ds::Geometry g, const float *x, const float *y, long long plane_stride, int ra, int rb,
                               int ca, int cb, unsigned long long *bad)
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
        mine += __float_as_uint(x[n * plane_stride + off]) != __float_as_uint(y[n * plane_stride + off]);
    }
    if (mine) atomicAdd(bad, mine);
}

}  // namespace

int main(int argc, char **argv)
{
    const int spacer_gib = argc > 1 ? std::atoi(argv[1]) : 0;
    const int rows_per_seg = argc > 2 ? std::atoi(argv[2]) : 64;
    ds::Geometry g{};
    g.H = 2160; g.W = 3840; g.D = DL; g.Dpad = DL; g.Wh = g.W / 2; g.row0 = 0; g.Hb = g.H;
    g.row_stride = (long long)g.Wh * g.Dpad;
    const long long plane = (long long)(g.Hb + 2) * g.row_stride;
    for (int c = 0; c < 2; ++c)
        for (int s = 0; s < 4; ++s) g.plane_off[c][s] = (long long)(c * 4 + s) * plane;
    g.cost_colour_stride = (long long)g.Hb * g.Wh * g.Dpad;
    float *phi, *out2, *ref1, *ref2;
    uint8_t *cost;
    void *spacer = nullptr;
    unsigned long long *bad;
    CK(hipMalloc(&phi, 8 * plane * sizeof(float)));
    if (spacer_gib > 0) CK(hipMalloc(&spacer, (size_t)spacer_gib << 30));
    CK(hipMalloc(&out2, 4 * plane * sizeof(float)));   // result of the two-iteration kernel
    CK(hipMalloc(&ref1, 4 * plane * sizeof(float)));   // library kernel, after one iteration
    CK(hipMalloc(&ref2, 4 * plane * sizeof(float)));   // ... after two
    CK(hipMalloc(&cost, 2 * g.cost_colour_stride));
    CK(hipMalloc(&bad, 8));
    hipLaunchKernelGGL(fill_kernel, dim3(8192), dim3(256), 0, 0, phi, 8 * plane, 12345u);
    hipLaunchKernelGGL(fill_bytes_kernel, dim3(8192), dim3(256), 0, 0, (uint32_t *)cost, 2 * g.cost_colour_stride / 4);
    CK(hipMemset(out2, 0, 4 * plane * sizeof(float)));
    CK(hipMemset(ref1, 0, 4 * plane * sizeof(float)));
    CK(hipMemset(ref2, 0, 4 * plane * sizeof(float)));
    CK(hipDeviceSynchronize());

    const int c_first = 270, row_first = 6, row_end = g.H - 6;
    const int strips = (g.W - 36 - c_first) / FC;   // every read of the last strip stays inside the image
    const int segs = (row_end - row_first + rows_per_seg - 1) / rows_per_seg;
    const int ca = c_first + 2, cb = c_first + 2 + strips * FC;
    std::printf("two-iteration kernel: strips %d x segments %d = %d blocks of %d threads, LDS %zu bytes\n", strips, segs,
                strips * segs, 64 * FW, sizeof(Lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const float lambda = 1.2f;
    float ms2 = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        float ms = 0;
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(fused2_kernel, dim3(strips, segs), dim3(64 * FW), 0, 0, g, phi, out2, plane, cost, lambda,
                           c_first, row_first, rows_per_seg, row_end, false);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::printf("two iterations in one launch over the interior: %.3f ms\n", ms);
        ms2 = ms < ms2 ? ms : ms2;
    }
    // the library's one-iteration kernel twice over the whole image: phi -> ref1 -> ref2
    float ms1 = 1e9f;
    {
        const int strips_all = (g.W + 29) / 30;
        const int nseg = 34, rps = (g.H + nseg - 1) / nseg;
        ds::FusedPlanes np1, np2;
        ds::Geometry g2 = g;
        for (int n = 0; n < 4; ++n) {
            np1.out_off[n] = (ref1 - phi) + n * plane;
            np2.out_off[n] = (ref2 - phi) + n * plane;
            g2.plane_off[1][n] = (ref1 - phi) + n * plane;
        }
        for (int rep = 0; rep < 3; ++rep) {
            float ms = 0;
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL((ds::fused_iteration_kernel<float, 64, 16, false>), dim3(strips_all, (g.H + rps - 1) / rps),
                               dim3(64 * 16), 0, 0, g, phi, np1, cost, lambda, 0, g.H, rps, 0, 0);
            hipLaunchKernelGGL((ds::fused_iteration_kernel<float, 64, 16, false>), dim3(strips_all, (g.H + rps - 1) / rps),
                               dim3(64 * 16), 0, 0, g2, phi, np2, cost, lambda, 0, g.H, rps, 0, 0);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            CK(hipEventElapsedTime(&ms, e0, e1));
            ms1 = ms < ms1 ? ms : ms1;
        }
    }
    CK(hipMemset(bad, 0, 8));
    hipLaunchKernelGGL(compare_kernel, dim3(4096), dim3(256), 0, 0, g, out2, ref2, plane, row_first, row_end, ca, cb, bad);
    unsigned long long h_bad = 0;
    CK(hipMemcpy(&h_bad, bad, 8, hipMemcpyDeviceToHost));
    const double frac = (double)(row_end - row_first) * (cb - ca) / ((double)g.H * g.W);
    std::printf("library kernel, two launches over the whole image: %.3f ms (%.3f ms for the interior's share, %.1f %% of the image)\n",
                ms1, ms1 * frac, 100 * frac);
    std::printf("two-in-one / two launches on the same pixels: %.3f   mismatching dwords in the interior: %llu\n",
                ms2 / (ms1 * frac), h_bad);

    // ---- second part: the kernel with the borders' masked paths, over the whole image
    {
        auto init_like_phi = [&](float *buf) {  // 0 on the reference's labels, +inf elsewhere, as ds_create leaves phi
            ds::Geometry t = g;
            for (int n = 0; n < 4; ++n) t.plane_off[1][n] = (buf - phi) + n * plane;  // colour-0 planes: phi's own, unused here
            hipLaunchKernelGGL(ds::init_phi_kernel<float>, dim3(8192), dim3(ds::kBlock), 0, 0, t, phi);
        };
        hipLaunchKernelGGL(ds::init_phi_kernel<float>, dim3(8192), dim3(ds::kBlock), 0, 0, g, phi);
        hipLaunchKernelGGL(perturb_kernel, dim3(8192), dim3(256), 0, 0, phi, 8 * plane, 777u);
        init_like_phi(out2);
        init_like_phi(ref1);
        init_like_phi(ref2);
        CK(hipDeviceSynchronize());
        const int strips_g = (g.W + FC - 1) / FC + 1, nseg_g = (g.H + rows_per_seg - 1) / rows_per_seg;
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            float ms = 0;
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(fused2_general_kernel, dim3(strips_g, nseg_g), dim3(64 * FW), 0, 0, g, phi, out2, plane, cost,
                               lambda, 0, g.H, rows_per_seg, false);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        const int strips_all = (g.W + 29) / 30;
        const int nseg = 34, rps = (g.H + nseg - 1) / nseg;
        ds::FusedPlanes np1, np2;
        ds::Geometry g2 = g;
        for (int n = 0; n < 4; ++n) {
            np1.out_off[n] = (ref1 - phi) + n * plane;
            np2.out_off[n] = (ref2 - phi) + n * plane;
            g2.plane_off[1][n] = (ref1 - phi) + n * plane;
        }
        float lib = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            float ms = 0;
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL((ds::fused_iteration_kernel<float, 64, 16, false>), dim3(strips_all, (g.H + rps - 1) / rps),
                               dim3(64 * 16), 0, 0, g, phi, np1, cost, lambda, 0, g.H, rps, 0, 0);
            hipLaunchKernelGGL((ds::fused_iteration_kernel<float, 64, 16, false>), dim3(strips_all, (g.H + rps - 1) / rps),
                               dim3(64 * 16), 0, 0, g2, phi, np2, cost, lambda, 0, g.H, rps, 0, 0);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            CK(hipEventElapsedTime(&ms, e0, e1));
            lib = ms < lib ? ms : lib;
        }
        CK(hipMemset(bad, 0, 8));
        hipLaunchKernelGGL(compare_kernel, dim3(4096), dim3(256), 0, 0, g, out2, ref2, plane, 0, g.H, 0, g.W, bad);
        unsigned long long whole_bad = 0;
        CK(hipMemcpy(&whole_bad, bad, 8, hipMemcpyDeviceToHost));
        std::printf("with borders, whole image: two iterations in one launch %.3f ms, two launches of the library kernel %.3f ms "
                    "(x%.3f); mismatching dwords: %llu\n", best, lib, best / lib, whole_bad);
        h_bad += whole_bad;

        // third part: the mask-free kernel on the blocks it can take, the masked one on the rest
        init_like_phi(out2);
        CK(hipDeviceSynchronize());
        float best_split = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            float ms = 0;
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(fused2_kernel, dim3(strips_g, nseg_g), dim3(64 * FW), 0, 0, g, phi, out2, plane, cost, lambda,
                               -2, 0, rows_per_seg, g.H, true);
            hipLaunchKernelGGL(fused2_general_kernel, dim3(strips_g, nseg_g), dim3(64 * FW), 0, 0, g, phi, out2, plane, cost,
                               lambda, 0, g.H, rows_per_seg, true);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            CK(hipEventElapsedTime(&ms, e0, e1));
            best_split = ms < best_split ? ms : best_split;
        }
        // the same two launches side by side on two streams (the masked one first: its blocks are the slow ones)
        {
            hipStream_t s1, s2;
            CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
            CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
            hipEvent_t done2;
            CK(hipEventCreate(&done2));
            CK(hipDeviceSynchronize());
            float best_par = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                float ms = 0;
                CK(hipEventRecord(e0, s1));
                CK(hipStreamWaitEvent(s2, e0, 0));
                hipLaunchKernelGGL(fused2_general_kernel, dim3(strips_g, nseg_g), dim3(64 * FW), 0, s2, g, phi, out2, plane,
                                   cost, lambda, 0, g.H, rows_per_seg, true);
                hipLaunchKernelGGL(fused2_kernel, dim3(strips_g, nseg_g), dim3(64 * FW), 0, s1, g, phi, out2, plane, cost,
                                   lambda, -2, 0, rows_per_seg, g.H, true);
                CK(hipEventRecord(done2, s2));
                CK(hipStreamWaitEvent(s1, done2, 0));
                CK(hipEventRecord(e1, s1));
                CK(hipEventSynchronize(e1));
                CK(hipGetLastError());
                CK(hipEventElapsedTime(&ms, e0, e1));
                best_par = ms < best_par ? ms : best_par;
            }
            std::printf("with borders, the two kernels side by side on two streams: %.3f ms (x%.3f)\n", best_par, best_par / lib);
            CK(hipStreamDestroy(s1));
            CK(hipStreamDestroy(s2));
        }
        CK(hipMemset(bad, 0, 8));
        hipLaunchKernelGGL(compare_kernel, dim3(4096), dim3(256), 0, 0, g, out2, ref2, plane, 0, g.H, 0, g.W, bad);
        unsigned long long split_bad = 0;
        CK(hipMemcpy(&split_bad, bad, 8, hipMemcpyDeviceToHost));
        std::printf("with borders, two kernels (mask-free blocks + the rest): %.3f ms (x%.3f of the library's two launches); "
                    "mismatching dwords: %llu\n", best_split, best_split / lib, split_bad);
        h_bad += split_bad;
    }
    return h_bad ? 1 : 0;
}
