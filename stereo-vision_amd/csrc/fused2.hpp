// Two whole iterations in one kernel, for the part of the image where nothing needs masking.
//
// Same column-pair design as fused_iteration_kernel (a pixel slot owns two columns; vertical and
// own-side messages stay in registers, the message to the neighbouring slot goes through LDS, one
// barrier per row step) with three more stages trailing the first:
//   A  red(r)   of iteration n+1, from phi_black(n) in memory
//   B  black(i) of iteration n+1, on chip only
//   C  red(r)   of iteration n+2, from B's messages
//   D  black(i) of iteration n+2, stored
// Step i runs A(i+1), B(i), C(i-1), D(i-2).  Per (pixel, label) and per TWO iterations the kernel
// reads 4 rows of the second colour and writes 4, i.e. half the one-iteration kernel's traffic;
// it is bound by the vector ALU instead.  Blocks of 12 wavefronts: the kernel wants about 160
// vector registers, which three wavefronts per SIMD leave it; with 16-wavefront blocks (128
// registers) it spills, and every reload also waits for the row loads in flight (measured at
// 4K x 256: 11.3 ms per launch against 9.8).  NP slots yield NP - 3 finished column pairs (A needs all
// NP, B NP - 1, C NP - 2, D NP - 3): a strip of FC2 = 2 (NP - 3) output columns whose first is
// c0 + 2, c0 = col_begin - 2 + blockIdx.x * FC2, reads columns c0 - 2 .. c0 + 2 NP.
//
// Only blocks in which every pixel any stage touches has four neighbours (D = 4 LPP): the host
// launches it over that interior and gives the rest of the image to the one-iteration kernel,
// twice (launch_fused2 in diffstereo.hip).  The arithmetic per pixel is the one-iteration
// kernel's, operation for operation, so the results are the same bits.
//
// The row loop is unrolled by two so that the row's parity (which of a slot's two columns holds
// the red pixel) is a compile-time constant: no selects between "own side" and "across".
#pragma once

namespace ds {

// messages() for f32 with fewer vector-ALU instructions, for a kernel that is bound by them.
// Same values as messages_fused, bit for bit (min is exactly associative and commutative here:
// no NaN reaches it), from a different instruction mix:
//  * the three suffix scans run as prefix scans on the lane-reversed wavefront: one ds_bpermute
//    (LDS crossbar, not the vector ALU) reverses and shifts the scan's input, the same six
//    DPP-min steps as a prefix scan follow, one ds_bpermute brings the result back.  That
//    replaces four DPP steps plus the readlane / select fix-up that adds the rows above.
//  * in-lane, the minimum over the lane's four labels is min(min3(a0, a1, a2), a3), and the four
//    running minima take the scanned value first: six instructions where seven were.
#define DS_DPP6_UP(ctrl, rows)                                                \
    "v_min_f32_dpp %0, %0, %0 " ctrl " row_mask:" rows " bank_mask:0xf\n\t" \
    "v_min_f32_dpp %1, %1, %1 " ctrl " row_mask:" rows " bank_mask:0xf\n\t" \
    "v_min_f32_dpp %2, %2, %2 " ctrl " row_mask:" rows " bank_mask:0xf\n\t" \
    "v_min_f32_dpp %3, %3, %3 " ctrl " row_mask:" rows " bank_mask:0xf\n\t" \
    "v_min_f32_dpp %4, %4, %4 " ctrl " row_mask:" rows " bank_mask:0xf\n\t" \
    "v_min_f32_dpp %5, %5, %5 " ctrl " row_mask:" rows " bank_mask:0xf\n\t"

// six inclusive prefix-min scans over LPP-lane segments, in lock step (see fused_scans6)
template <int LPP> __device__ __forceinline__ void prefix_scans6(float x[6])
{
    if (LPP >= 64) {
        asm("s_nop 1\n\t" DS_DPP6_UP("row_shr:1", "0xf") DS_DPP6_UP("row_shr:2", "0xf") DS_DPP6_UP("row_shr:4", "0xf")
                DS_DPP6_UP("row_shr:8", "0xf") DS_DPP6_UP("row_bcast:15", "0xa") DS_DPP6_UP("row_bcast:31", "0xc")
            : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]));
    } else if (LPP >= 32) {
        asm("s_nop 1\n\t" DS_DPP6_UP("row_shr:1", "0xf") DS_DPP6_UP("row_shr:2", "0xf") DS_DPP6_UP("row_shr:4", "0xf")
                DS_DPP6_UP("row_shr:8", "0xf") DS_DPP6_UP("row_bcast:15", "0xa")
            : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]));
    } else {
        asm("s_nop 1\n\t" DS_DPP6_UP("row_shr:1", "0xf") DS_DPP6_UP("row_shr:2", "0xf") DS_DPP6_UP("row_shr:4", "0xf")
                DS_DPP6_UP("row_shr:8", "0xf")
            : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]));
    }
}

// rev4 = 4 * (63 - lane): the byte address ds_bpermute takes for the mirrored lane
// PARTIAL: the node's labels fill only the lower lanes of its segment (D < 4 LPP, D a multiple of
// 4); lane_on says whether this lane holds labels.  The other lanes carry whatever arithmetic
// leaves in them; all that reaches the lanes below is their contribution to the suffix minima
// and the first of them as a lane's `next`, and both are put to +inf here.  (Prefix minima flow
// upwards only.)
template <int LPP, bool PARTIAL = false>
__device__ __forceinline__ void messages_lean(const V4<float> psi[4], const float lam[4], float lam1, int lseg,
                                              int rev4, V4<float> M[4], bool lane_on = true)
{
    using R = Real<float>;
    const float inf = R::inf();
    const int pslot[3] = {1, 2, 3}, sslot[3] = {0, 1, 3};
    float a[3][4], b[3][4], x[6];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a[t][k] = psi[pslot[t]].v[k] - lam[k];
            b[t][k] = psi[sslot[t]].v[k] + lam[k];
        }
        // the lane's own minimum: what the lanes above (prefix) / below (suffix) get from it
        const float pa = R::min(R::min(R::min(a[t][0], a[t][1]), a[t][2]), a[t][3]);
        float pb = R::min(R::min(R::min(b[t][3], b[t][2]), b[t][1]), b[t][0]);
        if (PARTIAL) pb = lane_on ? pb : inf;
        // prefix: exclusive, so shift one lane up first
        float up = dpp<DPP_WAVE_SHR1>(inf, pa);
        if (LPP < 64) up = (lseg == 0) ? inf : up;
        x[t] = up;
        // suffix: mirrored lane r = 63 - l takes the value of lane l + 1 = 64 - r, i.e. of the
        // mirrored lane r - 1: reversed and shifted in one move.  The segment's last lane has no
        // lane above it.
        const float mirrored = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(rev4 + 4, __builtin_bit_cast(int, pb)));
        x[3 + t] = (lseg == 0) ? inf : mirrored;  // this lane now plays the part of lane 63 - l: lseg 0 <-> the last
    }
    prefix_scans6<LPP>(x);
    float fwd[3][4], bwd[3][4];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const float down = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(rev4, __builtin_bit_cast(int, x[3 + t])));
        float f = x[t], e = down;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            f = R::min(f, a[t][k]);
            fwd[t][k] = f + lam[k];
        }
#pragma unroll
        for (int k = 3; k >= 0; --k) {
            e = R::min(e, b[t][k]);
            bwd[t][k] = e - lam[k];
        }
    }
    float prev = dpp<DPP_WAVE_SHR1>(inf, psi[0].v[3]);
    float next = dpp<DPP_WAVE_SHL1>(inf, PARTIAL ? (lane_on ? psi[2].v[0] : inf) : psi[2].v[0]);
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

template <typename T, int NP, int DL> struct Fused2Lds {
    T ring1[2][NP][DL];  // A: red -> the neighbouring slot's black (iteration n+1)
    T ring2[2][NP][DL];  // B: black -> the neighbouring slot's red (iteration n+2)
    T ring3[2][NP][DL];  // C: red -> the neighbouring slot's black (iteration n+2)
};

// Row segments of unequal length (blockIdx.y -> rows [begin[y], begin[y + 1])), for grids of only a
// few rounds of blocks: n = 0 means equal segments of rows_per_seg rows.
struct RowSegments {
    int n;
    int begin[17];
};

// PARTIAL != 0: D < 4 LPP.  The lanes of a slot beyond its label lanes (Dpad / 4 of them, Dpad = D
// rounded up to 4) load from clamped addresses, store nothing and keep out of the other lanes'
// minima (messages_lean); PARTIAL == 2: D is not a multiple of 4 either, and the padding labels
// (D <= d < Dpad) are put back to +inf after every update; everything else is the D = 4 LPP
// kernel.  Separate instantiations, so that kernel's code does not change (a condition more in
// the source moves its register allocation, even one that folds away).
template <typename T, int LPP, int FW, int PARTIAL = 0>
__global__ __launch_bounds__(64 * FW) void fused2_iteration_kernel(Geometry g, const T *__restrict__ phi,
                                                                   T *__restrict__ phi_out, FusedPlanes np,
                                                                   const uint8_t *__restrict__ cost, T lambda,
                                                                   int col_begin, int row_begin, int row_end,
                                                                   int rows_per_seg, RowSegments segments,
                                                                   int store_first)
{
    using R = Real<T>;
    constexpr int PPW = 64 / LPP, NP = FW * PPW, DL = 4 * LPP, FC2 = 2 * (NP - 3);
    __shared__ Fused2Lds<T, NP, DL> lds;
    const T inf = R::inf();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane / LPP, lseg = lane % LPP;
    // d0: the lane's first label as an address (PARTIAL: clamped into the slot's row); lane_on: the
    // lane holds labels
    const bool lane_on = PARTIAL == 0 || 4 * lseg < g.Dpad;
    const int d0 = PARTIAL != 0 ? (4 * lseg < g.Dpad ? 4 * lseg : g.Dpad - 4) : 4 * lseg;
    const int dl = 4 * lseg;  // the lane's place in an LDS row
    const int kp = wave * PPW + q;
    // the slots of this wavefront, as scalars: a stage skips a wavefront none of whose slots it
    // needs (the strip narrows by one slot per stage), every wavefront still reaches the barrier
    const int wave_first = __builtin_amdgcn_readfirstlane(wave) * PPW, wave_last = wave_first + PPW - 1;
    const int rev4 = 4 * (63 - lane);
    T lam[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) lam[c] = lambda * (T)(dl + c);
    const int c0 = col_begin - 2 + (int)blockIdx.x * FC2;
    const int a = c0 - 1 + 2 * kp;  // first column of the slot's pair (odd)
    int r0 = row_begin + (int)blockIdx.y * rows_per_seg;
    int r1 = r0 + rows_per_seg < row_end ? r0 + rows_per_seg : row_end;
    if (segments.n > 0) {
        r0 = segments.begin[blockIdx.y];
        r1 = segments.begin[blockIdx.y + 1];
    }
    const uint8_t *cost_red = cost, *cost_black = cost + g.cost_colour_stride;
    const int kl = kp > 0 ? kp - 1 : 0, kr = kp < NP - 1 ? kp + 1 : NP - 1;  // neighbouring slots (clamped: the
                                                                              // strip's outermost slots are not kept)
    using V = V4<T>;
    // Strips left of column D: the pixel in column j has the labels d <= j only.  The arithmetic
    // is the same (the labels that do not exist hold +inf in memory and drop out of every
    // minimum); only the results for such labels are put back to +inf (the sums above them may be
    // inf - inf).  One scalar branch per update for the other strips.
    // PARTIAL == 2: the last label lane also holds padding labels, which take the same way (they
    // hold +inf in memory and must hold it again after every update).
    const bool few_labels = [&]() {
        if constexpr (PARTIAL == 2) return true;
        else return c0 - 1 < g.D;
    }();
    auto update = [&](const V psi[4], uint32_t cost_word, int j, V out[4]) {
        V M[4];
        const V qc = unpack_cost<T>(cost_word);
        if constexpr (sizeof(T) == 4) messages_lean<LPP, PARTIAL != 0>(psi, lam, lambda, lseg, rev4, M, lane_on);
        else messages<T, LPP, PARTIAL != 0>(psi, lam, lambda, lseg, lane, M, lane_on);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const T avg = (qc.v[c] + (((M[0].v[c] + M[1].v[c]) + M[2].v[c]) + M[3].v[c])) * T(0.25);
#pragma unroll
            for (int n = 0; n < 4; ++n) out[n].v[c] = avg - M[n].v[c];
        }
        if (few_labels) {
            int last = j;  // the pixel's last label
            if constexpr (PARTIAL == 2) last = j > g.D - 1 ? g.D - 1 : j;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const bool valid = dl + c <= last;
#pragma unroll
                for (int n = 0; n < 4; ++n) out[n].v[c] = valid ? out[n].v[c] : inf;
            }
        }
    };
    auto cost_word = [&](const uint8_t *base, int row, int j) {
        return __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(
            base + ((long long)(row - g.row0) * g.Wh + (j >> 1)) * g.Dpad + d0));
    };

    V pre_psi[4];
    uint32_t pre_cost = 0;
    // loads of the slot's red pixel of row r (P = r & 1: red at column a if odd, a + 1 if even)
    auto fetch = [&](int r, auto parity) {
        constexpr int P = decltype(parity)::value;
        const int j = P ? a : a + 1;
        const int jc = j >> 1, il = r - g.row0 + 1;
        const long long own = (long long)il * g.row_stride + (long long)jc * g.Dpad + d0;
        const long long nb[4] = {own + (long long)(P - 1) * g.Dpad, own - g.row_stride,
                                 own + (long long)P * g.Dpad, own + g.row_stride};
#pragma unroll
        for (int n = 0; n < 4; ++n) pre_psi[n] = ld4_nt(phi + g.plane_off[1][n ^ 2] + nb[n]);
        pre_cost = cost_word(cost_red, r, j);
    };
    // messages kept in registers, named by the stage that made them and the row that uses them
    V a_down_prev = splat(inf), a_down_cur = splat(inf), a_side_cur = splat(inf), a_down_next = splat(inf), a_side_next = splat(inf), a_up_next = splat(inf);
    V b_down_prev = splat(inf), b_down_cur = splat(inf), b_side_cur = splat(inf), b_down_next = splat(inf), b_side_next = splat(inf), b_up_next = splat(inf);
    V c_down_prev = splat(inf), c_down_cur = splat(inf), c_side_cur = splat(inf), c_down_next = splat(inf), c_side_next = splat(inf), c_up_next = splat(inf);

    auto stage_a = [&](auto parity) {  // red of a row with parity P, iteration n+1
        constexpr int P = decltype(parity)::value;
        V psi[4], out[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) psi[n] = pre_psi[n];
        update(psi, pre_cost, P ? a : a + 1, out);
        // odd row: red at a, its right neighbour is the slot's own black pixel, its left one the
        // previous slot's; even row: mirrored
        store4(&lds.ring1[P][kp][dl], P ? out[0] : out[2]);
        a_side_next = P ? out[2] : out[0];
        a_up_next = out[1];
        a_down_next = out[3];
    };
    auto stage_b = [&](int i, auto parity) {  // black(i), iteration n+1: stays on chip
        constexpr int P = decltype(parity)::value;
        const int j = P ? a + 1 : a;
        if (P ? wave_first > NP - 2 : wave_last < 1) return;  // no stage uses this slot's black pixel of this row
        V psi[4], out[4];
        const V across = load4(&lds.ring1[P][P ? kr : kl][dl]);
        psi[0] = P ? a_side_cur : across;
        psi[1] = a_down_prev;
        psi[2] = P ? across : a_side_cur;
        psi[3] = a_up_next;
        update(psi, cost_word(cost_black, i, j), j, out);
        // odd row: black at a + 1, its left neighbour is the slot's own red pixel, its right one the
        // next slot's
        store4(&lds.ring2[P][kp][dl], P ? out[2] : out[0]);
        b_side_next = P ? out[0] : out[2];
        b_up_next = out[1];
        b_down_next = out[3];
    };
    auto stage_c = [&](int r, auto parity) {  // red(r), iteration n+2
        constexpr int P = decltype(parity)::value;
        const int j = P ? a : a + 1;
        if (wave_last < 1 || wave_first > NP - 2) return;  // stage D takes nothing from the outermost slots
        V psi[4], out[4];
        const V across = load4(&lds.ring2[P][P ? kl : kr][dl]);
        psi[0] = P ? across : b_side_cur;
        psi[1] = b_down_prev;
        psi[2] = P ? b_side_cur : across;
        psi[3] = b_up_next;
        update(psi, cost_word(cost_red, r, j), j, out);
        // the last launch of a call also stores the first colour of iteration n+2: the block's own
        // rows, the strip's own columns (odd row: the red pixels of slots 2 .. NP - 2, even row:
        // 1 .. NP - 3); nothing reads those planes while iterations run
        if (store_first && lane_on && r >= r0 && r < r1 && (P ? (kp >= 2 && kp <= NP - 2) : (kp >= 1 && kp <= NP - 3))) {
            const long long own = (long long)(r - g.row0 + 1) * g.row_stride + (long long)(j >> 1) * g.Dpad + d0;
#pragma unroll
            for (int n = 0; n < 4; ++n) st4_nt(phi_out + g.plane_off[0][n] + own, out[n]);
        }
        store4(&lds.ring3[P][kp][dl], P ? out[0] : out[2]);
        c_side_next = P ? out[2] : out[0];
        c_up_next = out[1];
        c_down_next = out[3];
    };
    auto stage_d = [&](int i, auto parity) {  // black(i), iteration n+2: to memory
        constexpr int P = decltype(parity)::value;
        const int j = P ? a + 1 : a;
        const bool mine = P ? (kp >= 1 && kp <= NP - 3) : (kp >= 2 && kp <= NP - 2);
        if (P ? (wave_last < 1 || wave_first > NP - 3) : (wave_last < 2 || wave_first > NP - 2)) return;
        V psi[4], out[4];
        const V across = load4(&lds.ring3[P][P ? kr : kl][dl]);
        psi[0] = P ? c_side_cur : across;
        psi[1] = c_down_prev;
        psi[2] = P ? across : c_side_cur;
        psi[3] = c_up_next;
        update(psi, cost_word(cost_black, i, j), j, out);
        if (mine && lane_on) {
            const long long own = (long long)(i - g.row0 + 1) * g.row_stride + (long long)(j >> 1) * g.Dpad + d0;
#pragma unroll
            for (int n = 0; n < 4; ++n) st4_nt(phi_out + np.out_off[n] + own, out[n]);
        }
    };
    auto rotate_a = [&]() { a_down_prev = a_down_cur; a_down_cur = a_down_next; a_side_cur = a_side_next; };
    auto rotate = [&]() {
        rotate_a();
        b_down_prev = b_down_cur; b_down_cur = b_down_next; b_side_cur = b_side_next;
        c_down_prev = c_down_cur; c_down_cur = c_down_next; c_side_cur = c_side_next;
    };
    // Step i: A(i+1), loads for A(i+2), B(i), C(i-1), D(i-2); P = parity of row i.
    auto step = [&](int i, auto parity) {
        constexpr int P = decltype(parity)::value;
        using Same = std::integral_constant<int, P>;
        using Other = std::integral_constant<int, 1 - P>;
        stage_a(Other{});
        stage_b(i, Same{});
        if (i - 1 >= r0 - 1) stage_c(i - 1, Other{});
        // the next step's loads: in flight through stage D and the barrier (they are not needed
        // earlier, and their sixteen registers are free for stages B and C this way)
        if (i + 2 < r1 + 3) fetch(i + 2, Same{});
        if (i - 2 >= r0) stage_d(i - 2, Same{});
        rotate();
        __syncthreads();
    };
    auto with_parity = [&](int row, auto &&f) {
        if (row & 1) f(std::integral_constant<int, 1>{});
        else f(std::integral_constant<int, 0>{});
    };

    with_parity(r0 - 3, [&](auto p) { fetch(r0 - 3, p); stage_a(p); });
    rotate_a();
    with_parity(r0 - 2, [&](auto p) { fetch(r0 - 2, p); stage_a(p); });
    rotate_a();
    with_parity(r0 - 1, [&](auto p) { fetch(r0 - 1, p); });
    __syncthreads();
    int i = r0 - 2;
    if (i & 1) {  // bring the loop to an even row
        step(i, std::integral_constant<int, 1>{});
        ++i;
    }
    for (; i + 1 < r1 + 2; i += 2) {
        step(i, std::integral_constant<int, 0>{});
        step(i + 1, std::integral_constant<int, 1>{});
    }
    if (i < r1 + 2) step(i, std::integral_constant<int, 0>{});
}

}  // namespace ds
