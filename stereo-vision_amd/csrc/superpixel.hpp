// Device code of the superpixel variant of the diffusion (the reference's crate
// superpixel_stereo) for gfx950.
//
// The graph: the image is cut into windows of super_height x super_width pixels; each window
// holds two nodes, its pixels at or below the window's mean intensity (superpixel 0) and the
// others (superpixel 1) (superpixels.rs:76-108).  A node has up to 9 neighbour slots
// (utils.rs:29-35): 0 the other node of its window, 1-2 / 3-4 / 5-6 / 7-8 the two nodes of the
// left / upper / right / lower window.  The pairwise term of an edge is weighted by the length
// of the boundary the two nodes share ("perimeter", diffusion_graph.rs:164-173); the ordering
// constraint between horizontal neighbours is switched off (:199-205), so every slot is
// unconstrained.  Labels d <= left_j, the leftmost column of the node's pixels (:192-198).
//
// Update rule per node t (diffusion_graph.rs:289-343,425-482), slot n with neighbour t' and
// reverse slot n', w_n = perimeter * smoothing:
//   psi_n  = phi[t'][n'][.]
//   M_n(d) = min_{d' <= K'} ( w_n |d - d'| + psi_n(d') )
//   phi[t][n][d] <- ( q_t(d) + sum_m M_m(d) ) / |N(t)| - M_n(d)          for d <= K
// evaluated with the same prefix / suffix minima over lanes as the pixel-grid kernels
// (kernels.hpp: 4 labels per lane, LPP lanes per node).
//
// Schedules.  Nodes of equal colour = 2 * ((super_i + super_j) & 1) + superpixel are never
// neighbours: four launches make an iteration.  The reference's own order -- windows in raster
// order, superpixel 0 then 1 (:357-415) -- is reproduced exactly by launching anti-diagonals of
// windows, super_i + super_j ascending, superpixel 0 then 1: a node then sees its left and upper
// windows after, its right and lower windows before, their update, and the other node of its
// window as the raster order has it.
//
// Layout: phi[colour][slot][node in colour][Dpad], node in colour = super_i * NJh + super_j / 2;
// labels d > K and the padding hold +inf.  cost[colour][node][Dpad] (T; sums of up to
// super_height * super_width byte differences: exact), K[colour][node], weights
// w[colour][node][9] (T).
#pragma once
#include "lanes.hpp"

namespace ds {

struct SpGeometry {
    int H, W, D, Dpad;
    int sh, sw;        // window
    int NI, NJ, NJh;   // windows down / across, (NJ + 1) / 2
    long long colour_nodes;                 // NI * NJh
    long long plane_elems;                  // colour_nodes * Dpad
};

__host__ __device__ __forceinline__ long long sp_plane(const SpGeometry &g, int colour, int slot)
{
    return (long long)(colour * 9 + slot) * g.plane_elems;
}

// neighbour of node (si, sj, s) through slot n: window, superpixel, reverse slot
__device__ __forceinline__ bool sp_neighbour(const SpGeometry &g, int si, int sj, int s, int n, int *ni, int *nj,
                                             int *ns, int *nrev)
{
    switch (n) {
    case 0: *ni = si; *nj = sj; *ns = 1 - s; *nrev = 0; return true;
    case 1: case 2: *ni = si; *nj = sj - 1; *ns = (n & 1) ? 0 : 1; *nrev = s == 0 ? 5 : 6; return sj > 0;
    case 3: case 4: *ni = si - 1; *nj = sj; *ns = (n & 1) ? 0 : 1; *nrev = s == 0 ? 7 : 8; return si > 0;
    case 5: case 6: *ni = si; *nj = sj + 1; *ns = (n & 1) ? 0 : 1; *nrev = s == 0 ? 1 : 2; return sj + 1 < g.NJ;
    default: *ni = si + 1; *nj = sj; *ns = (n & 1) ? 0 : 1; *nrev = s == 0 ? 3 : 4; return si + 1 < g.NI;
    }
}

__device__ __forceinline__ int sp_colour(int si, int sj, int s) { return 2 * ((si + sj) & 1) + s; }
__device__ __forceinline__ long long sp_index(const SpGeometry &g, int si, int sj) { return (long long)si * g.NJh + (sj >> 1); }

// ---- set-up: split, left_j, cost -----------------------------------------------------------
// One block per window.  superpixels[i][j] = L[i][j] <= round(mean of the window in f32) ? 0 : 1
// (superpixels.rs:92-103, utils.rs:230-235); K = min(left_j, D - 1), left_j = leftmost column of
// the node's pixels, the window's last column for a node without pixels (superpixels.rs:110-123);
// cost(d) = sum over the node's pixels of |L[i][j] - R[i][j - d]| (diffusion_graph.rs:137-148).
template <typename T>
__global__ __launch_bounds__(kBlock) void sp_setup_kernel(SpGeometry g, const uint8_t *__restrict__ left,
                                                          const uint8_t *__restrict__ right,
                                                          uint8_t *__restrict__ superpixels, int *__restrict__ Kout,
                                                          T *__restrict__ cost)
{
    const int si = blockIdx.y, sj = blockIdx.x;
    const int npx = g.sh * g.sw;
    __shared__ unsigned int ssum;
    __shared__ int left_j[2];
    if (threadIdx.x == 0) {
        ssum = 0;
        left_j[0] = left_j[1] = sj * g.sw + g.sw - 1;
    }
    __syncthreads();
    unsigned int part = 0;
    for (int k = threadIdx.x; k < npx; k += kBlock)
        part += left[(long long)(si * g.sh + k / g.sw) * g.W + sj * g.sw + k % g.sw];
    atomicAdd(&ssum, part);
    __syncthreads();
    // (sum as f32 / len as f32).round() as u32: round half away from zero
    const unsigned int avg = (unsigned int)__builtin_roundf((float)ssum / (float)npx);
    for (int k = threadIdx.x; k < npx; k += kBlock) {
        const int i = si * g.sh + k / g.sw, j = sj * g.sw + k % g.sw;
        const int s = left[(long long)i * g.W + j] <= avg ? 0 : 1;
        superpixels[(long long)i * g.W + j] = (uint8_t)s;
        atomicMin(&left_j[s], j);
    }
    __syncthreads();
    for (int s = 0; s < 2; ++s) {
        const int colour = 2 * ((si + sj) & 1) + s;
        const long long node = (long long)colour * g.colour_nodes + (long long)si * g.NJh + (sj >> 1);
        const int K = left_j[s] < g.D - 1 ? left_j[s] : g.D - 1;
        if (threadIdx.x == 0) Kout[node] = K;
        for (int d = threadIdx.x; d < g.Dpad; d += kBlock) {
            T q = T(0);
            if (d <= K)
                for (int k = 0; k < npx; ++k) {
                    const int i = si * g.sh + k / g.sw, j = sj * g.sw + k % g.sw;
                    if (superpixels[(long long)i * g.W + j] == s) {
                        const int a = left[(long long)i * g.W + j], b = right[(long long)i * g.W + j - d];
                        q += (T)(a > b ? a - b : b - a);
                    }
                }
            cost[node * g.Dpad + d] = q;
        }
    }
}

// Boundary length inside each window, one thread per window, the additions in the order of
// superpixels.rs:153-179 (the result depends on it in the last bits): perim[si][sj] for slot 0 of
// both nodes.  The loop looks at the window's rows and columns but the last; "differs" compares
// node ids, so the pixel to the lower left of the window's first column -- a pixel of the window
// to the left -- always differs.  Weights between windows: see sp_create (the reference's loops
// leave them 0).
__global__ void sp_inner_perimeter_kernel(SpGeometry g, const uint8_t *__restrict__ superpixels, double u, double v,
                                          double *__restrict__ perim)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= g.NI * g.NJ) return;
    const int si = w / g.NJ, sj = w % g.NJ;
    const int iend = si * g.sh + g.sh, jend = sj * g.sw + g.sw;
    auto sp = [&](int i, int j) { return (int)superpixels[(long long)i * g.W + j]; };
    double p = 0.;
    for (int i = si * g.sh; i < iend - 1; ++i)
        for (int j = sj * g.sw; j < jend - 1; ++j) {
            const int here = sp(i, j);
            if (here != sp(i, j + 1) || here != sp(i + 1, j)) p += u;
            if (here != sp(i + 1, j + 1)) p += v;
            if (j > 0 && (j == sj * g.sw || here != sp(i + 1, j - 1))) p += v;
        }
    perim[w] = p;
}

// ---- one node's update -------------------------------------------------------------------------
template <typename T> struct SpNode {
    bool active;
    int si, sj, s, colour, K, nn;
    long long idx;  // node in colour
};

// The nodes a launch updates: mode 0 = all nodes of `colour`; mode 1 = the nodes with superpixel
// `colour & 1` on the anti-diagonal of windows super_i + super_j = diag.
template <typename T>
__device__ __forceinline__ SpNode<T> sp_pick(const SpGeometry &g, long long slot, int mode, int colour, int diag)
{
    SpNode<T> t;
    t.s = colour & 1;
    if (mode == 0) {
        t.si = (int)(slot / g.NJh);
        const int jh = (int)(slot % g.NJh);
        t.sj = 2 * jh + ((t.si + (colour >> 1)) & 1);
        t.active = t.si < g.NI && t.sj < g.NJ;
    } else {
        const int first = diag - (g.NJ - 1) > 0 ? diag - (g.NJ - 1) : 0;
        t.si = first + (int)slot;
        t.sj = diag - t.si;
        t.active = t.si < g.NI && t.sj >= 0 && t.sj < g.NJ;
    }
    if (!t.active) t.si = t.sj = 0;
    t.colour = sp_colour(t.si, t.sj, t.s);
    t.idx = sp_index(g, t.si, t.sj);
    t.nn = 1 + (t.sj > 0 ? 2 : 0) + (t.si > 0 ? 2 : 0) + (t.sj + 1 < g.NJ ? 2 : 0) + (t.si + 1 < g.NI ? 2 : 0);
    return t;
}

// M_n for the lane's four labels; false (and M = +inf) where the slot has no neighbour
template <typename T, int LPP>
__device__ __forceinline__ bool sp_message(const SpGeometry &g, const T *__restrict__ phi, const T *__restrict__ weights,
                                           const SpNode<T> &t, int n, int d0, int lseg, int lane, V4<T> &M)
{
    using R = Real<T>;
    const T inf = R::inf();
    int ni, nj, ns, nrev;
    const bool has = sp_neighbour(g, t.si, t.sj, t.s, n, &ni, &nj, &ns, &nrev) && t.active;
    V4<T> psi = splat(inf);
    T w = T(0);
    if (has && d0 < g.Dpad) {
        psi = load4(phi + sp_plane(g, sp_colour(ni, nj, ns), nrev) + sp_index(g, ni, nj) * g.Dpad + d0);
        w = weights[((long long)t.colour * g.colour_nodes + t.idx) * 9 + n];
    }
    T lam[4], p[4], sfx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) lam[k] = w * (T)(d0 + k);
    p[0] = psi.v[0] - lam[0];
#pragma unroll
    for (int k = 1; k < 4; ++k) p[k] = R::min(p[k - 1], psi.v[k] - lam[k]);
    sfx[3] = psi.v[3] + lam[3];
#pragma unroll
    for (int k = 2; k >= 0; --k) sfx[k] = R::min(sfx[k + 1], psi.v[k] + lam[k]);
    // every lane of the wavefront takes part in the cross-lane steps
    const T xf = excl_prefix_min<T, LPP>(p[3], lseg);
    const T xb = excl_suffix_min<T, LPP>(sfx[0], lseg, lane);
#pragma unroll
    for (int k = 0; k < 4; ++k) M.v[k] = R::min(R::min(p[k], xf) + lam[k], R::min(sfx[k], xb) - lam[k]);
    return has;
}

template <typename T, int LPP>
__global__ __launch_bounds__(kBlock) void sp_update_kernel(SpGeometry g, T *__restrict__ phi, const T *__restrict__ cost,
                                                           const int *__restrict__ Kt, const T *__restrict__ weights,
                                                           int mode, int colour, int diag, long long count)
{
    using R = Real<T>;
    constexpr int PPW = 64 / LPP;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane / LPP, lseg = lane % LPP, d0 = 4 * lseg;
    const long long slot = ((long long)blockIdx.x * (kBlock / 64) + wave) * PPW + q;
    SpNode<T> t = sp_pick<T>(g, slot, mode, colour, diag);
    t.active = t.active && slot < count;
    const long long node = (long long)t.colour * g.colour_nodes + t.idx;
    const int K = t.active ? Kt[node] : -1;
    V4<T> M[9];
    bool has[9];
#pragma unroll
    for (int n = 0; n < 9; ++n) has[n] = sp_message<T, LPP>(g, phi, weights, t, n, d0, lseg, lane, M[n]);
    if (!t.active || d0 > K) return;  // the lane's labels do not exist: +inf stays
    const V4<T> qc = load4(cost + node * g.Dpad + d0);
    V4<T> avg;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        T acc = T(0);
#pragma unroll
        for (int n = 0; n < 9; ++n) acc += has[n] ? M[n].v[k] : T(0);
        avg.v[k] = (qc.v[k] + acc) / (T)t.nn;
    }
#pragma unroll
    for (int n = 0; n < 9; ++n) {
        if (!has[n]) continue;
        V4<T> out;
#pragma unroll
        for (int k = 0; k < 4; ++k) out.v[k] = (d0 + k <= K) ? avg.v[k] - M[n].v[k] : R::inf();
        store4(phi + sp_plane(g, t.colour, n) + t.idx * g.Dpad + d0, out);
    }
}

// ---- energy and labels ---------------------------------------------------------------------------
// Per node: min_d f_t(d), argmin (first minimum), and min_d (phi[t][n][d] + M_n(d)) for the
// slots that own an edge (0 from superpixel 0 only; 5-8) -- diffusion_graph.rs:584-671.
template <typename T, int LPP>
__global__ __launch_bounds__(kBlock) void sp_energy_kernel(SpGeometry g, const T *__restrict__ phi,
                                                           const T *__restrict__ cost, const int *__restrict__ Kt,
                                                           const T *__restrict__ weights, int colour,
                                                           double *__restrict__ node_energy,
                                                           uint32_t *__restrict__ node_label)
{
    using R = Real<T>;
    constexpr int PPW = 64 / LPP;
    const T inf = R::inf();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane / LPP, lseg = lane % LPP, d0 = 4 * lseg;
    const long long slot = ((long long)blockIdx.x * (kBlock / 64) + wave) * PPW + q;
    SpNode<T> t = sp_pick<T>(g, slot, 0, colour, 0);
    t.active = t.active && slot < g.colour_nodes;
    const long long node = (long long)t.colour * g.colour_nodes + t.idx;
    const int K = t.active ? Kt[node] : -1;
    const bool mine = t.active && d0 <= K;
    V4<T> f = mine ? load4(cost + node * g.Dpad + d0) : splat(inf);
    V4<T> sum = splat(T(0));
    double e = 0.;
#pragma unroll
    for (int n = 0; n < 9; ++n) {
        V4<T> M;
        const bool has = sp_message<T, LPP>(g, phi, weights, t, n, d0, lseg, lane, M);
        V4<T> own = splat(T(0));
        if (has && mine) own = load4(phi + sp_plane(g, t.colour, n) + t.idx * g.Dpad + d0);
#pragma unroll
        for (int k = 0; k < 4; ++k) sum.v[k] += has ? own.v[k] : T(0);
        const bool owns_edge = n >= 5 || (n == 0 && t.s == 0);
        T em = inf;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (mine && d0 + k <= K) em = R::min(em, own.v[k] + M.v[k]);
        em = segment_min<T, LPP>(em);
        if (has && owns_edge) e += (double)em;
    }
    T best = inf;
    int arg = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const T val = f.v[k] - sum.v[k];
        if (mine && d0 + k <= K && val < best) { best = val; arg = d0 + k; }
    }
#pragma unroll
    for (int m = LPP / 2; m >= 1; m >>= 1) {
        const T ob = __shfl_xor(best, m);
        const int oa = __shfl_xor(arg, m);
        if (ob < best || (ob == best && oa < arg)) { best = ob; arg = oa; }
    }
    if (t.active && lseg == 0) {
        node_energy[node] = e + (double)best;
        node_label[node] = arg == 0x7fffffff ? 0u : (uint32_t)arg;
    }
}

// Per pixel: the label of its node (build_depth_map / simple_best_labeling,
// diffusion_graph.rs:525-536, crossing_out_graph.rs:412-429); pixels outside every window keep 0.
__global__ void sp_expand_kernel(SpGeometry g, const uint8_t *__restrict__ superpixels,
                                 const uint32_t *__restrict__ node_label, uint32_t *__restrict__ out)
{
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= (long long)g.H * g.W) return;
    const int i = (int)(k / g.W), j = (int)(k % g.W);
    const int si = i / g.sh, sj = j / g.sw;
    uint32_t v = 0;
    if (si < g.NI && sj < g.NJ) {
        const int s = superpixels[k];
        v = node_label[(long long)sp_colour(si, sj, s) * g.colour_nodes + sp_index(g, si, sj)];
    }
    out[k] = v;
}

// ---- crossing out --------------------------------------------------------------------------------
// crossing_out_graph.rs:85-236 without the O(D^2) edge masks, at the scale of this graph (a few
// thousand nodes): one block per node, one thread per label, edge penalties by brute force in f64
// in the reference's expression order -- perimeter * smoothing * |d - n_d| + phi_t + phi_t'
// (diffusion_graph.rs:168-172) -- so that the thresholds fall where the reference's do.
//   init:  alive(t, d) = d <= K and f_t(d) <= min f_t + eps;  gmin(t, n) = min over existing (d, n_d)
//   relax: alive(t, d) stays iff towards every existing neighbour some n_d has
//          g(d, n_d) <= gmin(t, n) + eps  and  (n_d alive there  or not  g'(n_d, d) <= gmin(t', n') + eps),
//          g' being the same edge as the neighbour prices it: its own weight, its own order of
//          additions, its own pair minimum.  The reference keeps a copy of every edge at each end
//          and clears a node's copy on the death of the neighbour's vertex only through the
//          neighbour's copy (crossing_out_graph.rs:161-236), so a copy whose twin was never set
//          outlives that vertex.  With the crate's own weights (symmetric) the two copies differ by
//          rounding at most; with weights set per node they differ outright.  The condition is
//          monotone in the masks: repeated until nothing changes it reaches the reference's fixed
//          point whatever the order of deletions.
template <typename T> struct SpCoArgs {
    const T *phi;
    const T *cost;
    const int *Kt;
    const double *perim;  // [colour][node][9]
    double lambda, eps;
    uint8_t *alive;       // [colour][node][Dpad]
    double *gmin;         // [colour][node][9]
    int *flags;           // [0] a vertex died, [1] a node has none left
};

template <typename T>
__global__ __launch_bounds__(kBlock) void sp_co_kernel(SpGeometry g, SpCoArgs<T> a, int relax)
{
    extern __shared__ double lds[];  // psi[9][Dpad] (neighbours' reverse rows), then 9 slot minima, then a scratch row
    const long long node = blockIdx.x;
    const int colour = (int)(node / g.colour_nodes);
    const long long idx = node % g.colour_nodes;
    const int si = (int)(idx / g.NJh), s = colour & 1;
    const int sj = 2 * (int)(idx % g.NJh) + ((si + (colour >> 1)) & 1);
    if (sj >= g.NJ) return;  // uniform over the block
    const int d = threadIdx.x;
    const int K = a.Kt[node];
    double *psi = lds, *red = lds + 9 * g.Dpad;
    __shared__ int any_alive;
    bool has[9];
    int Kn[9], twin[9];
    long long nnode[9];
    for (int n = 0; n < 9; ++n) {
        int ni, nj, ns, nrev;
        has[n] = sp_neighbour(g, si, sj, s, n, &ni, &nj, &ns, &nrev);
        Kn[n] = -1;
        nnode[n] = 0;
        twin[n] = 0;
        if (has[n]) {
            const int nc = sp_colour(ni, nj, ns);
            nnode[n] = (long long)nc * g.colour_nodes + sp_index(g, ni, nj);
            twin[n] = nrev;
            Kn[n] = a.Kt[nnode[n]];
            if (d < g.Dpad)
                psi[n * g.Dpad + d] = d <= Kn[n] ? (double)a.phi[sp_plane(g, nc, nrev) + sp_index(g, ni, nj) * g.Dpad + d] : 0.;
        }
    }
    if (threadIdx.x == 0) any_alive = 0;
    __syncthreads();
    const bool valid = d <= K;
    double own[9];
    for (int n = 0; n < 9; ++n) own[n] = (has[n] && valid) ? (double)a.phi[sp_plane(g, colour, n) + idx * g.Dpad + d] : 0.;
    auto edge = [&](int n, int nd) {
        const double diff = (double)(d > nd ? d - nd : nd - d);
        return a.perim[node * 9 + n] * a.lambda * diff + own[n] + psi[n * g.Dpad + nd];
    };
    // block-wide minimum of one value per thread
    auto block_min = [&](double v) {
        red[threadIdx.x] = v;
        __syncthreads();
        for (int m = kBlock / 2; m >= 1; m >>= 1) {
            if ((int)threadIdx.x < m) red[threadIdx.x] = red[threadIdx.x] < red[threadIdx.x + m] ? red[threadIdx.x] : red[threadIdx.x + m];
            __syncthreads();
        }
        const double r = red[0];
        __syncthreads();
        return r;
    };
    const double inf = __builtin_huge_val();
    if (!relax) {
        double f = inf;
        if (valid) {
            double sum = 0.;
            for (int n = 0; n < 9; ++n)
                if (has[n]) sum += own[n];
            f = (double)a.cost[node * g.Dpad + d] - sum;
        }
        const double fmin = block_min(f);
        if (d < g.Dpad) a.alive[node * g.Dpad + d] = valid && f <= fmin + a.eps;
        for (int n = 0; n < 9; ++n) {
            double m = inf;
            if (has[n] && valid)
                for (int nd = 0; nd <= Kn[n]; ++nd) {
                    const double e = edge(n, nd);
                    if (e < m) m = e;
                }
            m = block_min(m);
            if (threadIdx.x == 0) a.gmin[node * 9 + n] = m;
        }
        return;
    }
    bool keep = valid && a.alive[node * g.Dpad + d];
    if (keep)
        for (int n = 0; n < 9 && keep; ++n) {
            if (!has[n]) continue;
            const double limit = a.gmin[node * 9 + n] + a.eps;
            // the neighbour's copy of the edge: its weight, its order of additions, its pair minimum
            const double twin_weight = a.perim[nnode[n] * 9 + twin[n]];
            const double twin_limit = a.gmin[nnode[n] * 9 + twin[n]] + a.eps;
            bool found = false;
            for (int nd = 0; nd <= Kn[n] && !found; ++nd) {
                if (!(edge(n, nd) <= limit)) continue;
                const double diff = (double)(d > nd ? d - nd : nd - d);
                const bool twin_set = twin_weight * a.lambda * diff + psi[n * g.Dpad + nd] + own[n] <= twin_limit;
                found = a.alive[nnode[n] * g.Dpad + nd] || !twin_set;
            }
            keep = found;
        }
    if (valid && a.alive[node * g.Dpad + d] && !keep) {
        a.alive[node * g.Dpad + d] = 0;
        a.flags[0] = 1;
    }
    if (keep) any_alive = 1;
    __syncthreads();
    if (threadIdx.x == 0 && !any_alive) a.flags[1] = 1;
}

// smallest alive label per node (top_vertex_between_existing, crossing_out_graph.rs:381-394);
// 0xffffffff where none is left
__global__ void sp_co_labels_kernel(SpGeometry g, const uint8_t *__restrict__ alive, const int *__restrict__ Kt,
                                    uint32_t *__restrict__ node_label, int *__restrict__ lost)
{
    const long long node = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (node >= 4 * g.colour_nodes) return;
    const long long idx = node % g.colour_nodes;
    const int colour = (int)(node / g.colour_nodes);
    const int si = (int)(idx / g.NJh);
    const int sj = 2 * (int)(idx % g.NJh) + ((si + (colour >> 1)) & 1);
    if (sj >= g.NJ) return;
    uint32_t v = 0xffffffffu;
    for (int d = 0; d <= Kt[node]; ++d)
        if (alive[node * g.Dpad + d]) {
            v = (uint32_t)d;
            break;
        }
    node_label[node] = v;
    if (v == 0xffffffffu) *lost = 1;
}

template <typename T> __global__ void sp_init_phi_kernel(SpGeometry g, const int *__restrict__ Kt, T *__restrict__ phi)
{
    const long long total = 36 * g.plane_elems;
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (long long)gridDim.x * blockDim.x) {
        const int d = (int)(k % g.Dpad);
        const long long within = (k % g.plane_elems) / g.Dpad;
        const int colour = (int)(k / (9 * g.plane_elems));
        phi[k] = d <= Kt[(long long)colour * g.colour_nodes + within] ? T(0) : Real<T>::inf();
    }
}

}  // namespace ds
