/*
 * TEST INFRASTRUCTURE ONLY -- O(D) CPU restatement of the checkerboard sweep.
 *
 * Included twice by fast_oracle.c (REAL=double SFX=64, REAL=float SFX=32).
 *
 * The reference's per-pixel update (diffusion_stereo/src/diffusion_graph.rs:
 * 302-341) is, for an existing neighbour slot n and a valid label d,
 *     phi[t][n][d] <- -M_n(d) + (q_t(d) + sum_m M_m(d)) / |N(t)|
 *     M_n(d) = min over allowed d' of ( lambda*|d-d'| + psi_n(d') ),
 *     psi_n = the neighbour's reverse row phi[t'][n'][.]
 * (the pixel's own old phi cancels algebraically: diffusion_graph.rs:135-141,
 * 212-237).  "Allowed" is edge_exists (diffusion_graph.rs:152-169): d' <= j',
 * and d' <= d+1 for the right neighbour, d' >= d-1 for the left one.
 *
 * The min over d' is evaluated here in O(D) through prefix / suffix minima of
 *     a(d') = psi(d') - fl(lambda*d')   and   b(d') = psi(d') + fl(lambda*d'),
 *     M(d)  = min( min_{d'<=d} a(d') + fl(lambda*d),  min_{d'>=d} b(d') - fl(lambda*d) )
 * which is order-independent (min is exactly associative), so a parallel scan
 * reproduces it bit for bit.  It differs from the reference's brute-force
 * evaluation only by rounding (a few ulp of |phi| per iteration); the tests
 * bound that difference against diffusion_oracle.c.
 *
 * Arrays use the reference layout pot[H][W][4][D]; entries with d > j are
 * never read or written (they stay 0, diffusion_graph.rs:306,330).
 * Build with -ffp-contract=off.
 */

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(CAT(fo, SFX), CAT(_, name))

#define FO_MAXD 256
#define RMIN(a, b) ((b) < (a) ? (b) : (a))
#define ROW(pot, W, D, i, j, n) ((pot) + ((((size_t)(i)) * (W) + (j)) * 4 + (n)) * (size_t)(D))

/* M_n(d) for d in [0, K], K = min(j, D-1), from the neighbour's reverse row
 * psi, whose valid labels are [0, Kn].  n selects the constraint. */
static void FN(message)(const REAL *psi, int Kn, int K, int n, const REAL *lam, REAL *M)
{
    const REAL INF = (REAL)INFINITY;
    REAL fwd[FO_MAXD], bwd[FO_MAXD];
    if (n != 0) { /* forward part: d' <= d */
        REAL run = INF;
        for (int d = 0; d <= K; ++d) {
            if (d <= Kn) { REAL a = psi[d] - lam[d]; run = RMIN(run, a); }
            fwd[d] = run + lam[d];
        }
    }
    if (n != 2) { /* backward part: d' >= d */
        REAL run = INF;
        int top = Kn > K ? Kn : K;
        for (int d = top; d >= 0; --d) {
            if (d <= Kn) { REAL b = psi[d] + lam[d]; run = RMIN(run, b); }
            if (d <= K) bwd[d] = run - lam[d];
        }
    }
    for (int d = 0; d <= K; ++d) {
        REAL m;
        if (n == 1 || n == 3) {
            m = RMIN(fwd[d], bwd[d]);
        } else if (n == 2) {
            REAL s = (d + 1 <= Kn) ? lam[1] + psi[d + 1] : INF;
            m = RMIN(fwd[d], s);
        } else {
            REAL s = (d >= 1) ? lam[1] + psi[d - 1] : INF;
            m = RMIN(bwd[d], s);
        }
        M[d] = m;
    }
}

/* Gathers the four messages of pixel (i,j); returns |N(t)|; ex[n] = slot exists. */
static int FN(messages)(const REAL *pot, int H, int W, int D, int i, int j, const REAL *lam,
                        REAL M[4][FO_MAXD], int ex[4])
{
    const int K = j < D - 1 ? j : D - 1;
    int nn = 0;
    for (int n = 0; n < 4; ++n) {
        int ni = i, nj = j, rn = 0;
        switch (n) {
        case 0: nj = j - 1; rn = 2; ex[n] = j > 0; break;
        case 1: ni = i - 1; rn = 3; ex[n] = i > 0; break;
        case 2: nj = j + 1; rn = 0; ex[n] = j + 1 < W; break;
        default: ni = i + 1; rn = 1; ex[n] = i + 1 < H; break;
        }
        if (!ex[n]) continue;
        ++nn;
        const int Kn = nj < D - 1 ? nj : D - 1;
        FN(message)(ROW(pot, W, D, ni, nj, rn), Kn, K, n, lam, M[n]);
    }
    return nn;
}

static void FN(lambda_table)(REAL lambda, int D, REAL *lam)
{
    for (int d = 0; d < D; ++d) lam[d] = lambda * (REAL)d;
    if (D < 2) lam[1] = lambda; /* lam[1] is used as "lambda * 1" */
}

static inline REAL FN(cost)(const uint8_t *L, const uint8_t *R, int W, int i, int j, int d)
{
    int a = L[(size_t)i * W + j], b = R[(size_t)i * W + j - d];
    return (REAL)(a > b ? a - b : b - a);
}

/* One colour's half-sweep over image rows [row_begin, row_end). */
void FN(half_sweep)(REAL *pot, const uint8_t *L, const uint8_t *R, int H, int W, int D, double lambda,
                    int colour, int row_begin, int row_end)
{
    REAL lam[FO_MAXD + 2];
    FN(lambda_table)((REAL)lambda, D, lam);
#pragma omp parallel for schedule(dynamic, 1)
    for (int i = row_begin; i < row_end; ++i) {
        REAL M[4][FO_MAXD];
        int ex[4];
        for (int j = (i + colour) & 1; j < W; j += 2) {
            const int K = j < D - 1 ? j : D - 1;
            const int nn = FN(messages)(pot, H, W, D, i, j, lam, M, ex);
            if (nn == 0) continue;
            for (int d = 0; d <= K; ++d) {
                REAL acc = 0;
                int first = 1;
                for (int n = 0; n < 4; ++n)
                    if (ex[n]) { acc = first ? M[n][d] : acc + M[n][d]; first = 0; }
                const REAL f = FN(cost)(L, R, W, i, j, d) + acc;
                const REAL avg = f / (REAL)nn;
                for (int n = 0; n < 4; ++n)
                    if (ex[n]) ROW(pot, W, D, i, j, n)[d] = avg - M[n][d];
            }
        }
    }
}

void FN(iterate)(REAL *pot, const uint8_t *L, const uint8_t *R, int H, int W, int D, double lambda,
                 int n_iters)
{
    for (int it = 0; it < n_iters; ++it) {
        FN(half_sweep)(pot, L, R, H, W, D, lambda, 0, 0, H);
        FN(half_sweep)(pot, L, R, H, W, D, lambda, 1, 0, H);
    }
}

/* f_t(d) = q_t(d) - sum_n phi[t][n][d], slots accumulated from 0.0 in order 0..3
 * (diffusion_graph.rs:99-122). */
static inline REAL FN(vertex_penalty)(const REAL *pot, const uint8_t *L, const uint8_t *R, int H, int W,
                                      int D, int i, int j, int d)
{
    REAL s = 0;
    if (j > 0) s += ROW(pot, W, D, i, j, 0)[d];
    if (i > 0) s += ROW(pot, W, D, i, j, 1)[d];
    if (j + 1 < W) s += ROW(pot, W, D, i, j, 2)[d];
    if (i + 1 < H) s += ROW(pot, W, D, i, j, 3)[d];
    return FN(cost)(L, R, W, i, j, d) - s;
}

/* energy (diffusion_graph.rs:502-518) with min_penalty_edge evaluated as
 * min_d( phi[t][n][d] + M_n(d) ); per-row partial sums, then summed in row order. */
double FN(energy)(const REAL *pot, const uint8_t *L, const uint8_t *R, int H, int W, int D, double lambda)
{
    REAL lam[FO_MAXD + 2];
    FN(lambda_table)((REAL)lambda, D, lam);
    double *rows = (double *)calloc((size_t)(H ? H : 1), sizeof(double));
#pragma omp parallel for schedule(dynamic, 1)
    for (int i = 0; i < H; ++i) {
        REAL M[4][FO_MAXD];
        int ex[4];
        double e = 0;
        for (int j = 0; j < W; ++j) {
            const int K = j < D - 1 ? j : D - 1;
            REAL mv = (REAL)INFINITY;
            for (int d = 0; d <= K; ++d) {
                REAL v = FN(vertex_penalty)(pot, L, R, H, W, D, i, j, d);
                mv = RMIN(mv, v);
            }
            e += (double)mv;
            FN(messages)(pot, H, W, D, i, j, lam, M, ex);
            for (int n = 2; n < 4; ++n)
                if (ex[n]) {
                    REAL me = (REAL)INFINITY;
                    for (int d = 0; d <= K; ++d) {
                        REAL v = ROW(pot, W, D, i, j, n)[d] + M[n][d];
                        me = RMIN(me, v);
                    }
                    e += (double)me;
                }
        }
        rows[i] = e;
    }
    double total = 0;
    for (int i = 0; i < H; ++i) total += rows[i];
    free(rows);
    return total;
}

/* mode 1: argmin_d f_t(d) (diffusion_graph.rs:456-470, first minimum wins);
 * mode 0: argmin_d of the belief b_t(d) = q_t(d) + sum_n M_n(d) (f_t as it
 * stands after the pixel's vertex step, diffusion_graph.rs:302-317). */
void FN(labels)(const REAL *pot, const uint8_t *L, const uint8_t *R, int H, int W, int D, double lambda,
                int mode, uint32_t *out)
{
    REAL lam[FO_MAXD + 2];
    FN(lambda_table)((REAL)lambda, D, lam);
#pragma omp parallel for schedule(dynamic, 1)
    for (int i = 0; i < H; ++i) {
        REAL M[4][FO_MAXD];
        int ex[4];
        for (int j = 0; j < W; ++j) {
            const int K = j < D - 1 ? j : D - 1;
            if (mode == 0) FN(messages)(pot, H, W, D, i, j, lam, M, ex);
            REAL best = (REAL)INFINITY;
            uint32_t arg = 0;
            for (int d = 0; d <= K; ++d) {
                REAL v;
                if (mode == 1) {
                    v = FN(vertex_penalty)(pot, L, R, H, W, D, i, j, d);
                } else {
                    REAL acc = 0;
                    int first = 1;
                    for (int n = 0; n < 4; ++n)
                        if (ex[n]) { acc = first ? M[n][d] : acc + M[n][d]; first = 0; }
                    v = FN(cost)(L, R, W, i, j, d) + acc;
                }
                if (v < best) { best = v; arg = (uint32_t)d; }
            }
            out[(size_t)i * W + j] = arg;
        }
    }
}

/* max over valid (t,d) of |f_t(d)|: 0 up to rounding after any whole iteration. */
double FN(max_residual)(const REAL *pot, const uint8_t *L, const uint8_t *R, int H, int W, int D)
{
    double worst = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(max : worst)
    for (int i = 0; i < H; ++i)
        for (int j = 0; j < W; ++j) {
            if (og_fast_nn(i, j, H, W) == 0) continue;
            const int K = j < D - 1 ? j : D - 1;
            for (int d = 0; d <= K; ++d) {
                double v = fabs((double)FN(vertex_penalty)(pot, L, R, H, W, D, i, j, d));
                if (v > worst) worst = v;
            }
        }
    return worst;
}

#undef RMIN
#undef ROW
#undef FN
#undef CAT
#undef CAT_
#undef FO_MAXD
