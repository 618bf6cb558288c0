/* TEST INFRASTRUCTURE ONLY.  Included by superpixel_oracle.c once per precision (REAL,
 * SUFFIX); see the comment there for the formulas.  No include guard on purpose. */

#define PHI(phi, g, si, sj, s, n, d) ((phi)[(NODE(g, si, sj, s) * 9 + (n)) * (g)->D + (d)])

/* M_n(d) for d in [0, D) of node (si, sj, s); returns 0 if the slot has no neighbour */
static int SUFFIX(fast_message)(const SpGraph *g, const REAL *phi, int si, int sj, int s, int n, REAL *M)
{
    if (sp_neighbor_exists(si, sj, n, g->NI, g->NJ) != 1) return 0;
    int ni, nj, nindex;
    sp_neighbor_index(si, sj, n, s, &ni, &nj, &nindex);
    const int ns = sp_neighbor_superpixel(s, n);
    int Kn = sp_left_j_in_superpixel(g, ni, nj, ns);
    if (Kn > g->D - 1) Kn = g->D - 1;
    const REAL w = (REAL)(PERIM(g, si, sj, s, n) * g->lambda);
    const REAL *psi = &PHI(phi, g, ni, nj, ns, nindex, 0);
    REAL run = (REAL)INFINITY;
    for (int d = 0; d < g->D; ++d) {  /* prefix: a(d') = psi - w d' */
        if (d <= Kn) {
            const REAL a = psi[d] - w * (REAL)d;
            if (a < run) run = a;
        }
        M[d] = run + w * (REAL)d;
    }
    run = (REAL)INFINITY;
    for (int d = g->D - 1; d >= 0; --d) {  /* suffix: b(d') = psi + w d' */
        if (d <= Kn) {
            const REAL b = psi[d] + w * (REAL)d;
            if (b < run) run = b;
        }
        const REAL bwd = run - w * (REAL)d;
        if (bwd < M[d]) M[d] = bwd;
    }
    return 1;
}

static REAL SUFFIX(fast_cost)(const SpGraph *g, int si, int sj, int s, int d)
{
    REAL q = 0;
    for (int i = si * g->sh; i < si * g->sh + g->sh; ++i)
        for (int j = sj * g->sw; j < sj * g->sw + g->sw; ++j)
            if (g->superpixels[PIX(g, i, j)] == s)
                q += (REAL)lut((int)g->left[PIX(g, i, j)], (int)g->right[PIX(g, i, j - d)]);
    return q;
}

void SUFFIX(sp_fast_update_node)(const SpGraph *g, REAL *phi, int si, int sj, int s)
{
    const int D = g->D;
    int K = sp_left_j_in_superpixel(g, si, sj, s);
    if (K > D - 1) K = D - 1;
    REAL *M = (REAL *)malloc(sizeof(REAL) * 9 * (size_t)(D + 1));
    int has[9];
    for (int n = 0; n < 9; ++n) has[n] = SUFFIX(fast_message)(g, phi, si, sj, s, n, M + (size_t)n * D);
    const REAL nn = (REAL)sp_number_of_neighbors(si, sj, g->NI, g->NJ);
    for (int d = 0; d <= K; ++d) {
        REAL acc = 0;
        for (int n = 0; n < 9; ++n)
            if (has[n]) acc += M[(size_t)n * D + d];
        const REAL avg = (SUFFIX(fast_cost)(g, si, sj, s, d) + acc) / nn;
        for (int n = 0; n < 9; ++n)
            if (has[n]) PHI(phi, g, si, sj, s, n, d) = avg - M[(size_t)n * D + d];
    }
    free(M);
}

/* schedule 0: four colours, colour = 2 * ((si + sj) & 1) + s; 1: the reference's raster order */
void SUFFIX(sp_fast_iterate)(const SpGraph *g, REAL *phi, int n_iters, int schedule)
{
    for (int it = 0; it < n_iters; ++it) {
        if (schedule == 1) {
            for (int si = 0; si < g->NI; ++si)
                for (int sj = 0; sj < g->NJ; ++sj)
                    for (int s = 0; s < 2; ++s) SUFFIX(sp_fast_update_node)(g, phi, si, sj, s);
        } else {
            for (int colour = 0; colour < 4; ++colour)
                for (int si = 0; si < g->NI; ++si)
                    for (int sj = 0; sj < g->NJ; ++sj)
                        if (((si + sj) & 1) == (colour >> 1)) SUFFIX(sp_fast_update_node)(g, phi, si, sj, colour & 1);
        }
    }
}

/* f_t(d) = q_t(d) - sum of phi over the existing slots (slot order, from 0) */
static REAL SUFFIX(fast_vertex)(const SpGraph *g, const REAL *phi, int si, int sj, int s, int d)
{
    REAL sum = 0;
    for (int n = 0; n < 9; ++n)
        if (sp_neighbor_exists(si, sj, n, g->NI, g->NJ) == 1) sum += PHI(phi, g, si, sj, s, n, d);
    return SUFFIX(fast_cost)(g, si, sj, s, d) - sum;
}

double SUFFIX(sp_fast_energy)(const SpGraph *g, const REAL *phi)
{
    double e = 0.;
    REAL *M = (REAL *)malloc(sizeof(REAL) * (size_t)(g->D + 1));
    static const int possible[5] = {0, 5, 6, 7, 8};
    for (int si = 0; si < g->NI; ++si)
        for (int sj = 0; sj < g->NJ; ++sj)
            for (int s = 0; s < 2; ++s) {
                int K = sp_left_j_in_superpixel(g, si, sj, s);
                if (K > g->D - 1) K = g->D - 1;
                REAL m = (REAL)INFINITY;
                for (int d = 0; d <= K; ++d) {
                    const REAL f = SUFFIX(fast_vertex)(g, phi, si, sj, s, d);
                    if (f < m) m = f;
                }
                e += (double)m;
                for (int q = 0; q < 5; ++q) {
                    const int n = possible[q];
                    if (n == 0 && s == 1) continue;  /* the window's inner edge once */
                    if (!SUFFIX(fast_message)(g, phi, si, sj, s, n, M)) continue;
                    REAL em = (REAL)INFINITY;
                    for (int d = 0; d <= K; ++d) {
                        const REAL x = PHI(phi, g, si, sj, s, n, d) + M[d];
                        if (x < em) em = x;
                    }
                    e += (double)em;
                }
            }
    free(M);
    return e;
}

/* argmin_d f_t(d) per node (first minimum), spread over the node's pixels: out [H][W] */
void SUFFIX(sp_fast_depth_map)(const SpGraph *g, const REAL *phi, int64_t *out)
{
    memset(out, 0, sizeof(int64_t) * (size_t)g->H * g->W);
    for (int si = 0; si < g->NI; ++si)
        for (int sj = 0; sj < g->NJ; ++sj) {
            int k[2] = {0, 0};
            for (int s = 0; s < 2; ++s) {
                int K = sp_left_j_in_superpixel(g, si, sj, s);
                if (K > g->D - 1) K = g->D - 1;
                REAL m = (REAL)INFINITY;
                for (int d = 0; d <= K; ++d) {
                    const REAL f = SUFFIX(fast_vertex)(g, phi, si, sj, s, d);
                    if (f < m) { m = f; k[s] = d; }
                }
            }
            for (int i = si * g->sh; i < si * g->sh + g->sh; ++i)
                for (int j = sj * g->sw; j < sj * g->sw + g->sw; ++j) out[PIX(g, i, j)] = k[g->superpixels[PIX(g, i, j)]];
        }
}

#undef PHI
