/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's diffusion
 * algorithm.  Nothing under oracle/ is linked, imported or executed by the
 * product library; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker.
 *
 * Parity status: every function below is pinned at function level by the
 * reference's own known-answer tests (ported in tests/test_oracle_kat.py).
 * No reference test pins a whole multi-iteration run, an energy trajectory or
 * a disparity map, and the reference (a Rust crate) cannot be built here, so
 * whole-run results are pinned only through this restatement.
 *
 * This file is a "template": it is included twice by diffusion_oracle.c, once
 * with REAL=double / SFX=64 (the reference's own precision) and once with
 * REAL=float / SFX=32.  All arithmetic keeps the reference's operation order;
 * build with -ffp-contract=off.
 *
 * Layouts (all row-major, last index fastest):
 *   pot, dummy : [H][W][4][D]      (reference: potentials / dummy_potentials)
 *   vertices   : [H][W][D]         (reference: CrossingOutGraph.vertices)
 *   edges      : [H][W][D][4][D]   (reference: CrossingOutGraph.edges)
 */

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(CAT(og, SFX), CAT(_, name))
#define GRAPH CAT(og_graph, SFX)

typedef struct GRAPH {
    int H, W, D;
    REAL lambda;          /* smoothing_term */
    REAL lut[256][256];   /* lookup_table: |a - b| */
    REAL *pot;            /* potentials */
    REAL *dummy;          /* dummy_potentials */
    uint32_t *L, *R;      /* left_image, right_image: [H][W] */
    uint8_t *vertices;    /* crossing-out masks (allocated lazily) */
    uint8_t *edges;
} GRAPH;

#define POT(g, i, j, n, d) ((g)->pot[((((size_t)(i)) * (g)->W + (j)) * 4 + (n)) * (g)->D + (d)])
#define DUM(g, i, j, n, d) ((g)->dummy[((((size_t)(i)) * (g)->W + (j)) * 4 + (n)) * (g)->D + (d)])
#define VERT(g, i, j, d) ((g)->vertices[(((size_t)(i)) * (g)->W + (j)) * (g)->D + (d)])
#define EDGE(g, i, j, d, n, nd) \
    ((g)->edges[(((((size_t)(i)) * (g)->W + (j)) * (g)->D + (d)) * 4 + (n)) * (g)->D + (nd)])

/* diffusion_stereo/src/diffusion_graph.rs:65-88 (DiffusionGraph::initialize) */
GRAPH *FN(initialize)(const uint32_t *left, const uint32_t *right, int H, int W, int D, double lambda)
{
    GRAPH *g = (GRAPH *)calloc(1, sizeof(GRAPH));
    if (!g) return NULL;
    g->H = H; g->W = W; g->D = D; g->lambda = (REAL)lambda;
    for (int a = 0; a < 256; ++a)
        for (int b = 0; b < 256; ++b)
            g->lut[a][b] = (REAL)fabs((double)a - (double)b);
    size_t n = (size_t)H * W * 4 * D;
    g->pot = (REAL *)calloc(n ? n : 1, sizeof(REAL));
    g->dummy = (REAL *)calloc(n ? n : 1, sizeof(REAL));
    size_t npx = (size_t)H * W;
    g->L = (uint32_t *)malloc(sizeof(uint32_t) * (npx ? npx : 1));
    g->R = (uint32_t *)malloc(sizeof(uint32_t) * (npx ? npx : 1));
    memcpy(g->L, left, sizeof(uint32_t) * (size_t)H * W);
    memcpy(g->R, right, sizeof(uint32_t) * (size_t)H * W);
    return g;
}

void FN(destroy)(GRAPH *g)
{
    if (!g) return;
    free(g->pot); free(g->dummy); free(g->L); free(g->R);
    free(g->vertices); free(g->edges); free(g);
}

REAL *FN(potentials)(GRAPH *g) { return g->pot; }
REAL *FN(dummy_potentials)(GRAPH *g) { return g->dummy; }

/* diffusion_graph.rs:99-107 */
REAL FN(sum_of_potentials)(const GRAPH *g, int i, int j, int d)
{
    REAL s = 0;
    for (int n = 0; n < 4; ++n)
        if (og_neighbor_exists(i, j, n, g->H, g->W)) s += POT(g, i, j, n, d);
    return s;
}

/* diffusion_graph.rs:118-122 */
REAL FN(vertex_penalty_with_potentials)(const GRAPH *g, int i, int j, int d)
{
    return g->lut[g->L[(size_t)i * g->W + j]][g->R[(size_t)i * g->W + (j - d)]] -
           FN(sum_of_potentials)(g, i, j, d);
}

/* diffusion_graph.rs:135-141: (lambda*|d-n_d| + phi[t][n][d]) + phi[t'][n'][n_d] */
REAL FN(edge_penalty_with_potential)(const GRAPH *g, int i, int j, int n, int d, int n_d)
{
    int ni, nj, nn;
    og_neighbor_index(i, j, n, &ni, &nj, &nn);
    return g->lambda * g->lut[d][n_d] + POT(g, i, j, n, d) + POT(g, ni, nj, nn, n_d);
}

/* diffusion_graph.rs:152-169 */
int FN(edge_exists)(const GRAPH *g, int i, int j, int n, int d, int n_d)
{
    if (!og_neighbor_exists(i, j, n, g->H, g->W)) return 0;
    int ni, nj, nn;
    og_neighbor_index(i, j, n, &ni, &nj, &nn);
    if (j >= d && nj >= n_d) {
        if (n == 2 && n_d > d + 1) return 0;
        if (n == 0 && d > n_d + 1) return 0;
        return 1;
    }
    return 0;
}

/* diffusion_graph.rs:177-201 */
REAL FN(penalty)(const GRAPH *g, const int32_t *map)
{
    REAL p = 0;
    for (int i = 0; i < g->H; ++i)
        for (int j = 0; j < g->W; ++j) {
            int d = map[(size_t)i * g->W + j];
            if (j >= d) {
                p += FN(vertex_penalty_with_potentials)(g, i, j, d);
                for (int n = 2; n < 4; ++n)
                    if (og_neighbor_exists(i, j, n, g->H, g->W)) {
                        int ni, nj, nn;
                        og_neighbor_index(i, j, n, &ni, &nj, &nn);
                        int nd = map[(size_t)ni * g->W + nj];
                        if (FN(edge_exists)(g, i, j, n, d, nd))
                            p += FN(edge_penalty_with_potential)(g, i, j, n, d, nd);
                        else
                            p += (REAL)INFINITY;
                    }
            } else {
                p += (REAL)INFINITY;
            }
        }
    return p;
}

/* diffusion_graph.rs:212-224 */
REAL FN(min_edge_between_neighbors)(const GRAPH *g, int i, int j, int n, int d)
{
    REAL m = (REAL)INFINITY;
    for (int n_d = 0; n_d < g->D; ++n_d)
        if (FN(edge_exists)(g, i, j, n, d, n_d)) {
            REAL e = FN(edge_penalty_with_potential)(g, i, j, n, d, n_d);
            if (e < m) m = e;
        }
    return m;
}

/* diffusion_graph.rs:235-237 */
void FN(update_vertex_potential)(GRAPH *g, int i, int j, int n, int d)
{
    DUM(g, i, j, n, d) -= FN(min_edge_between_neighbors)(g, i, j, n, d);
}

/* diffusion_graph.rs:248-251 */
void FN(update_edge_potential)(GRAPH *g, int i, int j, int n, int d)
{
    DUM(g, i, j, n, d) += FN(vertex_penalty_with_potentials)(g, i, j, d) /
                          (REAL)og_number_of_neighbors(i, j, g->H, g->W);
}

static void FN(copy_dummy_to_pot)(GRAPH *g, int i, int j)
{
    for (int n = 0; n < 4; ++n)
        for (int d = 0; d < g->D; ++d) POT(g, i, j, n, d) = DUM(g, i, j, n, d);
}

/* diffusion_graph.rs:302-317 */
void FN(diffusion_act_vertexes)(GRAPH *g, int i, int j)
{
    for (int n = 0; n < 4; ++n)
        if (og_neighbor_exists(i, j, n, g->H, g->W))
            for (int d = 0; d < g->D; ++d)
                if (j >= d) FN(update_vertex_potential)(g, i, j, n, d);
    FN(copy_dummy_to_pot)(g, i, j);
}

/* diffusion_graph.rs:326-341 */
void FN(diffusion_act_edges)(GRAPH *g, int i, int j)
{
    for (int n = 0; n < 4; ++n)
        if (og_neighbor_exists(i, j, n, g->H, g->W))
            for (int d = 0; d < g->D; ++d)
                if (j >= d) FN(update_edge_potential)(g, i, j, n, d);
    FN(copy_dummy_to_pot)(g, i, j);
}

/* diffusion_graph.rs:264-293: one iteration, raster (Gauss-Seidel) order = the
 * reference's own schedule ("O-seq"). */
void FN(diffusion_act)(GRAPH *g)
{
    for (int i = 0; i < g->H; ++i)
        for (int j = 0; j < g->W; ++j) {
            FN(diffusion_act_vertexes)(g, i, j);
            FN(diffusion_act_edges)(g, i, j);
        }
}

/* Same per-pixel update (diffusion_graph.rs:302-341) under the red-black
 * schedule: all pixels with (i+j) even, then all with (i+j) odd ("O-cb").
 * Same-colour updates commute: a pixel update reads only its own rows and its
 * neighbours' (opposite colour) rows and writes only its own rows. */
void FN(diffusion_half_sweep)(GRAPH *g, int colour)
{
    for (int i = 0; i < g->H; ++i)
        for (int j = 0; j < g->W; ++j)
            if (((i + j) & 1) == colour) {
                FN(diffusion_act_vertexes)(g, i, j);
                FN(diffusion_act_edges)(g, i, j);
            }
}

void FN(diffusion_act_checkerboard)(GRAPH *g)
{
    FN(diffusion_half_sweep)(g, 0);
    FN(diffusion_half_sweep)(g, 1);
}

/* diffusion_graph.rs:456-470; first minimum wins (strict <). */
REAL FN(min_penalty_vertex)(const GRAPH *g, int i, int j, int *disparity)
{
    REAL m = (REAL)INFINITY;
    int best = 0;
    for (int d = 0; d < g->D; ++d)
        if (j >= d) {
            REAL v = FN(vertex_penalty_with_potentials)(g, i, j, d);
            if (v < m) { best = d; m = v; }
        }
    if (disparity) *disparity = best;
    return m;
}

/* diffusion_graph.rs:480-494 */
REAL FN(min_penalty_edge)(const GRAPH *g, int i, int j, int n)
{
    REAL m = (REAL)INFINITY;
    for (int d = 0; d < g->D; ++d)
        for (int n_d = 0; n_d < g->D; ++n_d)
            if (FN(edge_exists)(g, i, j, n, d, n_d)) {
                REAL e = FN(edge_penalty_with_potential)(g, i, j, n, d, n_d);
                if (e < m) m = e;
            }
    return m;
}

/* diffusion_graph.rs:502-518 */
REAL FN(energy)(const GRAPH *g)
{
    REAL e = 0;
    for (int i = 0; i < g->H; ++i)
        for (int j = 0; j < g->W; ++j) {
            e += FN(min_penalty_vertex)(g, i, j, NULL);
            for (int n = 2; n < 4; ++n)
                if (og_neighbor_exists(i, j, n, g->H, g->W)) e += FN(min_penalty_edge)(g, i, j, n);
        }
    return e;
}

/* diffusion_graph.rs:376-390 without the file write: per-pixel argmin of f_t. */
void FN(depth_map)(const GRAPH *g, int32_t *out)
{
    for (int i = 0; i < g->H; ++i)
        for (int j = 0; j < g->W; ++j) {
            int d;
            FN(min_penalty_vertex)(g, i, j, &d);
            out[(size_t)i * g->W + j] = d;
        }
}

/* Decode taken after the vertex ("collect") half-step of a pixel: the vertex
 * step (diffusion_graph.rs:302-317) turns phi[t][n][d] into
 * phi[t][n][d] - min_edge_between_neighbors(t,n,d); f_t evaluated with those
 * values (diffusion_graph.rs:118-122) is the belief b_t(d).  argmin as in
 * min_penalty_vertex (diffusion_graph.rs:456-470).  Nothing is modified.
 * belief_out (optional): [H][W][D], invalid entries 0. */
void FN(belief_labels)(const GRAPH *g, int32_t *labels, REAL *belief_out)
{
    for (int i = 0; i < g->H; ++i)
        for (int j = 0; j < g->W; ++j) {
            REAL m = (REAL)INFINITY;
            int best = 0;
            for (int d = 0; d < g->D; ++d) {
                if (belief_out) belief_out[((size_t)i * g->W + j) * g->D + d] = 0;
                if (j < d) continue;
                REAL s = 0;
                for (int n = 0; n < 4; ++n)
                    if (og_neighbor_exists(i, j, n, g->H, g->W))
                        s += POT(g, i, j, n, d) - FN(min_edge_between_neighbors)(g, i, j, n, d);
                REAL b = g->lut[g->L[(size_t)i * g->W + j]][g->R[(size_t)i * g->W + (j - d)]] - s;
                if (belief_out) belief_out[((size_t)i * g->W + j) * g->D + d] = b;
                if (b < m) { m = b; best = d; }
            }
            labels[(size_t)i * g->W + j] = best;
        }
}

/* ------------------------------------------------------------------------ */
/* Crossing-out graph: diffusion_stereo/src/crossing_out_graph.rs            */

/* crossing_out_graph.rs:51-59; fill = initial value of both masks (main.rs:49-50 uses true). */
int FN(co_alloc)(GRAPH *g, int fill)
{
    size_t nv = (size_t)g->H * g->W * g->D;
    size_t ne = nv * 4 * g->D;
    free(g->vertices); free(g->edges);
    g->vertices = (uint8_t *)malloc(nv ? nv : 1);
    g->edges = (uint8_t *)malloc(ne ? ne : 1);
    if (!g->vertices || !g->edges) return -1;
    memset(g->vertices, fill ? 1 : 0, nv);
    memset(g->edges, fill ? 1 : 0, ne);
    return 0;
}

uint8_t *FN(co_vertices)(GRAPH *g) { return g->vertices; }
uint8_t *FN(co_edges)(GRAPH *g) { return g->edges; }

/* crossing_out_graph.rs:78-93 */
void FN(co_initialize_vertices)(GRAPH *g, double epsilon)
{
    for (int i = 0; i < g->H; ++i)
        for (int j = 0; j < g->W; ++j) {
            REAL mn = FN(min_penalty_vertex)(g, i, j, NULL);
            for (int d = 0; d < g->D; ++d)
                VERT(g, i, j, d) =
                    (j >= d && FN(vertex_penalty_with_potentials)(g, i, j, d) <= mn + (REAL)epsilon) ? 1 : 0;
        }
}

/* crossing_out_graph.rs:102-130 (min_penalty_edge hoisted out of the d loop:
 * it does not depend on d, so the result is identical). */
void FN(co_initialize_edges)(GRAPH *g, double epsilon)
{
    for (int i = 0; i < g->H; ++i)
        for (int j = 0; j < g->W; ++j)
            for (int n = 0; n < 4; ++n) {
                int ex = og_neighbor_exists(i, j, n, g->H, g->W);
                REAL mn = ex ? FN(min_penalty_edge)(g, i, j, n) : 0;
                for (int d = 0; d < g->D; ++d)
                    for (int n_d = 0; n_d < g->D; ++n_d)
                        EDGE(g, i, j, d, n, n_d) =
                            (ex && FN(edge_exists)(g, i, j, n, d, n_d) &&
                             FN(edge_penalty_with_potential)(g, i, j, n, d, n_d) <= mn + (REAL)epsilon)
                                ? 1 : 0;
            }
}

/* crossing_out_graph.rs:67-70 */
void FN(co_initialize_with_epsilon)(GRAPH *g, double epsilon)
{
    FN(co_initialize_vertices)(g, epsilon);
    FN(co_initialize_edges)(g, epsilon);
}

/* crossing_out_graph.rs:135-183 */
void FN(co_crossing_out)(GRAPH *g)
{
    int changed = 1;
    while (changed) {
        changed = 0;
        for (int i = 0; i < g->H; ++i)
            for (int j = 0; j < g->W; ++j)
                for (int d = 0; d < g->D; ++d) {
                    if (j < d) continue;
                    if (!VERT(g, i, j, d)) {
                        for (int n = 0; n < 4; ++n)
                            if (og_neighbor_exists(i, j, n, g->H, g->W)) {
                                int ni, nj, nn;
                                og_neighbor_index(i, j, n, &ni, &nj, &nn);
                                for (int n_d = 0; n_d < g->D; ++n_d)
                                    if (FN(edge_exists)(g, i, j, n, d, n_d) && EDGE(g, i, j, d, n, n_d)) {
                                        EDGE(g, i, j, d, n, n_d) = 0;
                                        EDGE(g, ni, nj, n_d, nn, d) = 0;
                                        changed = 1;
                                    }
                            }
                    }
                    for (int n = 0; n < 4; ++n) {
                        if (!og_neighbor_exists(i, j, n, g->H, g->W)) continue;
                        int found = 0;
                        for (int n_d = 0; n_d < g->D; ++n_d)
                            if (FN(edge_exists)(g, i, j, n, d, n_d) && EDGE(g, i, j, d, n, n_d)) {
                                found = 1;
                                break;
                            }
                        if (!found && VERT(g, i, j, d)) {
                            VERT(g, i, j, d) = 0;
                            changed = 1;
                        }
                    }
                }
    }
}

/* crossing_out_graph.rs:196-211 */
int FN(co_vertices_exist)(const GRAPH *g)
{
    for (int i = 0; i < g->H; ++i)
        for (int j = 0; j < g->W; ++j) {
            int found = 0;
            for (int d = 0; d < g->D; ++d)
                if (j >= d && VERT(g, i, j, d)) found = 1;
            if (!found) return 0;
        }
    return 1;
}

/* crossing_out_graph.rs:214-246 */
int FN(co_edges_exist)(const GRAPH *g)
{
    for (int i = 0; i < g->H; ++i)
        for (int j = 0; j < g->W; ++j)
            for (int n = 0; n < 4; ++n) {
                if (!og_neighbor_exists(i, j, n, g->H, g->W)) continue;
                int found = 0;
                for (int d = 0; d < g->D && !found; ++d)
                    for (int n_d = 0; n_d < g->D; ++n_d)
                        if (FN(edge_exists)(g, i, j, n, d, n_d) && EDGE(g, i, j, d, n, n_d)) {
                            found = 1;
                            break;
                        }
                if (!found) return 0;
            }
    return 1;
}

/* crossing_out_graph.rs:187-193 */
int FN(co_is_not_empty)(const GRAPH *g)
{
    return FN(co_vertices_exist)(g) && FN(co_edges_exist)(g);
}

/* crossing_out_graph.rs:284-300; returns -1 where the reference panics. */
int FN(co_min_vertex_between_existing)(const GRAPH *g, int i, int j)
{
    REAL m = (REAL)INFINITY;
    int best = 0, found = 0;
    for (int d = 0; d < g->D; ++d)
        if (j >= d && VERT(g, i, j, d)) {
            REAL v = FN(vertex_penalty_with_potentials)(g, i, j, d);
            if (v < m) { m = v; best = d; found = 1; }
        }
    return found ? best : -1;
}

/* crossing_out_graph.rs:307-315; returns -1 where the reference panics. */
int FN(co_top_vertex_between_existing)(const GRAPH *g, int i, int j)
{
    for (int d = 0; d < g->D; ++d)
        if (j >= d && VERT(g, i, j, d)) return d;
    return -1;
}

/* crossing_out_graph.rs:318-327; returns -1 if some pixel has no vertex left. */
int FN(co_simple_best_labeling)(const GRAPH *g, int32_t *out)
{
    int rc = 0;
    for (int i = 0; i < g->H; ++i)
        for (int j = 0; j < g->W; ++j) {
            int d = FN(co_top_vertex_between_existing)(g, i, j);
            if (d < 0) rc = -1;
            out[(size_t)i * g->W + j] = d;
        }
    return rc;
}

/* crossing_out_graph.rs:336-345; returns -1 where the reference panics. */
int FN(co_cross_vertex)(GRAPH *g, int i, int j, int disparity)
{
    if (disparity < 0 || disparity >= g->D || !VERT(g, i, j, disparity)) return -1;
    for (int d = 0; d < g->D; ++d)
        if (VERT(g, i, j, d) && d != disparity) VERT(g, i, j, d) = 0;
    return 0;
}

/* crossing_out_graph.rs:349-364; returns -1 where the reference panics. */
int FN(co_find_best_labeling)(GRAPH *g, int32_t *out)
{
    for (int i = 0; i < g->H; ++i)
        for (int j = 0; j < g->W; ++j) {
            int d = FN(co_min_vertex_between_existing)(g, i, j);
            if (d < 0 || d > j) return -1;
            out[(size_t)i * g->W + j] = d;
            if (FN(co_cross_vertex)(g, i, j, d)) return -1;
            FN(co_crossing_out)(g);
            if (!FN(co_is_not_empty)(g)) return -1;
        }
    return 0;
}

/* crossing_out_graph.rs:254-275 with the schedule selectable and a cap on the
 * number of batches (the reference loops forever if the graph never becomes
 * epsilon-consistent).  schedule 0 = raster (reference), 1 = checkerboard.
 * The reference's diffusion() (diffusion_graph.rs:349-368) also evaluates
 * energy() after every iteration; that only prints, so it is skipped here.
 * Returns the total number of iterations done, or -1 if the cap was hit. */
int FN(co_diffusion_while_not_consistent)(GRAPH *g, double epsilon, int batch_size, int max_batches,
                                          int schedule)
{
    int done = 0;
    for (int b = 0; b < max_batches; ++b) {
        for (int k = 0; k < batch_size; ++k) {
            if (schedule == 0) FN(diffusion_act)(g);
            else FN(diffusion_act_checkerboard)(g);
        }
        done += batch_size;
        FN(co_initialize_with_epsilon)(g, epsilon);
        if (FN(co_is_not_empty)(g)) {
            FN(co_crossing_out)(g);
            if (FN(co_is_not_empty)(g)) return done;
        }
    }
    return -1;
}

/* ------------------------------------------------------------------------ */
/* Epsilon search: diffusion_stereo/src/epsilon_search.rs                      */

static int FN(cmp_double)(const void *a, const void *b)
{
    double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

/* epsilon_search.rs:39-86.  Gaps of every vertex to its pixel's minimum and of every
 * existing edge to its pixel pair's minimum, sorted ascending, then dedup_f64(tolerance).
 * *out is malloc'ed (caller frees with free()); returns the count, or -1 where the
 * reference's assert_ge! would fire (a gap below zero) or memory runs out. */
long FN(create_array_of_epsilons)(const GRAPH *g, double tolerance, double **out)
{
    size_t cap = 1024, n = 0;
    double *a = (double *)malloc(sizeof(double) * cap);
    if (!a) return -1;
#define PUSH(v)                                                           \
    do {                                                                  \
        if (n == cap) {                                                   \
            cap *= 2;                                                     \
            double *b_ = (double *)realloc(a, sizeof(double) * cap);      \
            if (!b_) { free(a); return -1; }                              \
            a = b_;                                                       \
        }                                                                 \
        a[n++] = (double)(v);                                             \
    } while (0)
    for (int i = 0; i < g->H; ++i)
        for (int j = 0; j < g->W; ++j) {
            REAL mv = FN(min_penalty_vertex)(g, i, j, NULL);
            for (int d = 0; d < g->D; ++d) {
                if (j < d) continue;
                REAL v = FN(vertex_penalty_with_potentials)(g, i, j, d);
                if (v < mv) { free(a); return -1; }
                PUSH(v - mv);
                for (int nb = 0; nb < 4; ++nb) {
                    if (!og_neighbor_exists(i, j, nb, g->H, g->W)) continue;
                    REAL me = FN(min_penalty_edge)(g, i, j, nb);
                    for (int n_d = 0; n_d < g->D; ++n_d)
                        if (FN(edge_exists)(g, i, j, nb, d, n_d)) {
                            REAL e = FN(edge_penalty_with_potential)(g, i, j, nb, d, n_d);
                            if (e < me) { free(a); return -1; }
                            PUSH(e - me);
                        }
                }
            }
        }
#undef PUSH
    qsort(a, n, sizeof(double), FN(cmp_double));
    double *u = (double *)malloc(sizeof(double) * (n ? n : 1));
    if (!u) { free(a); return -1; }
    int m = og_dedup_f64(a, (int)n, tolerance, u);
    free(a);
    *out = u;
    return m;
}

/* epsilon_search.rs:94-112: binary search for the first array element whose
 * epsilon-consistent sub-graph is non-empty.  Needs co_alloc'ed masks.  n >= 1. */
double FN(epsilon_search)(GRAPH *g, const double *array, long n)
{
    long first = 0, last = n - 1, median = (first + last) / 2;
    while (first < last) {
        FN(co_initialize_with_epsilon)(g, array[median]);
        FN(co_crossing_out)(g);
        if (FN(co_is_not_empty)(g)) last = median;
        else first = median + 1;
        median = (first + last) / 2;
    }
    return array[median];
}

#undef POT
#undef DUM
#undef VERT
#undef EDGE
#undef FN
#undef GRAPH
#undef CAT
#undef CAT_
