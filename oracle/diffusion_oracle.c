/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle for the diffusion_stereo hot path.
 *
 * A plain-C restatement of the reference algorithm, used by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker and
 * as the CPU baseline.  The product library never links or loads it.
 *
 * Parity status: pinned at function level by the reference's own unit-test
 * known answers (ported in tests/test_oracle_kat.py); whole-run trajectories
 * are pinned only through this restatement (the reference is a Rust crate and
 * no Rust toolchain exists in the build image, so it cannot be run).
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC (see oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---- grid topology: diffusion_stereo/src/utils.rs ---------------------- */

/* utils.rs:42-67; slot 0=left 1=top 2=right 3=bottom; -1 where the reference panics. */
int og_neighbor_exists(int i, int j, int n, int H, int W)
{
    switch (n) {
    case 0: return j > 0;
    case 1: return i > 0;
    case 2: return j + 1 < W;
    case 3: return i + 1 < H;
    default: return -1;
    }
}

/* utils.rs:75-83; returns -1 where the reference panics. */
int og_neighbor_index(int i, int j, int n, int *ni, int *nj, int *nn)
{
    switch (n) {
    case 0: *ni = i; *nj = j - 1; *nn = 2; return 0;
    case 1: *ni = i - 1; *nj = j; *nn = 3; return 0;
    case 2: *ni = i; *nj = j + 1; *nn = 0; return 0;
    case 3: *ni = i + 1; *nj = j; *nn = 1; return 0;
    default: return -1;
    }
}

/* utils.rs:93-101 */
int og_number_of_neighbors(int i, int j, int H, int W)
{
    int c = 0;
    for (int n = 0; n < 4; ++n)
        if (og_neighbor_exists(i, j, n, H, W) == 1) ++c;
    return c;
}

/* utils.rs:109-114 */
int og_approx_equal(double x, double y, double eps) { return !(fabs(x - y) > eps); }

/* utils.rs:121-130 */
int og_check_disparity_map(const int32_t *map, int H, int W)
{
    for (int i = 0; i < H; ++i)
        for (int j = 0; j + 1 < W; ++j)
            if (j < map[(size_t)i * W + j] || map[(size_t)i * W + j + 1] > map[(size_t)i * W + j] + 1)
                return 0;
    return 1;
}

/* utils.rs:137-163; in: sorted array of n values; out: capacity n; returns count. */
int og_dedup_f64(const double *a, int n, double tol, double *out)
{
    if (n <= 0) return 0;
    int *idx = (int *)malloc(sizeof(int) * (size_t)(n + 1));
    int cur = 0, i = 0;
    while (i < n) {
        idx[cur++] = i;
        int j = i + 1;
        while (j < n && og_approx_equal(a[i], a[j], tol)) ++j;
        i = j;
    }
    if (idx[cur - 1] != n - 1) idx[cur++] = n - 1;
    for (int k = 0; k < cur; ++k) out[k] = a[idx[k]];
    free(idx);
    return cur;
}

/* frees arrays handed out by og*_create_array_of_epsilons */
void og_free(void *p) { free(p); }

/* ---- cost volume ------------------------------------------------------- */

/* q_t(d) = |L[i][j] - R[i][j-d]| (diffusion_graph.rs:71-76,120-121) materialised
 * label-major as out[d][i][j]; entries with d > j (never read by the
 * reference, guard diffusion_graph.rs:306) are defined as 0. */
void og_cost_volume(const uint8_t *L, const uint8_t *R, int H, int W, int D, uint8_t *out)
{
    for (int d = 0; d < D; ++d)
        for (int i = 0; i < H; ++i)
            for (int j = 0; j < W; ++j) {
                uint8_t v = 0;
                if (j >= d) {
                    int a = L[(size_t)i * W + j], b = R[(size_t)i * W + j - d];
                    v = (uint8_t)(a > b ? a - b : b - a);
                }
                out[((size_t)d * H + i) * W + j] = v;
            }
}

/* ---- PGM I/O: diffusion_stereo/src/pgm_handler.rs ---------------------- */

/* pgm_handler.rs:36-74.  Line 1 = magic (not checked), line 2 = "W H",
 * line 3 = maxval, rest = whitespace separated integers (i32 cast to u32).
 * Two-call protocol: call with out == NULL to get W, H; then with a buffer of
 * W*H entries.  Returns 0, or -1 where the reference panics. */
/* diffusion_graph.rs:399-424 (box_blur), with the arithmetic of a release build: usize
 * subtraction wraps instead of panicking.  Consequences, restated literally:
 *  - the window is rows [i - r, i + r) x columns [j - r, j + r); for i < r (or j < r) the
 *    start wraps to a huge value, the range is empty and the pixel keeps only its own value;
 *  - a neighbour smaller than the centre gives a wrapped (huge) difference, i.e. "> threshold":
 *    only neighbours with centre <= value <= centre + threshold are averaged;
 *  - up starts at the centre's value, down at 1; result = up * 255 / (down * norm), integer
 *    division (the f64 round() of an integer changes nothing).
 * A debug build panics in the first two cases; that is not restated.  norm == 0 -> returns 1. */
int og_box_blur(const uint32_t *map, int H, int W, uint64_t threshold, uint64_t radius, uint64_t norm,
                uint64_t *out)
{
    if (norm == 0) return 1;
    for (int i = 0; i < H; ++i)
        for (int j = 0; j < W; ++j) {
            const uint64_t centre = map[(size_t)i * W + j];
            uint64_t up = centre, down = 1;
            const uint64_t i_lo = (uint64_t)i - radius, i_hi = (uint64_t)i + radius;  /* wraps */
            const uint64_t j_lo = (uint64_t)j - radius, j_hi = (uint64_t)j + radius;
            for (uint64_t n_i = i_lo; n_i < i_hi; ++n_i)
                for (uint64_t n_j = j_lo; n_j < j_hi; ++n_j)
                    if (n_i < (uint64_t)H && n_j < (uint64_t)W) {
                        const uint64_t v = map[n_i * (uint64_t)W + n_j];
                        uint64_t filter_value = 1;
                        if ((double)(v - centre) > (double)threshold) filter_value = 0;  /* wraps when v < centre */
                        up += v * filter_value;
                        down += filter_value;
                    }
            out[(size_t)i * W + j] = up * 255 / (down * norm);
        }
    return 0;
}

int og_pgm_reader(const char *path, uint32_t *out, int *W, int *H, int *maxval)
{
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    char line[4096];
    int rc = -1;
    long w = 0, h = 0, mx = 0;
    if (!fgets(line, sizeof line, f)) goto done;
    if (!fgets(line, sizeof line, f)) goto done;
    if (sscanf(line, "%ld %ld", &w, &h) != 2) goto done;
    if (!fgets(line, sizeof line, f)) goto done;
    if (sscanf(line, "%ld", &mx) != 1) goto done;
    *W = (int)w; *H = (int)h; *maxval = (int)mx;
    if (!out) { rc = 0; goto done; }
    {
        long count = 0, v;
        while (fscanf(f, "%ld", &v) == 1) {
            if (count < w * h) out[count] = (uint32_t)(int32_t)v;
            ++count;
        }
        /* anything left that is not whitespace is a parse failure in the reference */
        int c;
        while ((c = fgetc(f)) != EOF)
            if (c != ' ' && c != '\n' && c != '\t' && c != '\r') goto done;
        if (count != w * h) goto done;
    }
    rc = 0;
done:
    fclose(f);
    return rc;
}

/* pgm_handler.rs:122-133 */
int og_pgm_writer(const int32_t *m, int H, int W, const char *path, int max_intensity)
{
    FILE *f = fopen(path, "w");
    if (!f) return -1;
    fprintf(f, "P2\n%d %d\n%d\n", W, H, max_intensity);
    for (int i = 0; i < H; ++i) {
        for (int j = 0; j < W; ++j) fprintf(f, j ? " %d" : "%d", m[(size_t)i * W + j]);
        fprintf(f, "\n");
    }
    fclose(f);
    return 0;
}

/* ---- the graph itself, in the reference's f64 and in f32 ---------------- */

#define REAL double
#define SFX 64
#include "diffusion_oracle_impl.h"
#undef REAL
#undef SFX

#define REAL float
#define SFX 32
#include "diffusion_oracle_impl.h"
#undef REAL
#undef SFX
