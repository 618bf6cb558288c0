/*
 * diffstereo -- C ABI of the MI355X (gfx950) implementation of diffusion_stereo's
 * hot path: cost volume -> min-sum diffusion sweeps (red-black checkerboard) ->
 * energy / label extraction.
 *
 * The reference (Helga-Helga/stereo-vision, crate diffusion_stereo) has no FFI or
 * plugin boundary: `DiffusionGraph` is a plain Rust struct with public fields and
 * methods that `CrossingOutGraph` and `main` call in-process.  Each entry point
 * below therefore replaces one public method (or public-field access) of that
 * struct; the citation names it.  A Rust host links this library from build.rs
 * and keeps the crate's public functions as thin wrappers.
 *
 * Conventions
 *  - Every function returns DS_OK (0) or a negative DS_ERR_* code.  No C++
 *    exception and no HIP error crosses the boundary; ds_last_error() gives text.
 *  - A handle is used by one host thread at a time.  It owns its device memory
 *    and, unless ds_set_stream() was called, its own HIP stream.
 *  - Host buffers stay owned by the caller and are only touched during the call.
 *  - Images are 8-bit, row-major [H][W].  1 <= D <= 256 (the reference reuses its
 *    256x256 |a-b| table for |d-d'|, diffusion_graph.rs:71-76,137).
 *  - "phi" arrays at the boundary use the reference layout potentials[i][j][n][d]
 *    (diffusion_graph.rs:43): row-major [rows][W][4][D], element type float
 *    (DS_F32) or double (DS_F64); neighbour slot n: 0=left 1=top 2=right 3=bottom
 *    (utils.rs:29-34).  Entries the reference never touches (d > j, or slot
 *    without neighbour) read back as 0.
 *  - A handle may own only the row band [row_begin, row_end) of the image (one
 *    band per GPU).  It then also stores one halo row above and below; the host
 *    moves halo rows between neighbouring bands after each colour's half-sweep
 *    (ds_halo gives the device pointers).
 */
#ifndef DIFFSTEREO_H
#define DIFFSTEREO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ds_handle ds_handle;

enum { DS_F32 = 0, DS_F64 = 1 };

enum {
    DS_OK = 0,
    DS_ERR_ARG = -1,    /* invalid argument */
    DS_ERR_HIP = -2,    /* a HIP runtime call failed (text in ds_last_error) */
    DS_ERR_STATE = -3,  /* call not valid in the handle's current state */
    DS_ERR_NOMEM = -4   /* host or device allocation failed */
};

/* Replaces DiffusionGraph::initialize (diffusion_stereo/src/diffusion_graph.rs:65-88):
 * copies the images, allocates phi (all zero) on the current HIP device and builds
 * the cost volume.  lambda = smoothing_term. */
int ds_create(const uint8_t *left, const uint8_t *right, int H, int W, int D, double lambda,
              int dtype, ds_handle **out);

/* Same, for the band of image rows [row_begin, row_end) of an H x W image;
 * left/right are still the full images.  device < 0 keeps the current device. */
int ds_create_band(const uint8_t *left, const uint8_t *right, int H, int W, int D, double lambda,
                   int dtype, int row_begin, int row_end, int device, ds_handle **out);

void ds_destroy(ds_handle *h);

/* Text of the last error on this handle (or of the last failed ds_create* when h is NULL). */
const char *ds_last_error(const ds_handle *h);

/* Run all further work of this handle on the given hipStream_t instead of the handle's own
 * stream (NULL = the device's default stream).  Pending work is waited for first. */
int ds_set_stream(ds_handle *h, void *hip_stream);

/* The same without waiting for the stream that is being left: for callers that order the
 * two streams themselves (events / stream waits), e.g. to run a band's boundary rows and
 * their exchange beside the kernel that updates its interior. */
int ds_switch_stream(ds_handle *h, void *hip_stream);

/* Waits until all work queued by this handle has finished. */
int ds_synchronize(ds_handle *h);

/* Stage 1.  Replaces the implicit data term lookup_table[L[i][j]][R[i][j-d]]
 * (diffusion_graph.rs:71-76,120-121): materialises cost[d][i][j] = |L[i][j] - R[i][j-d]|
 * as 8-bit label-major [D][rows][W] (0 where d > j) and repacks it for the sweep.
 * ds_create* already calls it; calling it again rebuilds. */
int ds_build_cost(ds_handle *h);

/* Copies the label-major cost volume [D][rows][W] (rows = the handle's band) to the host. */
int ds_get_cost(ds_handle *h, uint8_t *out);

/* Stage 2.  Replaces n_iters calls of DiffusionGraph::diffusion_act
 * (diffusion_graph.rs:264-341) under the red-black schedule: all pixels with
 * (i+j) even, then all with (i+j) odd.  Synchronous at return.  Only valid on a
 * handle that owns the whole image (bands need the halo exchange between colours:
 * use ds_sweep).  The iterations run in a kernel that keeps the first
 * colour's potentials on chip (they are a function of the second colour's) and moves only
 * the second colour's planes; the call's last launch stores the first colour's planes as well,
 * so every other entry point sees the state 2 n_iters half-sweeps leave.  Same bits as the
 * half-sweeps; DS_FUSED=off at ds_create disables it.  Handles whose potentials fit the GPU's
 * 256 MiB last-level cache (eight planes of rows * ceil(W/2) * D elements: the reference's own
 * 400 x 400 x 40 images) are iterated by the half-sweeps by default, which are faster there;
 * DS_CACHE_RESIDENT=fused at ds_create keeps the fused kernels for them. */
int ds_iterate(ds_handle *h, int n_iters);

/* Whole iterations on a row band (no counterpart in the reference, which has one process).  The fused kernel recomputes the first colour's rows i-1..i+1 for every row i
 * of the second colour it updates, so next to a neighbouring band it must leave the band's
 * first / last row to the half-sweeps.  One iteration of a band, all calls asynchronous on the
 * handle's stream:
 *   ds_sweep(h, 0, rows)        first colour, the band's two first and two last rows;
 *   exchange ds_halo(0, ..)     as after any colour-0 half-sweep; wait for it;
 *   ds_sweep_next(h, rows)      second colour, the band's first and last row, written to the
 *                               planes the iteration is producing (not the current ones);
 *   exchange ds_halo_next(..)   those two rows to the neighbours' ds_halo_next receive rows --
 *                               this transfer runs while
 *   ds_iterate_rows(h, b, e)    updates rows [b, e) = ds_fused_rows() in one kernel and makes
 *                               the produced planes current.
 * After the last iteration ds_finish_iterations brings the first colour's planes up to date
 * (they are not written by ds_iterate_rows).  ds_fused_rows returns an empty range on handles
 * created with DS_FUSED=off.  The result is bit-identical to half-sweeps with ds_sweep.
 * ds_iterate_rows takes exactly the range ds_fused_rows reports.  Between ds_iterate_rows and
 * ds_finish_iterations the first colour's planes are out of date: the entry points that read phi
 * (ds_energy, ds_labels, ds_get_phi*, ds_set_phi*, ds_checksum, ds_co_begin, ds_epsilon_gaps,
 * ds_iterate_raster) return DS_ERR_STATE until ds_finish_iterations has run, which is a no-op
 * when nothing is pending. */
int ds_fused_rows(const ds_handle *h, int *row_begin, int *row_end);
/* ds_sweep (next = 0) or ds_sweep_next (next = 1, colour 1 only) over two disjoint row ranges in
 * one launch: a band's first and last rows. */
int ds_sweep_pair(ds_handle *h, int colour, int next, int row_begin, int row_end, int row_begin2, int row_end2);
int ds_iterate_rows(ds_handle *h, int row_begin, int row_end);
int ds_sweep_next(ds_handle *h, int row_begin, int row_end);
int ds_halo_next(ds_handle *h, int side, int which, void **dev_ptr, size_t *bytes);
int ds_finish_iterations(ds_handle *h);

/* The same n_iters calls of DiffusionGraph::diffusion_act, in the reference's own order:
 * the raster loop of diffusion_graph.rs:264-269 (rows, then columns, vertex step + edge
 * step at each pixel).  A pixel reads its left and top neighbours after, and its right and
 * bottom neighbours before, their update of the iteration; all four lie on the neighbouring
 * anti-diagonals, so the sweep runs one anti-diagonal i + j = k per launch, k ascending, and
 * gives the sequential result (H + W - 1 small launches per iteration: the parity mode, not
 * the fast one).  Synchronous at return.  Whole-image handles only. */
int ds_iterate_raster(ds_handle *h, int n_iters);

/* The same for a host that calls diffusion_act() one iteration at a time (the reference's
 * diffusion() does, diffusion_graph.rs:355-365): the call only notes n_iters more iterations and
 * returns; they are queued as ONE batch -- whose launches carry a diagonal of every iteration in
 * flight, see ds_iterate_raster -- as soon as anything needs the result: an entry point that
 * reads or changes phi, or ds_synchronize.  n calls of ds_iterate_raster_async(h, 1) followed by
 * ds_energy cost what ds_iterate_raster(h, n) costs (H + W - 1 + 2 (n - 1) dependent launches)
 * instead of n (H + W - 1).  Same bits.  No counterpart in the reference.  Whole-image handles. */
int ds_iterate_raster_async(ds_handle *h, int n_iters);

/* One colour's half-sweep (diffusion_act_vertexes + diffusion_act_edges,
 * diffusion_graph.rs:302-341, at every pixel of that colour) over the image rows
 * [row_begin, row_end), which must lie inside the handle's band.  Asynchronous:
 * queued on the handle's stream. */
int ds_sweep(ds_handle *h, int colour, int row_begin, int row_end);

/* Replaces DiffusionGraph::energy (diffusion_graph.rs:502-518) for the handle's
 * rows: sum of min_d f_t(d) plus, for slots 2 and 3, min_{d,d'} g_tt'(d,d').  The
 * edge to the row below a band uses the bottom halo.  Band energies add up to the
 * image's energy.  *max_residual (may be NULL) receives max |f_t(d)| over the band. */
int ds_energy(ds_handle *h, double *energy, double *max_residual);

/* ds_energy for callers that cannot allocate (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  The reduction's
 * partial sums live in a static device array of the library (32 MiB per GPU, held until the
 * process ends) instead of the handle's buffers; the call allocates nothing, copies nothing and
 * does not synchronise: it queues four kernels on the handle's stream, the last of which writes
 * result_dev[0] = energy and result_dev[1] = max |f_t(d)|.  result_dev: two doubles of device
 * memory or pinned host memory, owned by the caller (anything else is DS_ERR_ARG).
 * Limit: the handle may need at most 2 097 152 partial sums (2 x blocks per row x rows of the
 * band: a whole 3840 x 2160 x 256 image needs 2 073 600); larger handles get DS_ERR_ARG.
 * All calls on one GPU share the array: calls that overlap in time on one GPU (two handles
 * on different streams) race; calls on one stream, or ordered by the caller, do not.
 * Same value as ds_energy, bit for bit (same kernels, same order of summation). */
int ds_energy_static(ds_handle *h, double *result_dev);

/* Stage 3.  mode 1 replaces min_penalty_vertex / build_depth_map
 * (diffusion_graph.rs:376-390,456-470): argmin_d f_t(d), first minimum wins.
 * mode 0 takes the same argmin of f_t as it stands after the pixel's vertex step
 * (diffusion_graph.rs:302-317), i.e. of q_t(d) + sum_n min_d' (lambda|d-d'| + phi_t'n'(d')).
 * The inner minima are evaluated as the sweep evaluates them (prefix / suffix minima), not label
 * pair by label pair: where two labels tie in exact arithmetic -- neighbouring labels of equal
 * cost on messages that rise and fall by lambda; about one pixel in 10 000 on random inputs --
 * the two roundings may order them differently and mode 0 may return the other label of the tie.
 * out: [rows][W]. */
int ds_labels(ds_handle *h, int mode, uint32_t *out);

/* Replaces CrossingOutGraph::initialize_with_epsilon + crossing_out + is_not_empty +
 * simple_best_labeling (diffusion_stereo/src/crossing_out_graph.rs:67-183,187-246,307-327) on the
 * current phi: vertices within epsilon of their pixel's minimum, edges within epsilon of their
 * pixel pair's minimum, deletion to the greatest consistent sub-graph, then the smallest
 * surviving label per pixel.  No O(D^2) edge mask is kept: edge tests are min-plus products,
 * re-evaluated in whole-image passes until nothing changes (at most max_passes, else
 * DS_ERR_STATE).  labels: [H][W], 0xffffffff where a pixel lost every label; alive_out
 * (optional): [H][W][D] bytes, the reference's `vertices`; *not_empty = is_not_empty();
 * *passes = relaxation passes run.  Whole-image handles only.
 * The comparisons run in the handle's precision, with the penalties evaluated as the sweep
 * evaluates them (prefix / suffix minima), not in the reference's expression order: a vertex or
 * edge whose gap differs from epsilon by less than the rounding of those sums (a few ulp of the
 * potentials' size) may fall on the other side of the threshold than in the reference.  That
 * matters only for an epsilon that is itself one of the gaps -- the candidates of epsilon search
 * (ds_epsilon_gaps yields them exactly, in f64, in the reference's order): the search's
 * predicate is then reproduced up to that rounding, i.e. the epsilon found may be a neighbour
 * in the sorted array of the one the reference finds.  On an f32 handle an epsilon below
 * 2^-23 * (255 + lambda * (D - 1)) -- the spacing of f32 numbers at the size the potentials
 * reach -- is DS_ERR_ARG (ds_co_begin likewise): the reference's own default, 1 / (10 W H),
 * needs an f64 handle from about 50 x 50 pixels on. */
int ds_crossing_out(ds_handle *h, double epsilon, int max_passes, uint32_t *labels, uint8_t *alive_out,
                    int *not_empty, int *passes);
/* The same with every comparison made as the reference makes it (crossing_out_graph.rs:78-130):
 * in f64 whatever the handle's type, each penalty in the reference's expression order from the
 * side of the pixel that owns the mask entry, thresholds min + epsilon.  The reference keeps two
 * copies of every edge, one per end, rounded in different orders; its crossing_out (:135-183)
 * clears a pixel's copy on the death of the neighbour's vertex only through the neighbour's copy,
 * so a copy whose twin was never set outlives that vertex -- reproduced.  Result: the reference's
 * vertex masks for ANY epsilon, including one that is itself a gap of the graph (the candidates of
 * epsilon search), where ds_crossing_out may fall on the other side of a threshold.  Price:
 * O(D^2) per pixel looked at instead of O(D) (1920 x 1080 x 128: 0.1 s for a pass over every
 * pixel; later passes look only at pixels with a neighbour whose mask has changed), and H*W*5
 * doubles + 2*H*W ints.  Whole-image handles.  ds_find_best_labeling may follow.
 * An empty result (*not_empty = 0): with alive_out the passes go on to the reference's fixed
 * point, where vertices kept by such a copy may still stand, and alive_out / labels hold them;
 * without alive_out the call returns as soon as a pixel has lost every label, with every label
 * 0xffffffff (the verdict is the same, and the search for epsilon asks thousands of times). */
int ds_crossing_out_exact(ds_handle *h, double epsilon, int max_passes, uint32_t *labels, uint8_t *alive_out,
                          int *not_empty, int *passes);

/* Replaces CrossingOutGraph::find_best_labeling (diffusion_stereo/src/crossing_out_graph.rs:
 * 349-364, with min_vertex_between_existing :284-300 and cross_vertex :336-345) on the masks the
 * handle holds after ds_crossing_out (or ds_co_begin + ds_co_pass to the fixed point) for the
 * phi they were made from: pixels in raster order; each keeps, of its alive labels, the one with
 * the smallest f_t(d) (f64, the first of equals), its other labels are crossed out and the masks
 * are relaxed to their fixed point again (at most max_passes passes each time) before the next
 * pixel is looked at.  labels: [H][W], the one label each pixel has left at the end.
 * *not_empty = 0 where the reference panics ("Graph is empty", or a pixel without labels): the
 * labels are then not meaningful (0xffffffff at pixels without labels).  *choices (optional) =
 * pixels that still had more than one label when their turn came.  Sequential by definition,
 * but only over the pixels that have several labels when the call starts (masks only shrink,
 * so no other pixel can have a choice later): one small launch per such pixel, and after a
 * choice relaxation passes over a window that grows by one pixel per pass around it -- the
 * same masks as whole-image passes, since a deletion can only follow one at a neighbour.
 * Masks left short of their fixed point (ds_co_pass stopped while it still reported changes) are
 * first relaxed to it with whole-image passes.  max_passes bounds each of these loops; a window
 * pass carries a deletion one pixel further, a whole-image pass (in place, raster order of
 * blocks) usually further, so a bound tuned for ds_crossing_out may be too tight here.
 * Whole-image handles only. */
int ds_find_best_labeling(ds_handle *h, int max_passes, uint32_t *labels, int *not_empty, long long *choices);

/* Replaces create_array_of_epsilons before its tolerance dedup
 * (diffusion_stereo/src/epsilon_search.rs:39-74): for the current phi, every vertex's penalty
 * minus its pixel's minimum and every edge's penalty minus its pixel pair's minimum (each edge
 * from both of its ends, as the reference walks them), computed in f64 in the reference's
 * expression order, sorted ascending on the device, exact duplicates removed.  *count = how
 * many values there are; they stay on the device until ds_epsilon_gaps_fetch copies them to
 * out (count doubles) and releases them.  *largest_repeated (optional) = 1 if the largest
 * value occurred more than once before the duplicates went: utils.rs:137-163 (dedup_f64)
 * appends the last element once more when its run is longer than one, so a caller that wants
 * the reference's array element for element applies dedup_f64 with its tolerance and, if the
 * last value stands alone there but was repeated, appends it again.  epsilon_search.rs:84-112
 * then runs with ds_crossing_out as the predicate.
 * Needs H*W*D*(1+4D) doubles three times over on the device (DS_ERR_NOMEM otherwise): for
 * images of the reference's size.  Whole-image handles only. */
int ds_epsilon_gaps(ds_handle *h, size_t *count, int *largest_repeated);
int ds_epsilon_gaps_fetch(ds_handle *h, double *out);

/* Replaces dedup_f64 (diffusion_stereo/src/utils.rs:137-163) for what ds_epsilon_gaps returns;
 * host code only, no handle.  sorted[0..n) ascending; out receives the first value of every run
 * of values within `tolerance` of that first value (utils.rs:109-114: not |x - y| > tolerance)
 * and then the last value once more if its run is longer than one value or largest_repeated is
 * set (the flag of ds_epsilon_gaps: exact duplicates of the largest value were removed on the
 * device).  out needs room for n + 1 values; *n_out = how many were written. */
int ds_dedup_gaps(const double *sorted, size_t n, double tolerance, int largest_repeated, double *out, size_t *n_out);

/* Replaces epsilon_search(create_array_of_epsilons(graph, 0.0)) (diffusion_stereo/src/
 * epsilon_search.rs:39-112) for images whose array of gaps cannot exist (H*W*D*(1+4D) values:
 * 2.2e12 at 3840 x 2160 x 256): the smallest gap epsilon for which the crossing-out leaves a
 * non-empty graph, or the largest gap if none but it does (the reference's loop never tries its
 * array's last element).  No array: the crossing-out is asked about real numbers until the
 * interval that holds the answer is a few candidates wide (the predicate only changes at gaps),
 * then the gaps above the interval's lower end are produced one at a time (ds_epsilon_next)
 * and tried in ascending order.  Memory: H*W*5 doubles beside the crossing-out's own state.
 * exact = 1: the predicate is ds_crossing_out_exact (the reference's comparisons bit for bit;
 * f32 handles allowed, their potentials are widened).  On f64 handles ds_crossing_out still does
 * the halving first; the interval it ends with is widened by a bound on its rounding, confirmed
 * at both ends with the reference's comparisons, and they finish the search inside it.
 * exact = 0: ds_crossing_out alone (O(D) per pixel and pass, the rounding described there: the
 * result may be a neighbour, in the sorted array, of the reference's; f64 handles only).
 * With tolerance 0 the reference's array is the set of distinct gaps and the result is the
 * reference's; its tolerance > 0 thins the array out to one value per run of values within the
 * tolerance, which needs the whole sorted array and is not reproduced (the result here is then
 * <= the reference's).  Whole-image handles only.
 * stats (optional, 3 values): crossing-outs run, gap passes run, relaxation passes in total. */
int ds_epsilon_search(ds_handle *h, int exact, int max_passes, double *epsilon, long long *stats);
/* One pass over every gap of create_array_of_epsilons (same values as ds_epsilon_gaps, nothing
 * stored): *next = the smallest gap that is > lo (+inf if there is none; lo = -1 gives the
 * smallest gap), *largest = the largest gap, *not_above = how many gaps (of the unsorted,
 * undeduplicated array) are <= lo.  Any of the three may be NULL.  Whole-image handles. */
int ds_epsilon_next(ds_handle *h, double lo, double *next, double *largest, unsigned long long *not_above);

/* The same computation in steps, for row bands (one handle per band; the halos of phi must be
 * current, as after ds_sweep + exchange):
 *   ds_co_begin   vertex masks and pair minima of the band's pixels (initialize_with_epsilon);
 *   ds_co_halo    device pointer / size of a mask halo row, same side / which meaning as
 *                 ds_halo: after ds_co_begin and after every pass, a band sends its first and
 *                 last row's masks (both colours) to the bands above / below;
 *   ds_co_pass    one relaxation pass over the band; *changed: a vertex died, *lost: a pixel
 *                 has no vertex left.  The caller ORs both over all bands; it stops when no
 *                 band changed, or when any band lost a pixel (the closure is then empty: the
 *                 pixel grid is connected);
 *   ds_co_finish  empty != 0 clears the band's masks (the early stop above); then labels /
 *                 alive_out / *not_empty of the band's rows as in ds_crossing_out. */
int ds_co_begin(ds_handle *h, double epsilon);
int ds_co_pass(ds_handle *h, int *changed, int *lost);
int ds_co_halo(ds_handle *h, int colour, int side, int which, void **dev_ptr, size_t *bytes);
int ds_co_finish(ds_handle *h, int empty, uint32_t *labels, uint8_t *alive_out, int *not_empty);

/* Replace direct access to the public field `potentials` (diffusion_graph.rs:43).
 * Rows [row_begin, row_end) may include the band's halo rows. */
int ds_get_phi_rows(ds_handle *h, int row_begin, int row_end, void *out);
int ds_set_phi_rows(ds_handle *h, int row_begin, int row_end, const void *in);
/* The handle's own rows (no halo). */
int ds_get_phi(ds_handle *h, void *out);
int ds_set_phi(ds_handle *h, const void *in);

/* Halo rows of a band.  side: 0 = top edge of the band, 1 = bottom edge.
 * which: 0 = the row to send to the neighbouring band after `colour`'s half-sweep
 * (top: slot-1 messages of the band's first row; bottom: slot-3 messages of its
 * last row), 1 = where the neighbour's row is received (top: slot-3 messages of the
 * row above; bottom: slot-1 messages of the row below).  Both are contiguous
 * device buffers of *bytes bytes with identical internal layout on both sides. */
int ds_halo(ds_handle *h, int colour, int side, int which, void **dev_ptr, size_t *bytes);

/* ---- Row bands over the GPUs of a node: one process (rank) per GPU ------------------------
 *
 * No counterpart in the reference, which is one process with one raster loop
 * (diffusion_stereo/src/diffusion_graph.rs:264-293, main.rs:36-98); under the red-black
 * schedule its image splits into horizontal bands that exchange one row of messages with each
 * neighbouring band after each colour.  Rank r of `world` owns the r-th band (ds_create_band)
 * and its neighbours are the ranks r - 1 and r + 1.  The entry points below run the whole
 * protocol inside the library -- boundary rows first, on a stream of the library's own beside
 * the kernel that updates the band's interior, the exchanges queued behind them, explicit
 * events between the two streams (no stream priorities) -- so that a host only says
 * "iterate".  Results are bit-identical to the unpartitioned image.
 *
 * The halo rows travel through a ds_transport.  The library's own is RCCL (ncclSend / ncclRecv
 * inside ncclGroupStart / ncclGroupEnd on the boundary stream, ncclAllReduce for the scalars;
 * librccl is loaded with dlopen when ds_comm_* is first used, so hosts that never use bands
 * need not link it).  A host may plug in another transport (tests do: gloo through host
 * memory, several ranks sharing one GPU). */
typedef struct ds_comm ds_comm;

enum { DS_COMM_ID_BYTES = 128 };
enum { DS_REDUCE_SUM_F64 = 0, DS_REDUCE_MAX_F64 = 1, DS_REDUCE_MAX_I32 = 2, DS_REDUCE_MIN_I32 = 3 };

typedef struct ds_transport {
    void *ctx;
    /* For k < n: send bytes[k] bytes of device memory at send[k] to rank peers[k] and receive
     * as many from that rank into recv[k].  The n transfers form one group (no transfer may
     * wait for another of the group; with the same peer twice, sends and receives match in
     * order).  Stream-ordered on hip_stream: the data is read after the work queued there
     * before the call, and work queued afterwards sees what was received (an implementation
     * may simply block the host until that holds).  Returns 0 or non-zero. */
    int (*exchange)(void *ctx, int n, const int *peers, void *const *send, void *const *recv,
                    const size_t *bytes, void *hip_stream);
    /* In-place reduction over all ranks of `count` values at dev (device memory), kind =
     * DS_REDUCE_*; stream-ordered on hip_stream like exchange. */
    int (*all_reduce)(void *ctx, void *dev, int count, int kind, void *hip_stream);
    /* Text of the last failure (may be NULL). */
    const char *(*last_error)(void *ctx);
} ds_transport;

/* RCCL communicator.  Rank 0 calls ds_comm_unique_id and hands the 128 bytes to the other
 * ranks by whatever means the host has (a file, a socket, MPI, torch.distributed); then
 * every rank calls ds_comm_create with its rank, the number of ranks and its HIP device
 * (collective: returns when all ranks have called it). */
int ds_comm_unique_id(uint8_t id[DS_COMM_ID_BYTES]);
int ds_comm_create(const uint8_t id[DS_COMM_ID_BYTES], int rank, int world, int device, ds_comm **out);
void ds_comm_destroy(ds_comm *c);
const char *ds_comm_last_error(const ds_comm *c);

/* Tells a band handle how its halo rows travel and who its neighbours are: up_rank owns the
 * band above, down_rank the band below, -1 = none (the image's edge).  ds_band_attach_comm
 * uses the library's RCCL transport (the communicator must outlive the handle's use of it);
 * ds_band_attach copies the given table.  Creates the handle's boundary stream and events.
 * Collective when the band has a neighbour: the ranks agree (two one-word reductions through
 * the transport) that all bands run the same halo protocol, DS_ERR_STATE otherwise.
 *
 * Halo depth.  By default a band keeps one halo row per neighbour and a two-iteration step
 * exchanges one row four times, each exchange depending on the one before.  With
 * DS_BAND_HALO=deep in the environment at ds_create_band, a band of 16 rows or more (with
 * an interior for the two-iteration kernel, and 4 rows or more of image beyond it wherever it
 * has a neighbour) keeps FOUR rows of each neighbouring band: the two-iteration kernel then takes
 * every row of the band itself and a step needs ONE grouped exchange (four rows of the second
 * colour's four planes per neighbour: 31.5 MB at 4K x 256 against 4 x 2 MB), queued behind the
 * band's first and last four rows and running beside the next step's long launch.  Same results,
 * bit for bit.  On such a handle the entry points for hosts that sequence a band themselves
 * (ds_sweep, ds_sweep_pair, ds_sweep_next, ds_iterate_rows, ds_finish_iterations, ds_halo,
 * ds_halo_next) return DS_ERR_STATE; every other entry point sees the band's own rows as on any
 * band. */
int ds_band_attach_comm(ds_handle *h, ds_comm *c, int up_rank, int down_rank);
int ds_band_attach(ds_handle *h, const ds_transport *t, int up_rank, int down_rank);

/* Brings both colours' halo rows up to date (after ds_set_phi on the bands).  Collective over
 * the neighbouring ranks, synchronous at return. */
int ds_band_exchange(ds_handle *h);

/* n_iters x diffusion_act (diffusion_graph.rs:264-341) over all bands; every rank calls it
 * with the same n_iters.  A band of 8 rows or more whose handle has the fused kernel runs, per
 * iteration: on the boundary stream the first colour's two first / two last rows, their
 * exchange, the second colour's first / last row into the planes the iteration is producing,
 * their exchange; on the handle's stream, at the same time, the kernel that updates all other
 * rows keeping the first colour on chip.  Shorter bands sweep colour by colour (boundary rows,
 * exchange beside the interior rows).  Both kinds exchange the same messages, so they may be
 * neighbours.  Synchronous at return; every other entry point sees the same state as after
 * 2 n half-sweeps. */
int ds_band_iterate(ds_handle *h, int n_iters);

/* DiffusionGraph::energy (diffusion_graph.rs:502-518) of the whole image: ds_energy of the
 * band, summed over the ranks (max_residual: maximum over the ranks; may be NULL). */
int ds_band_energy(ds_handle *h, double *energy, double *max_residual);

/* ds_crossing_out over all bands: ds_co_begin, the mask halo rows to the neighbours, then
 * passes (each followed by a reduction of the two flags over the ranks and the mask halo
 * exchange) until no band changed or some band lost a pixel; labels / alive_out cover this
 * band's rows, *not_empty and *passes are the same on every rank. */
int ds_band_crossing_out(ds_handle *h, double epsilon, int max_passes, uint32_t *labels, uint8_t *alive_out,
                         int *not_empty, int *passes);

/* Per-launch timing of the sweep kernel with HIP events on the handle's stream.
 * ds_profile_begin clears the counters and enables timing (at most max_launches
 * launches are timed); ds_profile_end synchronises, disables timing and reports the
 * number of timed launches and their total / min / max duration in milliseconds. */
int ds_profile_begin(ds_handle *h, int max_launches);
int ds_profile_end(ds_handle *h, int *n_launches, double *total_ms, double *min_ms, double *max_ms);

/* After ds_profile_end: the timed launches of that window split by kind -- half-sweeps
 * (ds_sweep, and the one that completes ds_iterate) and whole-iteration launches of the fused
 * kernel (ds_iterate, ds_iterate_rows). */
int ds_profile_split(ds_handle *h, int *n_half_sweeps, double *half_sweep_ms, int *n_fused, double *fused_ms);

/* The same for one kind of launch: 0 = half-sweeps, 1 = one-iteration launches of the fused
 * kernel, 2 = two-iteration launches (ds_fused2_info). */
int ds_profile_kind(ds_handle *h, int kind, int *n_launches, double *ms);

/* Two iterations per launch.  On whole-image f32 handles whose D fills the lanes (D = 64, 128 or
 * 256) ds_iterate runs its iterations in pairs: one kernel computes both iterations on chip
 * for the part of the image where every pixel has four neighbours -- output columns
 * [*col_begin, *col_end), rows [*row_begin, *row_end) -- and the frame around it goes
 * through the one-iteration kernel twice, beside it on a second stream (a third set of
 * second-colour planes, allocated on first use, holds the frame's middle state; without the
 * memory for it, or with DS_FUSED2=off, every iteration is a launch of its own).  Same bits
 * either way.  *available: the handle has such a part; *steps: two-iteration steps run so far.
 * No counterpart in the reference.  Any pointer may be NULL. */
int ds_fused2_info(const ds_handle *h, int *available, int *col_begin, int *col_end, int *row_begin, int *row_end,
                   long long *steps);

/* Geometry of the handle (any pointer may be NULL). */
int ds_info(const ds_handle *h, int *H, int *W, int *D, int *dtype, int *row_begin, int *row_end);

/* Replaces DiffusionGraph::box_blur (diffusion_stereo/src/diffusion_graph.rs:399-424), the
 * post-filter of a disparity map (unused by the reference's main).  PARITY UNPINNED: the
 * reference has no test, fixture or caller for box_blur, so what is reproduced is a reading of
 * its source text, checked against values worked out by hand from that text, nothing more.
 * disparity_map: [H][W] on the host; out: [H][W] of 64-bit values (the reference's usize).  The
 * source subtracts usize values that can be smaller than what is subtracted; the arithmetic
 * chosen here is that of a release build of the crate, where such a subtraction wraps (the
 * crate's README recommends the release build): the window is rows [i - r, i + r) x
 * columns [j - r, j + r) and is empty for i < r or j < r (such pixels keep value * 255 /
 * normalization_term); a neighbour counts iff centre <= value <= centre + threshold; out =
 * (centre + sum of counted values) * 255 / ((1 + count) * normalization_term), integer division.
 * (A debug build of the crate panics on those subtractions instead.)  normalization_term = 0 is
 * DS_ERR_ARG (the crate divides by it).  Whole-image handles only. */
int ds_box_blur(ds_handle *h, const uint32_t *disparity_map, uint64_t threshold, uint64_t filter_radius,
                uint64_t normalization_term, uint64_t *out);

/* How phi was placed in HBM at creation.  By default phi is one plain allocation (*tries = 0,
 * *split = 0).  With DS_PLACEMENT=search in the environment, problems whose plane groups reach
 * 1 GiB get phi in two separately allocated plane groups and ds_create times the handle's own
 * kernel with the second group at a few distances behind the first (temporary spacer allocations,
 * never more than half of the free memory; DS_PLACEMENT_TRIES bounds the candidates), keeping the
 * best: MI355X's HBM answers in three 96 GiB regions and a sweep whose streams all fall into one of
 * them can run up to 15 % slower.  The gain varies from process to process (0 - 10 % measured) and
 * makes ds_create slow, so it is opt-in, for callers that create once and iterate for long.
 * *tries = placements timed, *probe_ms = the time of the one kept, *split = 1 if the groups are
 * separate allocations.  No counterpart in the reference. */
int ds_placement_info(const ds_handle *h, int *tries, double *probe_ms, int *split);

/* Order-independent 64-bit checksum of the band's phi (sum of the bit patterns of
 * all entries the reference would touch), to compare large states without copying. */
int ds_checksum(ds_handle *h, uint64_t *out);

/* ---- The superpixel variant: crate superpixel_stereo of the reference -------------------------
 *
 * The same diffusion on a coarser graph: the image is cut into windows of super_height x
 * super_width pixels and each window holds two nodes, its pixels at or below the window's mean
 * intensity and the others (superpixel_stereo/src/superpixels.rs:76-108).  A node has up to nine
 * neighbour slots (utils.rs:29-35: 0 = the other node of its window, 1-2 / 3-4 / 5-6 / 7-8 = the
 * two nodes of the left / upper / right / lower window), its data term is the sum of
 * |L[i][j] - R[i][j - d]| over its pixels (diffusion_graph.rs:134-150), its labels are d <= the
 * leftmost column of its pixels (:192-198), the pairwise term of an edge is
 * perimeter * smoothing * |d - d'| with a weight per edge (:164-173), and the ordering constraint
 * between horizontal neighbours is switched off (:199-205).  The crate is unfinished in the
 * reference (not in its CI, its tests do not compile); what is reproduced is its source text:
 * tests/test_sp_oracle_kat.py says, test by test, which of the crate's known answers that text
 * reproduces.  Arrays at the boundary use the crate's layouts: potentials
 * [windows down][windows across][2][9][D] (diffusion_graph.rs:44), edge_perimeters
 * [windows down][windows across][2][9] (superpixels.rs:47), superpixels [H][W].  Entries the crate
 * never touches read back as 0.  One handle = one GPU = the whole image (the graph of a
 * 400 x 400 image in 5 x 5 windows, the crate's own working point, has 12 800 nodes).
 * Same conventions as above (DS_OK / DS_ERR_*, ds_sp_last_error). */
typedef struct ds_sp_handle ds_sp_handle;

/* Replaces SuperpixelRepresentation::initialize + split_into_superpixels (superpixels.rs:58-108),
 * calculate_edge_perimeters (:147-266) when perimeters != 0, and DiffusionGraph::initialize
 * (diffusion_graph.rs:69-96).  perimeters == 0 leaves every weight 0, the state the crate's own
 * tests run in.  With perimeters != 0 the weights are the crate's: the boundary length inside
 * each window for slot 0, and 0 between windows -- its loops over a window's last column / row
 * test `image_j + 1 < window end` there, which never holds (:195-213, 234-255).  Where the crate
 * would index beyond the image in that function (:235 walks columns super_i * super_width ..,
 * i.e. images with (windows down - 1) * super_width > W) the call fails with DS_ERR_ARG as the
 * crate panics.  Windows that do not fit (H / super_height or W / super_width = 0): DS_ERR_ARG. */
int ds_sp_create(const uint8_t *left, const uint8_t *right, int H, int W, int D, double lambda, int super_height,
                 int super_width, int perimeters, int dtype, ds_sp_handle **out);
void ds_sp_destroy(ds_sp_handle *h);
const char *ds_sp_last_error(const ds_sp_handle *h);
int ds_sp_info(const ds_sp_handle *h, int *windows_down, int *windows_across, int *D, int *dtype);

/* The public fields superpixels (superpixels.rs:39; [H][W], 0 or 1, 0 outside every window) and
 * edge_perimeters (:47); left_j_in_superpixel (:110-123) capped at D - 1 for every node,
 * [windows down][windows across][2]; the data term per node and label,
 * [windows down][windows across][2][D] (0 for labels beyond left_j). */
int ds_sp_get_superpixels(ds_sp_handle *h, uint8_t *out);
int ds_sp_get_perimeters(ds_sp_handle *h, double *out);
int ds_sp_set_perimeters(ds_sp_handle *h, const double *in);
int ds_sp_get_left_j(ds_sp_handle *h, int32_t *out);
int ds_sp_get_cost(ds_sp_handle *h, uint32_t *out);

/* The public field potentials (diffusion_graph.rs:44), element type float or double. */
int ds_sp_get_phi(ds_sp_handle *h, void *out);
int ds_sp_set_phi(ds_sp_handle *h, const void *in);

/* Replaces n_iters calls of DiffusionGraph::diffusion_act (diffusion_graph.rs:357-482).
 * schedule 0: four launches per iteration, one per colour 2 * ((window row + window column) & 1)
 * + superpixel (nodes of one colour are never neighbours).  schedule 1: the crate's own order --
 * windows in raster order, superpixel 0 then 1 -- as one launch per anti-diagonal of windows and
 * superpixel (every node sees the operands the sequential loop gives it): the parity mode.
 * Synchronous at return. */
int ds_sp_iterate(ds_sp_handle *h, int n_iters, int schedule);

/* Replaces DiffusionGraph::energy (diffusion_graph.rs:640-671). */
int ds_sp_energy(ds_sp_handle *h, double *energy);

/* Replaces build_depth_map (diffusion_graph.rs:513-548) without the file: per pixel the argmin_d
 * f_t(d) of its node (min_penalty_vertex, :584-604, the first minimum); out [H][W]. */
int ds_sp_labels(ds_sp_handle *h, uint32_t *out);

/* Replaces CrossingOutGraph::initialize_with_epsilon + crossing_out + is_not_empty +
 * simple_best_labeling (superpixel_stereo/src/crossing_out_graph.rs:72-236,240-312,397-480) on
 * the current potentials.  Edge penalties are evaluated by brute force in f64 in the crate's
 * expression order, so vertices and edges fall on the side of the thresholds where the crate
 * puts them (on an f32 handle: for the f32 potentials converted to f64).  The crate keeps a copy
 * of every edge at each end, priced with that end's weight, and clears a node's copy on the death
 * of the neighbour's vertex only through the neighbour's copy (:161-236), so a copy whose twin
 * was never set outlives that vertex: reproduced, also for weights that differ between the two
 * ends (ds_sp_set_edge_perimeters; the crate's own are symmetric).  labels [H][W]
 * (0xffffffff at pixels of a node without labels), alive_out (optional) = the crate's `vertices`
 * [windows down][windows across][2][D]; all zero when *not_empty = 0. */
int ds_sp_crossing_out(ds_sp_handle *h, double epsilon, int max_passes, uint32_t *labels, uint8_t *alive_out,
                       int *not_empty, int *passes);

#ifdef __cplusplus
}
#endif
#endif /* DIFFSTEREO_H */
