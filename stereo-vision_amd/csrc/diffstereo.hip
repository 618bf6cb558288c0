// Host side of libdiffstereo: handle management, launches and the extern "C" ABI
// declared in include/diffstereo.h.  Device code lives in kernels.hpp.
#include "../../include/diffstereo.h"

#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <new>
#include <algorithm>
#include <functional>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "eps_sort.hpp"
#include "kernels.hpp"

namespace {

thread_local std::string g_create_error;

}  // namespace

constexpr int kReduceBlocks = 1024;  // first-stage blocks of the energy reduction

// Scratch of ds_energy_static: the energy reduction's partials in memory that exists from load
// time, so that the call allocates nothing.  Layout (doubles): [0, kStaticPartials) block sums,
// then kReduceBlocks first-stage sums, then the same two ranges for the maxima.  Every GPU the
// library runs on holds its own copy (32 MiB: room for a whole 3840 x 2160 x 256 image's
// 2 073 600 partial sums, no more) until the process ends.  In host code the array's
// name is only a handle for hipGetSymbolAddress: kernels and memsets get the per-device address
// kept in g_scratch_ptr.
constexpr long long kStaticPartials = 1ll << 21;
constexpr long long kStaticDoubles = 2 * (kStaticPartials + kReduceBlocks);
static __device__ __align__(256) double scratch[kStaticDoubles];

constexpr int kMaxDevices = 64;
static double *g_scratch_ptr[kMaxDevices];  // per device, set once by scratch_for_device()
static std::mutex g_scratch_mutex;

struct ds_handle {
    ds::Geometry g{};
    int dtype = DS_F32;
    double lambda = 0.0;
    int device = 0;
    int lpp = 64;         // lanes per pixel
    int seg_pixels = 64;  // same-colour pixels per block
    int grid_x = 1;
    long long plane_elems = 0;  // elements of one (colour, slot) plane

    uint8_t *d_left = nullptr, *d_right = nullptr;  // [H][W]
    uint8_t *cost_lm = nullptr;                     // [D][Hb][W] (+3 bytes padding)
    uint8_t *cost_px = nullptr;                     // [2][Hb][Wh][Dpad]
    // phi: eight planes [colour][slot] of (Hb+2) x Wh x Dpad elements in two groups: group A
    // = slots 0,1 of both colours, group B = slots 2,3 of both colours.  The groups may be
    // separate allocations (see place_phi).  `phi` = base pointer of group A.
    // Handles that can run whole iterations in one kernel (fused_waves > 0) hold a second set
    // of colour-1 planes (`alt_off`, two more planes per group): the fused kernel reads the
    // current colour-1 planes and writes the other set, then the two swap roles.
    void *phi = nullptr;
    void *phi_b = nullptr;        // group B when allocated apart, else NULL
    int fused_waves = 0;          // wavefronts per block of the fused kernel; 0 = not available
    bool fused_pending = false;   // fused launches since the first colour's planes were last current
    // two iterations per launch (see plan_fused2): the interior's output columns [cL, cR) and
    // rows [rT, rB), and a third set of colour-1 planes for the frame's first iteration
    struct Fused2Plan {
        bool ok = false;
        int cL = 0, cR = 0, strips = 0, rT = 0, rB = 0, strip1 = 0, waves = 16;
        // the frame's column regions, in strips of the one-iteration kernel: first iteration
        // (widened by two columns towards the interior) and second
        int left1 = 0, left2 = 0, right1_begin = 0, right1 = 0, right2 = 0;
        // row segments of the launches of the two-iteration kernel, chosen when a row range is first
        // launched (launch_fused2_rows): the whole interior, or its parts in a deep-halo step
        struct Segments {
            int row_begin = 0, row_end = 0, segs = 0;
            int seg_begin[17] = {};  // unequal segments when seg_count > 0
            int seg_count = 0;
        } cached[6];
        int cached_n = 0;
    } plan2;
    void *phi_y = nullptr;
    long long y_off[4] = {0, 0, 0, 0};
    long long fused2_steps = 0;   // two-iteration steps run so far (ds_fused2_info)
    int cu_count = 256;
    long long alt_off[4] = {0, 0, 0, 0};
    std::vector<int> prof_kind;   // per timed launch: 0 = half-sweep, 1 = fused iteration
    double placement_ms = 0.0;    // probe result of the chosen placement (0 = not probed)
    int placement_tries = 0;
    bool probing = false;         // launches belong to the placement probe
    size_t elem = 4;

    double *part_sum = nullptr, *part_max = nullptr;  // per-wavefront partials
    long long part_n = 0;
    double *d_scalars = nullptr;                      // [0] energy, [1] max residual
    unsigned long long *d_checksum = nullptr;
    uint32_t *d_labels = nullptr;                     // [Hb][W]
    uint8_t *co_alive = nullptr;                      // crossing-out vertex mask [2][Hb][Wh][Dpad]
    void *co_gmin = nullptr;                          // [2][Hb][Wh][4] min edge penalty per (pixel, slot)
    uint8_t *co_host_order = nullptr;                 // [Hb][W][D], only when the mask is asked for
    int *co_flags = nullptr;                          // [0] changed, [1] some pixel empty
    uint8_t *co_dirty = nullptr;                      // [2][Hb][grid_x]: blocks that deleted something, last pass / this pass
    int co_prev = 0;                                  // which half of co_dirty the next pass reads
    bool co_dirty_all = true;                         // masks changed outside the whole-band passes: every block is due
    int co_passes_done = 0;                           // whole-band relaxation passes since the masks were made
    int *co_version = nullptr;                        // [2][H][W], ds_crossing_out_exact: passes in which a pixel lost labels / what it saw last
    long long *co_state = nullptr;                    // co_fix_kernel's report, [2]
    double co_epsilon = 0.0;
    long long raster_pending = 0;                     // iterations asked for by ds_iterate_raster_async, not yet queued
    bool co_begun = false;
    bool co_fixed = false;                            // the last relaxation pass deleted nothing (or emptied the graph)
    double *eps_buf[2] = {nullptr, nullptr};          // epsilon gaps: values and the sort's second buffer
    double *eps_sorted = nullptr;                     // one of eps_buf: sorted, duplicates removed
    size_t eps_count = 0;

    // Deep halo (DS_BAND_HALO=deep, bands.hpp): the handle stores ext_up / ext_down rows of the
    // neighbouring bands beyond its own rows (3 next to a neighbour), so that g describes rows
    // [own rows - ext_up, own rows + ext_down) and every kernel that takes g works on them
    // unchanged; what a caller sees as "the band" is own_view().  Both 0 on every other handle.
    int ext_up = 0, ext_down = 0;
    bool deep = false;
    void *deep_stage = nullptr;  // [send | recv][up | down][4 slots][4 rows]: one transfer per neighbour and step

    hipStream_t own_stream = nullptr, stream = nullptr;
    // set around the last launch(es) of ds_iterate: the fused kernels then store the first colour
    // too, and no half-sweep has to bring it up to date afterwards
    bool store_first = false;

    // row bands over several ranks (bands.hpp): how halo rows travel, who the neighbours are,
    // and the stream / events that run a band's boundary rows beside its interior kernel
    ds_transport transport{};
    bool attached = false;
    int up = -1, down = -1;
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_main = nullptr, ev_side = nullptr;
    // a band's two-iteration step: the frame's launches that no exchange waits for
    hipStream_t frame_stream = nullptr;
    hipEvent_t ev_frame = nullptr, ev_rows = nullptr, ev_halo = nullptr;
    int *d_flag = nullptr;  // one int on the device for the bands' agreement on not_empty

    bool prof = false;
    int prof_cap = 0, prof_n = 0;
    std::vector<hipEvent_t> events;

    std::string err;
};

namespace {

int fail(ds_handle *h, int code, const std::string &msg)
{
    if (h) h->err = msg;
    else g_create_error = msg;
    return code;
}

int fail_hip(ds_handle *h, const char *what, hipError_t e)
{
    return fail(h, DS_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

#define HIPCHK(h, call)                                         \
    do {                                                        \
        hipError_t e_ = (call);                                 \
        if (e_ != hipSuccess) return fail_hip(h, #call, e_);    \
    } while (0)

template <typename F> int guarded(ds_handle *h, F &&body)
{
    try {
        return body();
    } catch (const std::bad_alloc &) {
        return fail(h, DS_ERR_NOMEM, "host allocation failed");
    } catch (const std::exception &e) {
        return fail(h, DS_ERR_STATE, std::string("internal error: ") + e.what());
    } catch (...) {
        return fail(h, DS_ERR_STATE, "internal error");
    }
}

// Calls f(T{}, std::integral_constant<int, LPP>{}) for the handle's dtype and lane split.
template <typename F> int dispatch(ds_handle *h, F &&f)
{
    auto with_t = [&](auto t) -> int {
        switch (h->lpp) {
        case 16: return f(t, std::integral_constant<int, 16>{});
        case 32: return f(t, std::integral_constant<int, 32>{});
        default: return f(t, std::integral_constant<int, 64>{});
        }
    };
    return h->dtype == DS_F64 ? with_t(double{}) : with_t(float{});
}

// Plane (colour, slot) is the (2 * colour + slot % 2)-th plane of group slot / 2.
void set_planes(ds_handle *h, void *group_a, void *group_b)
{
    ds::Geometry &g = h->g;
    const long long b_off = (static_cast<char *>(group_b) - static_cast<char *>(group_a)) / (long long)h->elem;
    for (int colour = 0; colour < 2; ++colour)
        for (int slot = 0; slot < 4; ++slot)
            g.plane_off[colour][slot] = (slot / 2 ? b_off : 0) + (2 * colour + slot % 2) * h->plane_elems;
    for (int slot = 0; slot < 4; ++slot)
        h->alt_off[slot] = (slot / 2 ? b_off : 0) + (4 + slot % 2) * h->plane_elems;
}

int planes_per_group(const ds_handle *h) { return h->fused_waves ? 6 : 4; }

int use_device(ds_handle *h)
{
    HIPCHK(h, hipSetDevice(h->device));
    return DS_OK;
}

// The band's own rows as a geometry of their own: the same planes, entered ext_up rows further
// down.  Equal to h->g unless the handle stores rows of its neighbours (deep halo).
ds::Geometry own_view(const ds_handle *h)
{
    ds::Geometry v = h->g;
    v.row0 = h->g.row0 + h->ext_up;
    v.Hb = h->g.Hb - h->ext_up - h->ext_down;
    for (int c = 0; c < 2; ++c)
        for (int s = 0; s < 4; ++s) v.plane_off[c][s] += (long long)h->ext_up * h->g.row_stride;
    return v;
}
// the packed cost of the own rows (the colour stride stays the stored band's)
const uint8_t *own_cost(const ds_handle *h) { return h->cost_px + (size_t)h->ext_up * h->g.Wh * h->g.Dpad; }
int own_begin(const ds_handle *h) { return h->g.row0 + h->ext_up; }
int own_end(const ds_handle *h) { return h->g.row0 + h->g.Hb - h->ext_down; }

int not_deep(ds_handle *h, const char *who);

// Address of the current device's copy of `scratch`, looked up once per device.
int scratch_for_device(ds_handle *h, int device, double **out)
{
    if (device < 0 || device >= kMaxDevices) return fail(h, DS_ERR_ARG, "device index beyond the library's table");
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void *p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) return fail_hip(h, "hipGetSymbolAddress(scratch)", e);
        g_scratch_ptr[device] = static_cast<double *>(p);
    }
    if (out) *out = g_scratch_ptr[device];
    return DS_OK;
}

// The entry points with which a host sequences a band's half-sweeps and halo rows itself work on
// one halo row; a deep-halo handle runs its protocol inside ds_band_iterate only.
int not_deep(ds_handle *h, const char *who)
{
    if (h->deep)
        return fail(h, DS_ERR_STATE, std::string(who) + ": this band was created with DS_BAND_HALO=deep; "
                                     "use ds_band_iterate / ds_band_exchange");
    return DS_OK;
}

int flush_raster(ds_handle *h);

// The readers of phi need both colours' planes current.
int need_current(ds_handle *h, const char *who)
{
    if (h->raster_pending) {
        const int rc = flush_raster(h);
        if (rc != DS_OK) return rc;
    }
    if (h->fused_pending)
        return fail(h, DS_ERR_STATE, std::string(who) + ": the first colour's planes are out of date after "
                                     "ds_iterate_rows; call ds_finish_iterations first");
    return DS_OK;
}

void release(ds_handle *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    for (hipEvent_t e : h->events) (void)hipEventDestroy(e);
    (void)hipFree(h->d_left);
    (void)hipFree(h->d_right);
    (void)hipFree(h->cost_lm);
    (void)hipFree(h->cost_px);
    (void)hipFree(h->phi);
    (void)hipFree(h->phi_b);
    (void)hipFree(h->phi_y);
    (void)hipFree(h->deep_stage);
    (void)hipFree(h->part_sum);
    (void)hipFree(h->part_max);
    (void)hipFree(h->d_scalars);
    (void)hipFree(h->d_checksum);
    (void)hipFree(h->d_labels);
    (void)hipFree(h->co_alive);
    (void)hipFree(h->co_gmin);
    (void)hipFree(h->co_host_order);
    (void)hipFree(h->co_flags);
    (void)hipFree(h->co_dirty);
    (void)hipFree(h->co_version);
    (void)hipFree(h->co_state);
    (void)hipFree(h->eps_buf[0]);
    (void)hipFree(h->eps_buf[1]);
    if (h->side_stream) {
        (void)hipStreamSynchronize(h->side_stream);
        (void)hipStreamDestroy(h->side_stream);
    }
    if (h->frame_stream) {
        (void)hipStreamSynchronize(h->frame_stream);
        (void)hipStreamDestroy(h->frame_stream);
    }
    if (h->ev_frame) (void)hipEventDestroy(h->ev_frame);
    if (h->ev_rows) (void)hipEventDestroy(h->ev_rows);
    if (h->ev_halo) (void)hipEventDestroy(h->ev_halo);
    if (h->ev_main) (void)hipEventDestroy(h->ev_main);
    if (h->ev_side) (void)hipEventDestroy(h->ev_side);
    (void)hipFree(h->d_flag);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
}

// Half-sweep of `colour` over image rows [row_begin, row_end) and, in the same launch,
// [row_begin2, row_end2) (may be empty).
int launch_sweep(ds_handle *h, int colour, int row_begin, int row_end, const ds::Geometry *planes = nullptr,
                 int row_begin2 = 0, int row_end2 = 0)
{
    const int split = row_end - row_begin > 0 ? row_end - row_begin : 0;
    const int rows = split + (row_end2 - row_begin2 > 0 ? row_end2 - row_begin2 : 0);
    if (rows <= 0) return DS_OK;
    const ds::Geometry &geo = planes ? *planes : h->g;
    const bool timed = h->prof && h->prof_n < h->prof_cap;
    if (timed) {
        HIPCHK(h, hipEventRecord(h->events[2 * h->prof_n], h->stream));
        h->prof_kind[h->prof_n] = 0;
    }
    const dim3 grid(h->grid_x, rows), block(ds::kBlock);
    int rc = dispatch(h, [&](auto t, auto lpp) -> int {
        using T = decltype(t);
        constexpr int L = decltype(lpp)::value;
        if (h->probing)
            hipLaunchKernelGGL((ds::sweep_kernel<T, L, true>), grid, block, 0, h->stream, geo,
                               static_cast<T *>(h->phi), h->cost_px, (T)h->lambda, colour, row_begin, h->seg_pixels,
                               split, row_begin2);
        else
            hipLaunchKernelGGL((ds::sweep_kernel<T, L, false>), grid, block, 0, h->stream, geo,
                               static_cast<T *>(h->phi), h->cost_px, (T)h->lambda, colour, row_begin, h->seg_pixels,
                               split, row_begin2);
        return DS_OK;
    });
    if (rc != DS_OK) return rc;
    HIPCHK(h, hipGetLastError());
    if (timed) {
        HIPCHK(h, hipEventRecord(h->events[2 * h->prof_n + 1], h->stream));
        ++h->prof_n;
    }
    return DS_OK;
}

// One whole iteration over the colour-1 rows [row_begin, row_end) in one kernel: reads the
// current colour-1 planes, writes the alternate set, and makes that set current.  The colour-0
// planes are neither read nor written (see materialise_first_colour).
// Rows of the second colour the fused kernel may update on this handle: it recomputes the
// first colour's rows i-1, i, i+1, which read the second colour's rows i-2 .. i+2, and a band
// holds only one halo row beyond its own.  So next to a neighbouring band the band's first /
// last row is left to the half-sweeps; at the image's top / bottom there is nothing beyond.
void fused_safe_rows(const ds_handle *h, int *begin, int *end)
{
    const ds::Geometry &g = h->g;
    *begin = g.row0 == 0 ? 0 : g.row0 + 1;
    *end = g.row0 + g.Hb == g.H ? g.H : g.row0 + g.Hb - 1;
}

// Row segments of a one-iteration launch.  One or two blocks fill a CU's LDS, so the grid runs in
// rounds; many short blocks even out the slower border strips and whatever CUs a concurrent
// transfer kernel holds, and a segment only costs the two extra red rows it recomputes.  Measured
// with 16-wavefront blocks, 4K x 256 f32, ms per iteration for 4 / 8 / 16 / 32 / 48 segments of the
// 2160 rows: 8.25 / 7.85 / 7.17 / 6.77 / 6.76; a 270-row band next to RCCL traffic: 1.12 ms with
// 2..8 segments, 1.07 with 12.  So: about 64 rows per segment, but at least 6 blocks per CU.
int fused_rows_per_segment(const ds_handle *h, int strips, int rows)
{
    int segs = (rows + 63) / 64;
    const int want = (6 * h->cu_count + strips - 1) / strips;
    if (segs < want) segs = want;
    if (segs > (rows + 15) / 16) segs = (rows + 15) / 16;
    if (segs < 1) segs = 1;
    if (const char *e = std::getenv("DS_FUSED_SEGMENTS")) segs = std::atoi(e) > 0 ? std::atoi(e) : segs;  // tuning knob
    if (segs > rows) segs = rows;
    return (rows + segs - 1) / segs;
}

// The one-iteration kernel over the colour-1 rows [row_begin, row_end) and `strips` column
// strips from column col_begin (even) on: reads the colour-1 planes `in_off`, writes `out_off`
// (element offsets from the phi base).  kind: how the launch is booked by ds_profile_*.
int launch_fused_region(ds_handle *h, const long long in_off[4], const long long out_off[4], int col_begin, int strips,
                        int row_begin, int row_end, int kind)
{
    const int rows = row_end - row_begin;
    if (rows <= 0 || strips <= 0) return DS_OK;
    ds::Geometry geo = h->g;
    for (int s = 0; s < 4; ++s) geo.plane_off[1][s] = in_off[s];
    const bool timed = h->prof && h->prof_n < h->prof_cap;
    if (timed) {
        HIPCHK(h, hipEventRecord(h->events[2 * h->prof_n], h->stream));
        h->prof_kind[h->prof_n] = kind;
    }
    const int fw = h->fused_waves;
    const int rows_per_seg = fused_rows_per_segment(h, strips, rows);
    const dim3 grid(strips, (rows + rows_per_seg - 1) / rows_per_seg), block(64 * fw);
    ds::FusedPlanes np;
    for (int s = 0; s < 4; ++s) np.out_off[s] = out_off[s];
    int rc = dispatch(h, [&](auto t, auto lpp) -> int {
        using T = decltype(t);
        constexpr int L = decltype(lpp)::value;
        auto go = [&](auto fwc) {
            constexpr int FW = decltype(fwc)::value;
            if (h->probing)
                hipLaunchKernelGGL((ds::fused_iteration_kernel<T, L, FW, true>), grid, block, 0, h->stream, geo,
                                   static_cast<T *>(h->phi), np, h->cost_px, (T)h->lambda, row_begin, row_end,
                                   rows_per_seg, col_begin, h->store_first ? 1 : 0);
            else
                hipLaunchKernelGGL((ds::fused_iteration_kernel<T, L, FW, false>), grid, block, 0, h->stream, geo,
                                   static_cast<T *>(h->phi), np, h->cost_px, (T)h->lambda, row_begin, row_end,
                                   rows_per_seg, col_begin, h->store_first ? 1 : 0);
        };
        if (fw == 16) go(std::integral_constant<int, 16>{});
        else if (fw == 8) go(std::integral_constant<int, 8>{});
        else go(std::integral_constant<int, 4>{});
        return DS_OK;
    });
    if (rc != DS_OK) return rc;
    HIPCHK(h, hipGetLastError());
    if (timed) {
        HIPCHK(h, hipEventRecord(h->events[2 * h->prof_n + 1], h->stream));
        ++h->prof_n;
    }
    return DS_OK;
}

// Several regions of the same planes in one launch (empty ones are left out).
struct FrameRegion {
    int col_begin, strips, row_begin, row_end;
    int col_end = 0x7fffffff;  // columns from here on are another launch's
};

int launch_fused_regions(ds_handle *h, const long long in_off[4], const long long out_off[4], const FrameRegion *regions,
                         int count, int kind)
{
    ds::FusedRegions reg{};
    int blocks = 0;
    for (int k = 0; k < count && reg.n < 4; ++k) {
        const FrameRegion &r = regions[k];
        const int rows = r.row_end - r.row_begin;
        if (rows <= 0 || r.strips <= 0) continue;
        const int per = fused_rows_per_segment(h, r.strips, rows);
        const int n = reg.n++;
        reg.first[n] = blocks;
        reg.col_begin[n] = r.col_begin;
        reg.col_end[n] = r.col_end;
        reg.strips[n] = r.strips;
        reg.row_begin[n] = r.row_begin;
        reg.row_end[n] = r.row_end;
        reg.rows_per_seg[n] = per;
        blocks += r.strips * ((rows + per - 1) / per);
    }
    if (reg.n == 0) return DS_OK;
    reg.first[reg.n] = blocks;
    ds::Geometry geo = h->g;
    for (int s = 0; s < 4; ++s) geo.plane_off[1][s] = in_off[s];
    const bool timed = h->prof && h->prof_n < h->prof_cap;
    if (timed) {
        HIPCHK(h, hipEventRecord(h->events[2 * h->prof_n], h->stream));
        h->prof_kind[h->prof_n] = kind;
    }
    const int fw = h->fused_waves;
    const dim3 grid(blocks), block(64 * fw);
    ds::FusedPlanes np;
    for (int s = 0; s < 4; ++s) np.out_off[s] = out_off[s];
    int rc = dispatch(h, [&](auto t, auto lpp) -> int {
        using T = decltype(t);
        constexpr int L = decltype(lpp)::value;
        auto go = [&](auto fwc) {
            constexpr int FW = decltype(fwc)::value;
            hipLaunchKernelGGL((ds::fused_regions_kernel<T, L, FW>), grid, block, 0, h->stream, geo,
                               static_cast<T *>(h->phi), np, h->cost_px, (T)h->lambda, reg, h->store_first ? 1 : 0);
        };
        if (fw == 16) go(std::integral_constant<int, 16>{});
        else if (fw == 8) go(std::integral_constant<int, 8>{});
        else go(std::integral_constant<int, 4>{});
        return DS_OK;
    });
    if (rc != DS_OK) return rc;
    HIPCHK(h, hipGetLastError());
    if (timed) {
        HIPCHK(h, hipEventRecord(h->events[2 * h->prof_n + 1], h->stream));
        ++h->prof_n;
    }
    return DS_OK;
}

// One whole iteration over the handle's fused rows, all columns: current colour-1 planes ->
// the alternate set, which becomes current.
int launch_fused(ds_handle *h, int row_begin, int row_end)
{
    if (row_end - row_begin <= 0) return DS_OK;
    ds::Geometry &g = h->g;
    int safe_begin = 0, safe_end = 0;
    fused_safe_rows(h, &safe_begin, &safe_end);
    // Exactly the safe rows: the launch makes the other set of colour-1 planes current, so rows
    // it leaves out would silently fall back to what that set held two iterations ago.
    if (!h->fused_waves || row_begin != safe_begin || row_end != safe_end)
        return fail(h, DS_ERR_ARG, "a fused iteration must cover exactly the rows ds_fused_rows reports");
    const int strip_cols = 2 * (h->fused_waves * (64 / h->lpp) - 1);
    int rc = launch_fused_region(h, g.plane_off[1], h->alt_off, 0, (g.W + strip_cols - 1) / strip_cols, row_begin,
                                 row_end, 1);
    if (rc != DS_OK) return rc;
    for (int s = 0; s < 4; ++s) std::swap(g.plane_off[1][s], h->alt_off[s]);
    h->fused_pending = !h->store_first;
    return DS_OK;
}

// Work queued on `from` so far happens before work queued on `to` from now on.
int order_streams(ds_handle *h, hipStream_t from, hipEvent_t ev, hipStream_t to)
{
    HIPCHK(h, hipEventRecord(ev, from));
    HIPCHK(h, hipStreamWaitEvent(to, ev, 0));
    return DS_OK;
}

// Launches of the library go to h->stream: run `body` with another stream in its place.
template <typename F> int on_stream(ds_handle *h, hipStream_t stream, F &&body)
{
    hipStream_t keep = h->stream;
    h->stream = stream;
    const int rc = body();
    h->stream = keep;
    return rc;
}
template <typename F> int on_side_stream(ds_handle *h, F &&body) { return on_stream(h, h->side_stream, body); }

// The second stream of a handle (a band's boundary rows, the borders of a two-iteration step)
// and the two events that order it against the handle's stream.
int need_side_stream(ds_handle *h)
{
    if (!h->side_stream) HIPCHK(h, hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking));
    if (!h->ev_main) HIPCHK(h, hipEventCreateWithFlags(&h->ev_main, hipEventDisableTiming));
    if (!h->ev_side) HIPCHK(h, hipEventCreateWithFlags(&h->ev_side, hipEventDisableTiming));
    return DS_OK;
}

int need_frame_stream(ds_handle *h)
{
    if (!h->frame_stream) HIPCHK(h, hipStreamCreateWithFlags(&h->frame_stream, hipStreamNonBlocking));
    if (!h->ev_frame) HIPCHK(h, hipEventCreateWithFlags(&h->ev_frame, hipEventDisableTiming));
    if (!h->ev_rows) HIPCHK(h, hipEventCreateWithFlags(&h->ev_rows, hipEventDisableTiming));
    if (!h->ev_halo) HIPCHK(h, hipEventCreateWithFlags(&h->ev_halo, hipEventDisableTiming));
    return DS_OK;
}

// ---- two iterations per launch -------------------------------------------------------------
// fused2_iteration_kernel takes the part of the image where nothing needs masking: output
// columns [cL, cR) x rows [rT, rB).  The frame around it goes through the one-iteration kernel
// twice, via a third set of colour-1 planes Y (allocated on first use): X -> Y on the frame
// widened by two pixels (the second iteration of a frame pixel reads first-iteration values up
// to two pixels away), then Y -> Z on the frame itself.  Z is also where the two-iteration
// kernel writes, so after the step Z holds iteration n + 2 everywhere and becomes current.
// The frame's eight launches run on the side stream beside the interior kernel: both read X
// only (and Y, which nothing else touches) and write disjoint parts of Z.  On a row band the
// rows next to a neighbouring band also need that band's rows after each of the two iterations:
// band_fused2_step (bands.hpp) runs the band protocol on them, twice.
void plan_fused2(ds_handle *h)
{
    ds_handle::Fused2Plan &p = h->plan2;
    p = ds_handle::Fused2Plan{};
    const ds::Geometry &g = h->g;
    const char *e = std::getenv("DS_FUSED2");
    if (e && std::strcmp(e, "off") == 0) return;
    // D fills the lanes of a pixel slot (64 / 128 / 256), or only the lower ones (any other D: the
    // kernel's PARTIAL instantiation)
    // f64: 8-wavefront blocks (the registers of a 12-wavefront block do not hold f64 rows); 4K x 256:
    // 10.4 ms per iteration against 13.7 for one iteration per launch on the same device.
    // DS_FUSED2_F64=off keeps f64 to one iteration per launch.
    const char *f64 = std::getenv("DS_FUSED2_F64");
    const bool f64_probe = h->dtype == DS_F64 && !(f64 && std::strcmp(f64, "off") == 0) && h->fused_waves == 8;
    if (!f64_probe && (h->dtype != DS_F32 || h->fused_waves != 16)) return;
    if (const char *pd = std::getenv("DS_FUSED2_PARTIAL"))  // checking knob: off = only D = 4 LPP
        if (std::strcmp(pd, "off") == 0 && g.D != 4 * h->lpp) return;
    p.waves = f64_probe ? 8 : 12;
    if (const char *w = std::getenv("DS_F2_WAVES"))  // tuning knob
        if (!f64_probe) p.waves = std::atoi(w) == 16 ? 16 : 12;
    const int FC1 = 2 * (h->fused_waves * (64 / h->lpp) - 1);     // the one-iteration kernel's strip
    const int NP = p.waves * (64 / h->lpp), FC2 = 2 * (NP - 3);   // the two-iteration kernel's slots and strip
    // Left edge: two columns short of a multiple of the one-iteration strip, so that the frame's
    // first iteration, which must reach two columns into the interior, still fits whole strips.
    // One strip is enough for every pixel the kernel touches to have a left neighbour (c0 - 2 =
    // cL - 4 >= 0); DS_FUSED2_LEFT=full keeps the kernel to the columns where every label exists
    // (c0 - 1 = cL - 3 >= D), for checking.
    p.cL = FC1 - 2;
    if (const char *l = std::getenv("DS_FUSED2_LEFT")) {
        if (std::strcmp(l, "full") == 0) p.cL = ((g.D + 3 + FC1 - 1) / FC1) * FC1 - 2;
        else if (std::atoi(l) > 0) p.cL = ((std::atoi(l) + FC1 - 1) / FC1) * FC1 - 2;  // tuning knob: columns left to the one-iteration kernel
    }
    const long long room = (long long)g.W - 1 - 2 * NP - (p.cL - 2);  // c0 of the last strip + 2 NP < W
    p.strips = room >= 0 ? (int)(room / FC2) + 1 : 0;
    p.cR = p.cL + p.strips * FC2;
    // Rows: every pixel any stage touches needs four neighbours, and the first stage reads the
    // second colour four rows beyond the output rows.  At the image's top / bottom that leaves out
    // four rows; next to a neighbouring band three (the band holds one halo row beyond its own).
    p.rT = g.row0 == 0 ? 4 : g.row0 + 3;
    p.rB = g.row0 + g.Hb == g.H ? g.H - 4 : g.row0 + g.Hb - 3;
    p.strip1 = FC1;
    p.left1 = (p.cL + 2 + FC1 - 1) / FC1;
    p.left2 = (p.cL + FC1 - 1) / FC1;  // its last strip is cut off at cL (FrameRegion::col_end)
    p.right1_begin = p.cR - 2;
    p.right1 = (g.W - p.right1_begin + FC1 - 1) / FC1;
    p.right2 = (g.W - p.cR + FC1 - 1) / FC1;
    p.ok = p.strips >= 1 && p.rB - p.rT >= 8;
}

int launch_fused2_rows(ds_handle *h, int row_begin, int row_end);
int launch_fused2_interior(ds_handle *h) { return launch_fused2_rows(h, h->plan2.rT, h->plan2.rB); }

int fused2_step(ds_handle *h)
{
    ds::Geometry &g = h->g;
    const ds_handle::Fused2Plan &p = h->plan2;
    const long long *X = g.plane_off[1], *Y = h->y_off, *Z = h->alt_off;
    const int all = (g.W + p.strip1 - 1) / p.strip1;  // one-iteration strips over the whole width
    int rc = order_streams(h, h->stream, h->ev_main, h->side_stream);
    if (rc != DS_OK) return rc;
    rc = on_side_stream(h, [&]() -> int {
        // first iteration on the widened frame, X -> Y; second iteration on the frame, Y -> Z: one
        // launch each (a stream of eight small dependent launches beside a kernel that fills the
        // device gets compute units only as that kernel's blocks end, launch after launch)
        const FrameRegion first[4] = {{0, all, 0, p.rT + 2},
                                      {0, all, p.rB - 2, g.H},
                                      {0, p.left1, p.rT + 2, p.rB - 2},
                                      {p.right1_begin, p.right1, p.rT + 2, p.rB - 2}};
        const FrameRegion second[4] = {{0, all, 0, p.rT},
                                       {0, all, p.rB, g.H},
                                       {0, p.left2, p.rT, p.rB, p.cL},
                                       {p.cR, p.right2, p.rT, p.rB}};
        // the first pass makes iteration n+1 (also on pixels the interior kernel finishes): it never
        // stores the first colour, the second pass does when this is the call's last step
        const bool last = h->store_first;
        h->store_first = false;
        int r = launch_fused_regions(h, X, Y, first, 4, 1);
        h->store_first = last;
        if (r == DS_OK) r = launch_fused_regions(h, Y, Z, second, 4, 1);
        return r;
    });
    if (rc != DS_OK) return rc;
    rc = launch_fused2_interior(h);
    if (rc == DS_OK) rc = order_streams(h, h->side_stream, h->ev_side, h->stream);
    if (rc != DS_OK) return rc;
    for (int s = 0; s < 4; ++s) std::swap(g.plane_off[1][s], h->alt_off[s]);
    h->fused_pending = !h->store_first;
    return DS_OK;
}

// The two-iteration kernel over rows [row_begin, row_end) of the handle's interior (all of its
// columns): current second-colour planes -> the alternate set, on the handle's stream.
int launch_fused2_rows(ds_handle *h, int row_begin, int row_end)
{
    ds::Geometry &g = h->g;
    ds_handle::Fused2Plan &p = h->plan2;
    const long long *Z = h->alt_off;
    const int rows = row_end - row_begin;
    if (rows <= 0) return DS_OK;
    ds_handle::Fused2Plan::Segments *cache = nullptr;
    for (int k = 0; k < p.cached_n; ++k)
        if (p.cached[k].row_begin == row_begin && p.cached[k].row_end == row_end) cache = &p.cached[k];
    // Row segments: about 96 rows each (a segment recomputes five rows of the first stages).  One
    // block fills a CU (157 registers x 12 wavefronts), so the grid runs in rounds of cu_count
    // blocks, and when that leaves only a few rounds (a band, a small image) equal segments waste
    // much of the last one: 264 rows x 211 strips in 6 segments of 44 rows are 4.95 rounds of 49
    // row steps = 245, against 230 if the work could be cut freely.  Then the segments get unequal
    // lengths, longest first (blocks are dispatched segment by segment, so this is longest-
    // processing-time-first list scheduling): k segments with lengths falling linearly to a
    // fraction f of the first; the (k, f) with the smallest simulated makespan wins (135 + 88 + 41
    // rows: 233).  Blocks that end at different times also hand compute units to the boundary
    // stream's small launches more often than once per round.
    int segs = (rows + 95) / 96;
    ds::RowSegments table{};
    if (cache) {
        segs = cache->segs;
        table.n = cache->seg_count;
        for (int i = 0; i <= cache->seg_count; ++i) table.begin[i] = cache->seg_begin[i];
    } else if ((long long)p.strips * segs < 4ll * h->cu_count && !std::getenv("DS_FUSED2_ROWS")) {
        long long best = -1;
        std::vector<long long> cu((size_t)h->cu_count);
        for (int k = 1; k <= 16 && k * 8 <= rows; ++k) {
            for (int f10 = 10; f10 >= 1; --f10) {
                if (k == 1 && f10 < 10) break;
                // lengths proportional to 1 - (1 - f) i / (k - 1)
                int len[16], sum = 0;
                double total = 0.;
                for (int i = 0; i < k; ++i) total += 1. - (1. - f10 / 10.) * (k > 1 ? (double)i / (k - 1) : 0.);
                bool usable = true;
                for (int i = 0; i < k; ++i) {
                    const double w = 1. - (1. - f10 / 10.) * (k > 1 ? (double)i / (k - 1) : 0.);
                    len[i] = (int)(rows * w / total + 0.5);
                    sum += len[i];
                }
                len[0] += rows - sum;
                for (int i = 0; i < k; ++i) usable = usable && len[i] >= 8;
                if (!usable) continue;
                // list scheduling in dispatch order, a min-heap over the compute units
                std::fill(cu.begin(), cu.end(), 0ll);
                for (int i = 0; i < k; ++i)
                    for (int sidx = 0; sidx < p.strips; ++sidx) {
                        std::pop_heap(cu.begin(), cu.end(), std::greater<long long>());
                        cu.back() += len[i] + 5;
                        std::push_heap(cu.begin(), cu.end(), std::greater<long long>());
                    }
                const long long span = *std::max_element(cu.begin(), cu.end());
                if (best < 0 || span < best) {
                    best = span;
                    table.n = k;
                    table.begin[0] = row_begin;
                    for (int i = 0; i < k; ++i) table.begin[i + 1] = table.begin[i] + len[i];
                }
            }
        }
        if (table.n > 0) segs = table.n;
    }
    if (!cache && p.cached_n < 6) {  // the search above costs milliseconds: once per handle and row range
        cache = &p.cached[p.cached_n++];
        cache->row_begin = row_begin;
        cache->row_end = row_end;
        cache->segs = segs;
        cache->seg_count = table.n;
        for (int i = 0; i <= table.n; ++i) cache->seg_begin[i] = table.begin[i];
    }
    if (const char *e = std::getenv("DS_FUSED2_ROWS"))  // tuning knob: rows per segment
        if (std::atoi(e) > 0) segs = (rows + std::atoi(e) - 1) / std::atoi(e);
    int rows_per_seg = (rows + segs - 1) / segs;
    if (table.n == 0) segs = (rows + rows_per_seg - 1) / rows_per_seg;
    const bool timed = h->prof && h->prof_n < h->prof_cap;
    if (timed) {
        HIPCHK(h, hipEventRecord(h->events[2 * h->prof_n], h->stream));
        h->prof_kind[h->prof_n] = 2;
    }
    ds::Geometry in = g;
    ds::FusedPlanes np;
    for (int s = 0; s < 4; ++s) np.out_off[s] = Z[s];
    const dim3 grid(p.strips, segs), block(64 * p.waves);
    float *base = static_cast<float *>(h->phi);
    const float lambda = (float)h->lambda;
    // 0: D fills the slot's lanes; 1: the lower lanes, whole ones; 2: the last of them partly
    const int partial = g.D == 4 * h->lpp ? 0 : g.D % 4 == 0 ? 1 : 2;
    auto go = [&](auto lppc, auto wavesc) {
        constexpr int L = decltype(lppc)::value, FW = decltype(wavesc)::value;
        auto launch = [&](auto mode) {
            hipLaunchKernelGGL((ds::fused2_iteration_kernel<float, L, FW, decltype(mode)::value>), grid, block, 0, h->stream,
                               in, base, base, np, h->cost_px, lambda, p.cL, row_begin, row_end, rows_per_seg, table,
                               h->store_first ? 1 : 0);
        };
        if (partial == 0) launch(std::integral_constant<int, 0>{});
        else if (partial == 1) launch(std::integral_constant<int, 1>{});
        else launch(std::integral_constant<int, 2>{});
    };
    auto with_waves = [&](auto lppc) {
        if (p.waves == 12) go(lppc, std::integral_constant<int, 12>{});
        else go(lppc, std::integral_constant<int, 16>{});
    };
    if (h->dtype == DS_F64) {  // the probe: 8 wavefronts
        double *b64 = static_cast<double *>(h->phi);
        auto go64 = [&](auto lppc) {
            auto launch = [&](auto mode) {
                hipLaunchKernelGGL((ds::fused2_iteration_kernel<double, decltype(lppc)::value, 8, decltype(mode)::value>), grid,
                                   block, 0, h->stream, in, b64, b64, np, h->cost_px, h->lambda, p.cL, row_begin, row_end,
                                   rows_per_seg, table, h->store_first ? 1 : 0);
            };
            if (partial == 0) launch(std::integral_constant<int, 0>{});
            else if (partial == 1) launch(std::integral_constant<int, 1>{});
            else launch(std::integral_constant<int, 2>{});
        };
        switch (h->lpp) {
        case 16: go64(std::integral_constant<int, 16>{}); break;
        case 32: go64(std::integral_constant<int, 32>{}); break;
        default: go64(std::integral_constant<int, 64>{}); break;
        }
    } else
    switch (h->lpp) {
    case 16: with_waves(std::integral_constant<int, 16>{}); break;
    case 32: with_waves(std::integral_constant<int, 32>{}); break;
    default: with_waves(std::integral_constant<int, 64>{}); break;
    }
    HIPCHK(h, hipGetLastError());
    if (timed) {
        HIPCHK(h, hipEventRecord(h->events[2 * h->prof_n + 1], h->stream));
        ++h->prof_n;
    }
    return DS_OK;
}

// Y planes and the side stream, on first use.  Returns false (and switches the plan off) if the
// memory is not there: the caller then iterates one iteration per launch.
bool prepare_fused2(ds_handle *h)
{
    if (!h->plan2.ok) return false;
    if (h->phi_y) return true;
    if (need_side_stream(h) != DS_OK) {
        h->plan2.ok = false;
        return false;
    }
    const size_t bytes = (size_t)(4 * h->plane_elems) * h->elem;
    if (hipMalloc(&h->phi_y, bytes) != hipSuccess) {
        (void)hipGetLastError();
        h->phi_y = nullptr;
        h->plan2.ok = false;
        return false;
    }
    const long long off = (static_cast<char *>(h->phi_y) - static_cast<char *>(h->phi)) / (long long)h->elem;
    for (int s = 0; s < 4; ++s) h->y_off[s] = off + s * h->plane_elems;
    // same +inf / 0 pattern as every other plane (the entries no kernel writes must hold +inf)
    ds::Geometry y = h->g;
    for (int s = 0; s < 4; ++s) y.plane_off[1][s] = h->y_off[s];
    long long blocks = (8 * h->plane_elems + ds::kBlock - 1) / ds::kBlock;
    if (blocks > 65536) blocks = 65536;
    if (h->dtype == DS_F64)
        hipLaunchKernelGGL(ds::init_phi_kernel<double>, dim3((unsigned)blocks), dim3(ds::kBlock), 0, h->stream, y,
                           static_cast<double *>(h->phi));
    else
        hipLaunchKernelGGL(ds::init_phi_kernel<float>, dim3((unsigned)blocks), dim3(ds::kBlock), 0, h->stream, y,
                           static_cast<float *>(h->phi));
    if (hipGetLastError() != hipSuccess) {
        h->plan2.ok = false;
        return false;
    }
    return true;
}

// After fused iterations the colour-0 planes are out of date: phi of colour 0 after iteration
// n is F(phi of colour 1 after iteration n-1), and that input is what the alternate planes
// hold right after a fused launch.
int materialise_first_colour(ds_handle *h, int row_begin, int row_end)
{
    if (!h->fused_pending) return DS_OK;  // nothing to bring up to date (and the alternate planes hold no input)
    ds::Geometry before = h->g;
    for (int s = 0; s < 4; ++s) before.plane_off[1][s] = h->alt_off[s];
    int rc = launch_sweep(h, 0, row_begin, row_end, &before);
    if (rc == DS_OK) h->fused_pending = false;
    return rc;
}

int build_cost(ds_handle *h)
{
    const ds::Geometry &g = h->g;
    const long long total = (long long)g.D * g.Hb * g.W;
    const long long threads = (total + 3) / 4;
    const unsigned blocks = (unsigned)((threads + ds::kBlock - 1) / ds::kBlock);
    hipLaunchKernelGGL(ds::cost_volume_kernel, dim3(blocks), dim3(ds::kBlock), 0, h->stream, h->d_left,
                       h->d_right, g.W, g.D, g.row0, g.Hb, h->cost_lm);
    HIPCHK(h, hipGetLastError());
    const dim3 grid((g.W + ds::kPackCols - 1) / ds::kPackCols, g.Hb);
    const size_t lds = (size_t)ds::kPackCols * (g.Dpad + 4);
    hipLaunchKernelGGL(ds::pack_cost_kernel, grid, dim3(ds::kBlock), lds, h->stream, g, h->cost_lm,
                       h->cost_px);
    HIPCHK(h, hipGetLastError());
    return DS_OK;
}

int init_phi(ds_handle *h)
{
    long long blocks = (8 * h->plane_elems + ds::kBlock - 1) / ds::kBlock;
    if (blocks > 65536) blocks = 65536;
    if (h->dtype == DS_F64)
        hipLaunchKernelGGL(ds::init_phi_kernel<double>, dim3((unsigned)blocks), dim3(ds::kBlock), 0,
                           h->stream, h->g, static_cast<double *>(h->phi));
    else
        hipLaunchKernelGGL(ds::init_phi_kernel<float>, dim3((unsigned)blocks), dim3(ds::kBlock), 0,
                           h->stream, h->g, static_cast<float *>(h->phi));
    HIPCHK(h, hipGetLastError());
    if (h->fused_waves) {  // the alternate colour-1 planes: same +inf / 0 pattern
        ds::Geometry alt = h->g;
        for (int s = 0; s < 4; ++s) alt.plane_off[1][s] = h->alt_off[s];
        if (h->dtype == DS_F64)
            hipLaunchKernelGGL(ds::init_phi_kernel<double>, dim3((unsigned)blocks), dim3(ds::kBlock), 0,
                               h->stream, alt, static_cast<double *>(h->phi));
        else
            hipLaunchKernelGGL(ds::init_phi_kernel<float>, dim3((unsigned)blocks), dim3(ds::kBlock), 0,
                               h->stream, alt, static_cast<float *>(h->phi));
        HIPCHK(h, hipGetLastError());
    }
    h->fused_pending = false;
    return DS_OK;
}

// Milliseconds of one red + one black half-sweep over the band with the current planes.
int probe_sweeps(ds_handle *h, double *ms)
{
    struct Flag {
        bool &f;
        explicit Flag(bool &x) : f(x) { f = true; }
        ~Flag() { f = false; }
    } flag(h->probing);
    int rc = init_phi(h);
    const int r0 = h->g.row0, r1 = h->g.row0 + h->g.Hb;
    // a handle that iterates with the fused kernel is placed for that kernel's streams
    int f0 = 0, f1 = 0;
    fused_safe_rows(h, &f0, &f1);
    const bool fused = h->fused_waves > 0 && f1 - f0 >= (r1 - r0) / 2;
    auto step = [&](int k) { return fused ? ((k & 1) ? DS_OK : launch_fused(h, f0, f1)) : launch_sweep(h, k & 1, r0, r1); };
    for (int k = 0; k < 2 && rc == DS_OK; ++k) rc = step(k);
    if (rc != DS_OK) return rc;
    hipEvent_t a, b;
    HIPCHK(h, hipEventCreate(&a));
    HIPCHK(h, hipEventCreate(&b));
    HIPCHK(h, hipEventRecord(a, h->stream));
    const int reps = 2;
    for (int k = 0; k < 2 * reps && rc == DS_OK; ++k) rc = step(k);
    hipError_t e = hipEventRecord(b, h->stream);
    if (e == hipSuccess) e = hipEventSynchronize(b);
    float t = 0;
    if (e == hipSuccess) e = hipEventElapsedTime(&t, a, b);
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    if (rc != DS_OK) return rc;
    if (e != hipSuccess) return fail_hip(h, "placement probe", e);
    *ms = t / reps;
    return DS_OK;
}

// Allocates phi: one plain allocation, unless DS_PLACEMENT=search (or exhaustive) asks for the
// placement search below.  Background: MI355X's HBM answers in three 96 GiB physical regions; a
// sweep whose eight streams all fall into one region was measured 15 % slower than one whose
// streams spread over two or three (5.1 vs 6.1 TB/s), and a big hipMalloc is usually one
// contiguous physical range.  The search cuts phi in two plane groups, each holding half of
// every launch's read planes and half of its write planes, and tries group B at growing
// distances behind group A (temporary spacer allocations), timing the handle's own kernel.
// What it gains varies from process to process with where the allocator happens to put things:
// measured on one box, one-iteration kernel, plain / searched: 4K x 256 f32 6.12 / 6.15 ms,
// 4K x 256 f64 11.88 / 11.94, 1080p x 128 f32 0.855 / 0.834, a 540-row band of 8K x 256 f64
// 6.74 / 6.03; a fixed rule (B one region behind A, no timing) gave 6.15 / 12.32 / 0.879 / 6.34,
// no better than the plain allocation on average.  The two-iteration kernel is bound by the
// vector ALU and does not care.  So the default is the deterministic one; the search is for
// callers who create a handle once and iterate for long.
int place_phi_search(ds_handle *h, size_t group_bytes);

int place_phi(ds_handle *h)
{
    const size_t group_bytes = (size_t)(planes_per_group(h) * h->plane_elems) * h->elem;
    const char *mode = std::getenv("DS_PLACEMENT");
    const bool search = mode && (std::strcmp(mode, "search") == 0 || std::strcmp(mode, "exhaustive") == 0);
    if (search && group_bytes >= (1ull << 30)) return place_phi_search(h, group_bytes);
    hipError_t e = hipMalloc(&h->phi, 2 * group_bytes);
    if (e != hipSuccess)
        return fail(h, e == hipErrorOutOfMemory ? DS_ERR_NOMEM : DS_ERR_HIP,
                    std::string("hipMalloc(phi): ") + hipGetErrorString(e));
    set_planes(h, h->phi, static_cast<char *>(h->phi) + group_bytes);
    return DS_OK;
}

// DS_PLACEMENT=search / exhaustive: group B behind a growing stack of 32 GiB spacers, every
// candidate timed with the handle's own kernel, the best kept.
int place_phi_search(ds_handle *h, size_t group_bytes)
{
    const char *mode = std::getenv("DS_PLACEMENT");
    int max_tries = 8;
    if (const char *v = std::getenv("DS_PLACEMENT_TRIES")) max_tries = std::atoi(v) > 0 ? std::atoi(v) : 1;
    const bool verbose = std::getenv("DS_PLACEMENT_VERBOSE") != nullptr;
    const bool exhaustive = mode && std::strcmp(mode, "exhaustive") == 0;
    // 32 GiB steps: three of them cross a 96 GiB region whatever the groups' size
    size_t spacer_bytes = 32ull << 30;
    if (const char *v = std::getenv("DS_PLACEMENT_SPACER_GIB")) spacer_bytes = (size_t)(std::atoi(v) > 0 ? std::atoi(v) : 32) << 30;
    std::vector<void *> spacers;
    auto cleanup = [&]() {
        for (void *p : spacers) (void)hipFree(p);
        spacers.clear();
    };
    hipError_t e = hipMalloc(&h->phi, group_bytes);
    if (e != hipSuccess)
        return fail(h, e == hipErrorOutOfMemory ? DS_ERR_NOMEM : DS_ERR_HIP,
                    std::string("hipMalloc(phi group A): ") + hipGetErrorString(e));
    double best = 0.0;
    int rc = DS_OK, since_gain = 0;
    // The spacers are transient, but other tenants of the GPU see them: never hold more than
    // half of what was free when the search began.
    size_t spacer_budget = 0;
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) spacer_budget = free_b / 2;
    }
    for (int attempt = 0; attempt < max_tries; ++attempt) {
        if (attempt > 0) {  // push the next candidate further away
            size_t free_b = 0, total_b = 0;
            if ((spacers.size() + 1) * spacer_bytes > spacer_budget) break;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < spacer_bytes + group_bytes + (4ull << 30)) break;
            void *sp = nullptr;
            if (hipMalloc(&sp, spacer_bytes) != hipSuccess) {
                (void)hipGetLastError();
                break;
            }
            spacers.push_back(sp);
        }
        void *cand = nullptr;
        e = hipMalloc(&cand, group_bytes);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            if (!h->phi_b) {
                cleanup();
                return fail(h, e == hipErrorOutOfMemory ? DS_ERR_NOMEM : DS_ERR_HIP,
                            std::string("hipMalloc(phi group B): ") + hipGetErrorString(e));
            }
            break;
        }
        set_planes(h, h->phi, cand);
        double ms = 0.0;
        rc = probe_sweeps(h, &ms);
        if (rc != DS_OK) {
            (void)hipFree(cand);
            break;
        }
        ++h->placement_tries;
        if (verbose) std::fprintf(stderr, "[diffstereo] placement try %d (%zu spacers): %.3f ms\n", attempt, spacers.size(), ms);
        since_gain = (!h->phi_b || ms < 0.98 * best) ? 0 : since_gain + 1;
        if (!h->phi_b || ms < best) {
            (void)hipFree(h->phi_b);
            h->phi_b = cand;
            best = ms;
        } else {
            (void)hipFree(cand);
        }
        if (exhaustive) continue;
        // Stop when three more candidates, spread over a whole region, brought nothing.  (A first
        // improvement is not the end: at 4K x 256 the fused kernel measured 7.47, 6.87, 6.37, 6.27,
        // 6.29, 6.85, 6.87 ms per iteration for 0..6 spacers.)
        if (since_gain >= 3 && spacers.size() * spacer_bytes >= (96ull << 30)) break;
    }
    cleanup();
    if (rc != DS_OK) return rc;
    set_planes(h, h->phi, h->phi_b);
    h->placement_ms = best;
    return DS_OK;
}

int create(const uint8_t *left, const uint8_t *right, int H, int W, int D, double lambda, int dtype,
           int row_begin, int row_end, int device, ds_handle **out)
{
    if (!out) return fail(nullptr, DS_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (!left || !right) return fail(nullptr, DS_ERR_ARG, "image pointer is NULL");
    if (H < 1 || W < 1) return fail(nullptr, DS_ERR_ARG, "H and W must be >= 1");
    if (D < 1 || D > 256) return fail(nullptr, DS_ERR_ARG, "D must be in [1, 256]");
    if (dtype != DS_F32 && dtype != DS_F64) return fail(nullptr, DS_ERR_ARG, "dtype must be DS_F32 or DS_F64");
    if (!(lambda == lambda) || std::isinf(lambda)) return fail(nullptr, DS_ERR_ARG, "lambda must be finite");
    if (row_begin < 0 || row_end > H || row_begin >= row_end)
        return fail(nullptr, DS_ERR_ARG, "band must satisfy 0 <= row_begin < row_end <= H");

    ds_handle *h = new ds_handle();
    if (device < 0) {
        hipError_t e = hipGetDevice(&h->device);
        if (e != hipSuccess) {
            delete h;
            return fail_hip(nullptr, "hipGetDevice", e);
        }
    } else {
        h->device = device;
    }
    auto bail = [&](int rc) {
        g_create_error = h->err;
        release(h);
        return rc;
    };
    int rc = use_device(h);
    if (rc == DS_OK) rc = scratch_for_device(h, h->device, nullptr);  // once per device, with it current
    if (rc != DS_OK) return bail(rc);

    ds::Geometry &g = h->g;
    g.H = H; g.W = W; g.D = D;
    g.Dpad = (D + 3) & ~3;
    g.Wh = (W + 1) / 2;
    h->elem = dtype == DS_F64 ? 8 : 4;
    g.row_stride = (long long)g.Wh * g.Dpad;
    h->dtype = dtype;
    h->lambda = lambda;
    h->lpp = g.Dpad <= 64 ? 16 : g.Dpad <= 128 ? 32 : 64;
    // Deep halo (band_deep_step in bands.hpp): a band that steps by two iterations keeps three more
    // rows of each neighbouring band than the one halo row, so that one exchange per step is all it
    // needs.  Asked for with DS_BAND_HALO=deep; taken if the band and its neighbours are tall enough
    // and the band has an interior for the two-iteration kernel (checked below, once the plan exists).
    {
        const char *mode = std::getenv("DS_BAND_HALO");
        const bool band = row_begin > 0 || row_end < H;
        h->deep = mode && std::strcmp(mode, "deep") == 0 && band &&
                  (row_begin == 0 || row_begin >= 4) && (row_end == H || H - row_end >= 4) && row_end - row_begin >= 16;
    }
    auto set_rows = [&]() {
        h->ext_up = h->deep && row_begin > 0 ? 3 : 0;
        h->ext_down = h->deep && row_end < H ? 3 : 0;
        g.row0 = row_begin - h->ext_up;
        g.Hb = row_end - row_begin + h->ext_up + h->ext_down;
        h->plane_elems = (long long)(g.Hb + 2) * g.row_stride;
        g.cost_colour_stride = (long long)g.Hb * g.Wh * g.Dpad;
    };
    set_rows();
    {
        // Wavefronts per block of the fused iteration kernel.  f32 (95 VGPRs): 16, measured at
        // 4K x 256 6.24 ms per iteration against 6.42 with 8 (whose strips re-read 8/7 instead of
        // 16/15 of phi); f64 (178 VGPRs, two wavefronts per SIMD): 8, 6.37 ms on a 540-row band of
        // 8K x 256 against 6.80 with 4.
        const char *e = std::getenv("DS_FUSED");
        bool off = e && std::strcmp(e, "off") == 0;
        // Potentials that fit the GPU's 256 MiB last-level cache (all eight planes: the reference's own
        // working scale, 400 x 400 x 40 in f64, is 205 MB) are iterated by half-sweeps: their planes
        // then come from that cache at several times HBM's rate and a half-sweep is thousands of small
        // blocks, where the fused kernels' few long blocks and the launches for their frame leave most
        // of the GPU idle.  Measured, ms per iteration, fused / half-sweeps: 400 x 400 x 40 f32
        // 0.090 / 0.038, f64 0.090 / 0.067; 300 x 200 x 40 f64 0.088 / 0.029; beyond the cache the
        // fused kernels win: 512 x 512 x 40 f64 0.112 / 0.147, 800 x 600 x 40 f32 0.101 / 0.116.
        // Same bits either way.  DS_CACHE_RESIDENT=fused keeps the fused kernels (for checking them
        // on small images).
        {
            const char *small = std::getenv("DS_CACHE_RESIDENT");
            const bool keep = small && std::strcmp(small, "fused") == 0;
            const size_t eight_planes = (size_t)(8 * h->plane_elems) * (dtype == DS_F64 ? 8 : 4);
            if (!keep && eight_planes <= (256ull << 20)) off = true;
        }
        h->fused_waves = off ? 0 : (dtype == DS_F64 ? 8 : 16);
        if (const char *w = std::getenv("DS_FUSED_WAVES"))  // tuning knob
            if (h->fused_waves && (std::atoi(w) == 4 || std::atoi(w) == 8 || (std::atoi(w) == 16 && dtype != DS_F64)))
                h->fused_waves = std::atoi(w);
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device) == hipSuccess && cus > 0)
            h->cu_count = cus;
    }
    {
        const int ppw = 64 / h->lpp;
        const int per_round = (ds::kBlock / 64) * ppw;  // pixels one pass of the block's 4 waves covers
        // Passes of the block's 4 wavefronts over its segment.  Measured at 4K (ms per launch,
        // 1 / 2 / 4 passes): D=256 f32 5.38 / 5.44 / 5.48, f64 11.13 / 11.39 / 11.52;
        // D=128 f32 2.98 / 2.72 / 2.76, f64 5.67 / - / 5.80; D=40 f32 0.99 / 0.91 / 0.92.
        int rounds = (g.Wh + per_round - 1) / per_round;
        int cap = (h->lpp == 64 || (h->lpp == 32 && dtype == DS_F64)) ? 1 : 2;
        if (const char *e = std::getenv("DS_ROUNDS")) cap = std::atoi(e) > 0 ? std::atoi(e) : cap;  // tuning knob
        if (rounds > cap) rounds = cap;
        if (rounds < 1) rounds = 1;
        h->seg_pixels = rounds * per_round;
        h->grid_x = (g.Wh + h->seg_pixels - 1) / h->seg_pixels;
    }
    plan_fused2(h);
    if (h->deep && !h->plan2.ok) {  // no interior for the two-iteration kernel: an ordinary band
        h->deep = false;
        set_rows();
        plan_fused2(h);
    }

#define CREATE_HIPCHK(call)                                                    \
    do {                                                                       \
        hipError_t e_ = (call);                                                \
        if (e_ != hipSuccess) {                                                \
            int rc_ = e_ == hipErrorOutOfMemory ? DS_ERR_NOMEM : DS_ERR_HIP;   \
            h->err = std::string(#call) + ": " + hipGetErrorString(e_);        \
            return bail(rc_);                                                  \
        }                                                                      \
    } while (0)

    CREATE_HIPCHK(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    h->stream = h->own_stream;
    const size_t img = (size_t)H * W;
    CREATE_HIPCHK(hipMalloc(&h->d_left, img));
    CREATE_HIPCHK(hipMalloc(&h->d_right, img));
    CREATE_HIPCHK(hipMemcpyAsync(h->d_left, left, img, hipMemcpyHostToDevice, h->stream));
    CREATE_HIPCHK(hipMemcpyAsync(h->d_right, right, img, hipMemcpyHostToDevice, h->stream));
    CREATE_HIPCHK(hipMalloc(&h->cost_lm, (size_t)D * g.Hb * W + 4));
    CREATE_HIPCHK(hipMalloc(&h->cost_px, (size_t)(2 * g.cost_colour_stride)));
    // phi is allocated last, by place_phi(), once the kernels it probes with can run
    h->part_n = 2ll * h->grid_x * (row_end - row_begin);  // one partial per block of the two energy launches
    // + kReduceBlocks entries behind the partials for the first reduction stage's results
    CREATE_HIPCHK(hipMalloc(&h->part_sum, (size_t)(h->part_n + kReduceBlocks) * sizeof(double)));
    CREATE_HIPCHK(hipMalloc(&h->part_max, (size_t)(h->part_n + kReduceBlocks) * sizeof(double)));
    CREATE_HIPCHK(hipMalloc(&h->d_scalars, 2 * sizeof(double)));
    CREATE_HIPCHK(hipMalloc(&h->d_checksum, sizeof(unsigned long long)));
    CREATE_HIPCHK(hipMalloc(&h->d_labels, (size_t)g.Hb * W * sizeof(uint32_t)));
    if (h->deep) CREATE_HIPCHK(hipMalloc(&h->deep_stage, (size_t)(4 * 16 * g.row_stride) * h->elem));
#undef CREATE_HIPCHK

    rc = build_cost(h);
    if (rc == DS_OK) rc = place_phi(h);
    if (rc == DS_OK) rc = init_phi(h);
    if (rc == DS_OK) {
        hipError_t e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) rc = fail_hip(h, "hipStreamSynchronize", e);
    }
    if (rc != DS_OK) return bail(rc);
    *out = h;
    return DS_OK;
}

// Host <-> device conversion between the reference layout [rows][W][4][D] and the
// colour-split device layout, for image rows [rb, re) (halo rows allowed).
template <typename T> int copy_phi_rows(ds_handle *h, int rb, int re, T *host_out, const T *host_in)
{
    const ds::Geometry &g = h->g;
    const int rows = re - rb;
    const long long chunk = (long long)rows * g.Wh * g.Dpad;  // per (colour, slot)
    std::vector<T> stage((size_t)(8 * chunk));
    const long long rowpitch = (long long)g.Wh * g.Dpad;
    const long long first = (long long)(rb - g.row0 + 1) * g.row_stride;
    auto plane_off = [&](int p) { return g.plane_off[p / 4][p % 4] + first; };
    T *dev = static_cast<T *>(h->phi);
    const T inf = std::numeric_limits<T>::infinity();
    if (host_out) {
        for (int p = 0; p < 8; ++p)
            HIPCHK(h, hipMemcpy2DAsync(stage.data() + p * chunk, (size_t)rowpitch * sizeof(T), dev + plane_off(p),
                                       (size_t)g.row_stride * sizeof(T), (size_t)rowpitch * sizeof(T), rows,
                                       hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    } else {
        for (auto &x : stage) x = inf;
    }
    for (int r = 0; r < rows; ++r) {
        const int i = rb + r;
        for (int j = 0; j < g.W; ++j) {
            const int colour = (i + j) & 1, jc = j >> 1;
            const int K = j < g.D - 1 ? j : g.D - 1;
            for (int n = 0; n < 4; ++n) {
                T *s = stage.data() + (colour * 4 + n) * chunk + ((long long)r * g.Wh + jc) * g.Dpad;
                const long long hoff = (((long long)r * g.W + j) * 4 + n) * g.D;
                if (host_out) {
                    for (int d = 0; d < g.D; ++d) host_out[hoff + d] = d <= K ? s[d] : T(0);
                } else {
                    for (int d = 0; d <= K; ++d) s[d] = host_in[hoff + d];
                }
            }
        }
    }
    if (host_in) {
        for (int p = 0; p < 8; ++p)
            HIPCHK(h, hipMemcpy2DAsync(dev + plane_off(p), (size_t)g.row_stride * sizeof(T), stage.data() + p * chunk,
                                       (size_t)rowpitch * sizeof(T), (size_t)rowpitch * sizeof(T), rows,
                                       hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    return DS_OK;
}

int phi_rows(ds_handle *h, int rb, int re, void *out, const void *in)
{
    if (!h) return DS_ERR_ARG;
    if (!out && !in) return fail(h, DS_ERR_ARG, "buffer is NULL");
    const ds::Geometry &g = h->g;
    const int lo = g.row0 > 0 ? g.row0 - 1 : 0;
    const int hi = g.row0 + g.Hb < g.H ? g.row0 + g.Hb + 1 : g.H;
    if (rb < lo || re > hi || rb > re) return fail(h, DS_ERR_ARG, "row range outside the band and its halo");
    if (rb == re) return DS_OK;
    int rc = need_current(h, out ? "ds_get_phi" : "ds_set_phi");
    if (rc == DS_OK) rc = use_device(h);
    if (rc != DS_OK) return rc;
    return guarded(h, [&]() -> int {
        if (h->dtype == DS_F64)
            return copy_phi_rows<double>(h, rb, re, static_cast<double *>(out), static_cast<const double *>(in));
        return copy_phi_rows<float>(h, rb, re, static_cast<float *>(out), static_cast<const float *>(in));
    });
}

}  // namespace

extern "C" {

int ds_create(const uint8_t *left, const uint8_t *right, int H, int W, int D, double lambda, int dtype,
              ds_handle **out)
{
    return guarded(nullptr, [&]() { return create(left, right, H, W, D, lambda, dtype, 0, H, -1, out); });
}

int ds_create_band(const uint8_t *left, const uint8_t *right, int H, int W, int D, double lambda,
                   int dtype, int row_begin, int row_end, int device, ds_handle **out)
{
    return guarded(nullptr, [&]() {
        return create(left, right, H, W, D, lambda, dtype, row_begin, row_end, device, out);
    });
}

void ds_destroy(ds_handle *h) { release(h); }

const char *ds_last_error(const ds_handle *h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int ds_set_stream(ds_handle *h, void *hip_stream)
{
    if (!h) return DS_ERR_ARG;
    if (h->raster_pending) {
        const int rc_pending = flush_raster(h);
        if (rc_pending != DS_OK) return rc_pending;
    }
    int rc = use_device(h);
    if (rc != DS_OK) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->stream = static_cast<hipStream_t>(hip_stream);
    return DS_OK;
}

int ds_switch_stream(ds_handle *h, void *hip_stream)
{
    if (!h) return DS_ERR_ARG;
    h->stream = static_cast<hipStream_t>(hip_stream);
    return DS_OK;
}

int ds_synchronize(ds_handle *h)
{
    if (!h) return DS_ERR_ARG;
    if (h->raster_pending) {
        const int rc_pending = flush_raster(h);
        if (rc_pending != DS_OK) return rc_pending;
    }
    int rc = use_device(h);
    if (rc != DS_OK) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return DS_OK;
}

int ds_build_cost(ds_handle *h)
{
    if (!h) return DS_ERR_ARG;
    int rc = use_device(h);
    if (rc != DS_OK) return rc;
    rc = build_cost(h);
    if (rc != DS_OK) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return DS_OK;
}

int ds_get_cost(ds_handle *h, uint8_t *out)
{
    if (!h) return DS_ERR_ARG;
    if (!out) return fail(h, DS_ERR_ARG, "out is NULL");
    int rc = use_device(h);
    if (rc != DS_OK) return rc;
    // [D][stored rows][W] on the device, the band's own rows of every label to the host
    const size_t own_bytes = (size_t)(own_end(h) - own_begin(h)) * h->g.W, stored_bytes = (size_t)h->g.Hb * h->g.W;
    HIPCHK(h, hipMemcpy2DAsync(out, own_bytes, h->cost_lm + (size_t)h->ext_up * h->g.W, stored_bytes, own_bytes,
                               (size_t)h->g.D, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return DS_OK;
}

int ds_sweep(ds_handle *h, int colour, int row_begin, int row_end)
{
    if (!h) return DS_ERR_ARG;
    if (h->raster_pending) {
        const int rc_pending = flush_raster(h);
        if (rc_pending != DS_OK) return rc_pending;
    }
    if (colour != 0 && colour != 1) return fail(h, DS_ERR_ARG, "colour must be 0 or 1");
    if (row_begin < h->g.row0 || row_end > h->g.row0 + h->g.Hb || row_begin > row_end)
        return fail(h, DS_ERR_ARG, "rows outside the handle's band");
    int rc = not_deep(h, "ds_sweep");
    if (rc == DS_OK) rc = use_device(h);
    if (rc != DS_OK) return rc;
    return launch_sweep(h, colour, row_begin, row_end);
}

int ds_sweep_pair(ds_handle *h, int colour, int next, int row_begin, int row_end, int row_begin2, int row_end2)
{
    if (!h) return DS_ERR_ARG;
    if (h->raster_pending) {
        const int rc_pending = flush_raster(h);
        if (rc_pending != DS_OK) return rc_pending;
    }
    if (colour != 0 && colour != 1) return fail(h, DS_ERR_ARG, "colour must be 0 or 1");
    if (next && (colour != 1 || !h->fused_waves))
        return fail(h, DS_ERR_ARG, "only the second colour has upcoming planes (and not with DS_FUSED=off)");
    const int lo = h->g.row0, hi = h->g.row0 + h->g.Hb;
    if (row_begin < lo || row_end > hi || row_begin > row_end || row_begin2 < lo || row_end2 > hi ||
        row_begin2 > row_end2 || (row_begin2 < row_end && row_begin < row_end2 && row_begin < row_end && row_begin2 < row_end2))
        return fail(h, DS_ERR_ARG, "row ranges must lie inside the handle's band and not overlap");
    int rc = not_deep(h, "ds_sweep_pair");
    if (rc == DS_OK) rc = use_device(h);
    if (rc != DS_OK) return rc;
    if (!next) return launch_sweep(h, colour, row_begin, row_end, nullptr, row_begin2, row_end2);
    ds::Geometry upcoming = h->g;
    for (int s = 0; s < 4; ++s) upcoming.plane_off[1][s] = h->alt_off[s];
    return launch_sweep(h, 1, row_begin, row_end, &upcoming, row_begin2, row_end2);
}

int ds_fused_rows(const ds_handle *h, int *row_begin, int *row_end)
{
    if (!h || !row_begin || !row_end) return DS_ERR_ARG;
    *row_begin = *row_end = own_begin(h);
    if (h->fused_waves && !h->deep) fused_safe_rows(h, row_begin, row_end);
    if (*row_end < *row_begin) *row_end = *row_begin;
    return DS_OK;
}

int ds_iterate_rows(ds_handle *h, int row_begin, int row_end)
{
    if (!h) return DS_ERR_ARG;
    if (h->raster_pending) {
        const int rc_pending = flush_raster(h);
        if (rc_pending != DS_OK) return rc_pending;
    }
    if (!h->fused_waves) return fail(h, DS_ERR_STATE, "the fused iteration kernel is switched off on this handle (DS_FUSED=off)");
    if (row_begin >= row_end) return fail(h, DS_ERR_ARG, "ds_iterate_rows needs at least one row");
    int rc = not_deep(h, "ds_iterate_rows");
    if (rc == DS_OK) rc = use_device(h);
    if (rc != DS_OK) return rc;
    return launch_fused(h, row_begin, row_end);
}

int ds_sweep_next(ds_handle *h, int row_begin, int row_end)
{
    if (!h) return DS_ERR_ARG;
    if (h->raster_pending) {
        const int rc_pending = flush_raster(h);
        if (rc_pending != DS_OK) return rc_pending;
    }
    if (!h->fused_waves) return fail(h, DS_ERR_STATE, "this handle has no second set of colour-1 planes");
    if (row_begin < h->g.row0 || row_end > h->g.row0 + h->g.Hb || row_begin > row_end)
        return fail(h, DS_ERR_ARG, "rows outside the handle's band");
    int rc = not_deep(h, "ds_sweep_next");
    if (rc == DS_OK) rc = use_device(h);
    if (rc != DS_OK) return rc;
    ds::Geometry next = h->g;
    for (int s = 0; s < 4; ++s) next.plane_off[1][s] = h->alt_off[s];
    return launch_sweep(h, 1, row_begin, row_end, &next);
}

int ds_finish_iterations(ds_handle *h)
{
    if (!h) return DS_ERR_ARG;
    if (!h->fused_waves) return fail(h, DS_ERR_STATE, "this handle has no fused iteration kernel");
    int rc = not_deep(h, "ds_finish_iterations");
    if (rc == DS_OK) rc = use_device(h);
    if (rc != DS_OK) return rc;
    return materialise_first_colour(h, h->g.row0, h->g.row0 + h->g.Hb);
}

int ds_iterate(ds_handle *h, int n_iters)
{
    if (!h) return DS_ERR_ARG;
    if (h->raster_pending) {
        const int rc_pending = flush_raster(h);
        if (rc_pending != DS_OK) return rc_pending;
    }
    if (n_iters < 0) return fail(h, DS_ERR_ARG, "n_iters must be >= 0");
    if (h->g.row0 != 0 || h->g.Hb != h->g.H)
        return fail(h, DS_ERR_STATE, "ds_iterate needs the whole image; a band must use ds_sweep + halo exchange");
    int rc = use_device(h);
    if (rc != DS_OK) return rc;
    if (h->fused_waves && n_iters > 0) {
        // n whole iterations that move only the second colour's planes; the call's last launch
        // also stores the first colour (the kernels compute it on the way): the same bits as 2n
        // half-sweeps.  Two iterations per launch while two or more remain.
        // DS_STORE_FIRST=off, for checking: the last iteration runs alone and a half-sweep brings the
        // first colour up to date from the second colour's planes of the iteration before
        const char *sf = std::getenv("DS_STORE_FIRST");
        const bool fold = !(sf && std::strcmp(sf, "off") == 0);
        const int keep = fold ? 2 : 3;
        int left = n_iters;
        const bool by_two = left >= keep && prepare_fused2(h);
        struct Restore {
            ds_handle *h;
            ~Restore() { h->store_first = false; }
        } restore{h};
        while (by_two && left >= keep) {
            h->store_first = fold && left == 2;
            rc = fused2_step(h);
            if (rc != DS_OK) return rc;
            ++h->fused2_steps;
            left -= 2;
        }
        for (; left > 0; --left) {
            h->store_first = fold && left == 1;
            rc = launch_fused(h, 0, h->g.H);
            if (rc != DS_OK) return rc;
        }
        h->store_first = false;
        rc = materialise_first_colour(h, 0, h->g.H);  // nothing left to do unless DS_STORE_FIRST=off
        if (rc != DS_OK) return rc;
    } else {
        for (int it = 0; it < n_iters; ++it)
            for (int colour = 0; colour < 2; ++colour) {
                rc = launch_sweep(h, colour, 0, h->g.H);
                if (rc != DS_OK) return rc;
            }
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->side_stream) HIPCHK(h, hipStreamSynchronize(h->side_stream));
    return DS_OK;
}

// n raster-order iterations as K + 2 (n - 1) dependent launches (see sweep_diagonals_kernel):
// launch s carries diagonal s - 2 t of every iteration t that is live at that step.
static int launch_raster(ds_handle *h, int n_iters)
{
    const ds::Geometry &g = h->g;
    const int K = g.H + g.W - 1;
    const int per_block = (ds::kBlock / 64) * (64 / h->lpp);  // pixels per block
    const int longest = g.H < g.W ? g.H : g.W;
    const unsigned grid_x = (unsigned)((longest + per_block - 1) / per_block);
    for (long long s = 0; s <= (long long)K - 1 + 2ll * (n_iters - 1); ++s) {
        // live iterations: 0 <= s - 2 t <= K - 1 and 0 <= t < n
        const long long lo = s - (K - 1) > 0 ? (s - (K - 1) + 1) / 2 : 0;
        long long hi = s / 2;
        if (hi > n_iters - 1) hi = n_iters - 1;
        if (hi < lo) continue;
        // at most 65535 iterations per launch (grid.y); the rest of the step in further launches,
        // which are independent of one another
        for (long long t0 = lo; t0 <= hi; t0 += 65535) {
            const unsigned live = (unsigned)((hi - t0 + 1) < 65535 ? (hi - t0 + 1) : 65535);
            const dim3 grid(grid_x, live), block(ds::kBlock);
            int rc = dispatch(h, [&](auto t, auto lpp) -> int {
                using T = decltype(t);
                hipLaunchKernelGGL((ds::sweep_diagonals_kernel<T, decltype(lpp)::value>), grid, block, 0, h->stream, g,
                                   static_cast<T *>(h->phi), h->cost_px, (T)h->lambda, (int)s, (int)t0);
                return DS_OK;
            });
            if (rc != DS_OK) return rc;
            HIPCHK(h, hipGetLastError());
        }
    }
    return DS_OK;
}

// Queues the raster-order iterations that ds_iterate_raster_async has collected, as one batch.
namespace {
int flush_raster(ds_handle *h)
{
    const long long n = h->raster_pending;
    h->raster_pending = 0;
    if (n <= 0) return DS_OK;
    int rc = use_device(h);
    if (rc != DS_OK) return rc;
    for (long long done = 0; done < n && rc == DS_OK;) {
        const int batch = (int)(n - done < 1000000 ? n - done : 1000000);
        rc = launch_raster(h, batch);
        done += batch;
    }
    return rc;
}
}  // namespace

int ds_iterate_raster_async(ds_handle *h, int n_iters)
{
    if (!h) return DS_ERR_ARG;
    if (n_iters < 0) return fail(h, DS_ERR_ARG, "n_iters must be >= 0");
    const ds::Geometry &g = h->g;
    if (g.row0 != 0 || g.Hb != g.H) return fail(h, DS_ERR_STATE, "ds_iterate_raster_async needs the whole image");
    if (h->fused_pending)
        return fail(h, DS_ERR_STATE, "ds_iterate_raster_async: the first colour's planes are out of date after "
                                     "ds_iterate_rows; call ds_finish_iterations first");
    h->raster_pending += n_iters;
    return DS_OK;
}

int ds_iterate_raster(ds_handle *h, int n_iters)
{
    if (!h) return DS_ERR_ARG;
    if (n_iters < 0) return fail(h, DS_ERR_ARG, "n_iters must be >= 0");
    const ds::Geometry &g = h->g;
    if (g.row0 != 0 || g.Hb != g.H)
        return fail(h, DS_ERR_STATE, "ds_iterate_raster needs the whole image");
    if (h->fused_pending) return need_current(h, "ds_iterate_raster");  // the error text; nothing is queued
    // The launches of a step depend on the step before, so the cost is the GPU's kernel-to-kernel
    // turnaround (replaying them from a recorded HIP graph was measured and changes nothing).  With
    // n iterations in one call the steps number H + W - 1 + 2 (n - 1) instead of n (H + W - 1),
    // and each carries up to n diagonals: the longer the call, the closer to the rate the update
    // itself allows.  Iterations noted by ds_iterate_raster_async join the batch.
    h->raster_pending += n_iters;
    int rc = flush_raster(h);
    if (rc != DS_OK) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return DS_OK;
}

// The energy launches: two first-stage launches (one per colour) into part_sum / part_max
// [0, part_n), a 1024-block reduction into block_sum / block_max, a one-block reduction into
// result[0] (energy) and result[1] (largest |f_t(d)|).  Queued on the handle's stream.
static int launch_energy(ds_handle *h, double *part_sum, double *part_max, double *block_sum, double *block_max,
                         double *result)
{
    const ds::Geometry own = own_view(h);
    const dim3 grid(h->grid_x, own.Hb), block(ds::kBlock);
    const long long half = h->part_n / 2;
    for (int colour = 0; colour < 2; ++colour) {
        int rc = dispatch(h, [&](auto t, auto lpp) -> int {
            using T = decltype(t);
            hipLaunchKernelGGL((ds::energy_kernel<T, decltype(lpp)::value>), grid, block, 0, h->stream,
                               own, static_cast<const T *>(h->phi), own_cost(h), (T)h->lambda, colour,
                               h->seg_pixels, part_sum + colour * half, part_max + colour * half);
            return DS_OK;
        });
        if (rc != DS_OK) return rc;
        HIPCHK(h, hipGetLastError());
    }
    const long long chunk = (h->part_n + kReduceBlocks - 1) / kReduceBlocks;
    hipLaunchKernelGGL(ds::reduce_partials_kernel, dim3(kReduceBlocks), block, 0, h->stream, part_sum, part_max,
                       h->part_n, chunk, block_sum, block_max);
    HIPCHK(h, hipGetLastError());
    hipLaunchKernelGGL(ds::reduce_partials_kernel, dim3(1), block, 0, h->stream, block_sum, block_max,
                       (long long)kReduceBlocks, (long long)kReduceBlocks, result, result + 1);
    HIPCHK(h, hipGetLastError());
    return DS_OK;
}

int ds_energy(ds_handle *h, double *energy, double *max_residual)
{
    if (!h) return DS_ERR_ARG;
    if (!energy) return fail(h, DS_ERR_ARG, "energy is NULL");
    int rc = need_current(h, "ds_energy");
    if (rc == DS_OK) rc = use_device(h);
    if (rc != DS_OK) return rc;
    rc = launch_energy(h, h->part_sum, h->part_max, h->part_sum + h->part_n, h->part_max + h->part_n, h->d_scalars);
    if (rc != DS_OK) return rc;
    double host[2] = {0, 0};
    HIPCHK(h, hipMemcpyAsync(host, h->d_scalars, sizeof host, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    *energy = host[0];
    if (max_residual) *max_residual = host[1];
    return DS_OK;
}

int ds_energy_static(ds_handle *h, double *result_dev)
{
    if (!h) return DS_ERR_ARG;
    if (!result_dev) return fail(h, DS_ERR_ARG, "result_dev is NULL");
    if (h->part_n > kStaticPartials)
        return fail(h, DS_ERR_ARG,
                    "ds_energy_static: this handle needs " + std::to_string(h->part_n) +
                        " partial sums, the static scratch holds " + std::to_string(kStaticPartials) +
                        " (use ds_energy, or a band with fewer rows)");
    int rc = need_current(h, "ds_energy_static");
    if (rc == DS_OK) rc = use_device(h);
    if (rc != DS_OK) return rc;
    // the two doubles are written by a kernel: pageable host memory would fault it
    hipPointerAttribute_t attr{};
    if (hipPointerGetAttributes(&attr, result_dev) != hipSuccess) {
        (void)hipGetLastError();
        return fail(h, DS_ERR_ARG, "ds_energy_static: result_dev must be device memory or pinned host memory");
    }
    if (attr.type != hipMemoryTypeDevice && attr.type != hipMemoryTypeHost && attr.type != hipMemoryTypeManaged)
        return fail(h, DS_ERR_ARG, "ds_energy_static: result_dev must be device memory or pinned host memory");
    double *base = (h->device >= 0 && h->device < kMaxDevices) ? g_scratch_ptr[h->device] : nullptr;
    if (!base) return fail(h, DS_ERR_STATE, "ds_energy_static: no scratch address for this device");
    double *sums = base, *maxes = base + kStaticPartials + kReduceBlocks;
    // Nothing needs clearing: every entry the reductions read ([0, part_n) and the kReduceBlocks
    // block results) is written earlier in this same call.
    return launch_energy(h, sums, maxes, sums + kStaticPartials, maxes + kStaticPartials, result_dev);
}

int ds_labels(ds_handle *h, int mode, uint32_t *out)
{
    if (!h) return DS_ERR_ARG;
    if (!out) return fail(h, DS_ERR_ARG, "out is NULL");
    if (mode != 0 && mode != 1) return fail(h, DS_ERR_ARG, "mode must be 0 or 1");
    int rc = need_current(h, "ds_labels");
    if (rc == DS_OK) rc = use_device(h);
    if (rc != DS_OK) return rc;
    const ds::Geometry own = own_view(h);
    const dim3 grid(h->grid_x, own.Hb), block(ds::kBlock);
    for (int colour = 0; colour < 2; ++colour) {
        rc = dispatch(h, [&](auto t, auto lpp) -> int {
            using T = decltype(t);
            hipLaunchKernelGGL((ds::labels_kernel<T, decltype(lpp)::value>), grid, block, 0, h->stream,
                               own, static_cast<const T *>(h->phi), own_cost(h), (T)h->lambda, colour,
                               h->seg_pixels, mode, h->d_labels);
            return DS_OK;
        });
        if (rc != DS_OK) return rc;
        HIPCHK(h, hipGetLastError());
    }
    HIPCHK(h, hipMemcpyAsync(out, h->d_labels, (size_t)own.Hb * own.W * sizeof(uint32_t),
                             hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return DS_OK;
}

int ds_get_phi_rows(ds_handle *h, int row_begin, int row_end, void *out)
{
    return phi_rows(h, row_begin, row_end, out, nullptr);
}

int ds_set_phi_rows(ds_handle *h, int row_begin, int row_end, const void *in)
{
    return phi_rows(h, row_begin, row_end, nullptr, in);
}

int ds_get_phi(ds_handle *h, void *out)
{
    if (!h) return DS_ERR_ARG;
    return phi_rows(h, own_begin(h), own_end(h), out, nullptr);
}

int ds_set_phi(ds_handle *h, const void *in)
{
    if (!h) return DS_ERR_ARG;
    return phi_rows(h, own_begin(h), own_end(h), nullptr, in);
}

int ds_halo(ds_handle *h, int colour, int side, int which, void **dev_ptr, size_t *bytes)
{
    if (!h) return DS_ERR_ARG;
    if ((colour | 1) != 1 || (side | 1) != 1 || (which | 1) != 1 || !dev_ptr || !bytes)
        return fail(h, DS_ERR_ARG, "bad ds_halo argument");
    if (not_deep(h, "ds_halo") != DS_OK) return DS_ERR_STATE;
    const ds::Geometry &g = h->g;
    // top: send slot 1 of band row 1, receive slot 3 into halo row 0;
    // bottom: send slot 3 of band row Hb, receive slot 1 into halo row Hb + 1.
    const int slot = side == 0 ? (which == 0 ? 1 : 3) : (which == 0 ? 3 : 1);
    const int il = side == 0 ? (which == 0 ? 1 : 0) : (which == 0 ? g.Hb : g.Hb + 1);
    const long long off = g.plane_off[colour][slot] + (long long)il * g.row_stride;
    *dev_ptr = static_cast<char *>(h->phi) + (size_t)off * h->elem;
    *bytes = (size_t)g.Wh * g.Dpad * h->elem;
    return DS_OK;
}

int ds_halo_next(ds_handle *h, int side, int which, void **dev_ptr, size_t *bytes)
{
    if (!h) return DS_ERR_ARG;
    if ((side | 1) != 1 || (which | 1) != 1 || !dev_ptr || !bytes) return fail(h, DS_ERR_ARG, "bad ds_halo_next argument");
    if (!h->fused_waves) return fail(h, DS_ERR_STATE, "this handle has no second set of colour-1 planes");
    if (not_deep(h, "ds_halo_next") != DS_OK) return DS_ERR_STATE;
    const ds::Geometry &g = h->g;
    const int slot = side == 0 ? (which == 0 ? 1 : 3) : (which == 0 ? 3 : 1);
    const int il = side == 0 ? (which == 0 ? 1 : 0) : (which == 0 ? g.Hb : g.Hb + 1);
    const long long off = h->alt_off[slot] + (long long)il * g.row_stride;
    *dev_ptr = static_cast<char *>(h->phi) + (size_t)off * h->elem;
    *bytes = (size_t)g.Wh * g.Dpad * h->elem;
    return DS_OK;
}

int ds_profile_begin(ds_handle *h, int max_launches)
{
    if (!h) return DS_ERR_ARG;
    if (max_launches < 1 || max_launches > (1 << 20)) return fail(h, DS_ERR_ARG, "max_launches out of range");
    int rc = use_device(h);
    if (rc != DS_OK) return rc;
    return guarded(h, [&]() -> int {
        while ((int)h->events.size() < 2 * max_launches) {
            hipEvent_t e;
            HIPCHK(h, hipEventCreate(&e));
            h->events.push_back(e);
        }
        h->prof_cap = max_launches;
        h->prof_kind.assign(max_launches, 0);
        h->prof_n = 0;
        h->prof = true;
        return DS_OK;
    });
}

int ds_profile_end(ds_handle *h, int *n_launches, double *total_ms, double *min_ms, double *max_ms)
{
    if (!h) return DS_ERR_ARG;
    int rc = use_device(h);
    if (rc != DS_OK) return rc;
    h->prof = false;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    double tot = 0, lo = 0, hi = 0;
    for (int k = 0; k < h->prof_n; ++k) {
        float ms = 0;
        HIPCHK(h, hipEventElapsedTime(&ms, h->events[2 * k], h->events[2 * k + 1]));
        tot += ms;
        lo = k == 0 ? ms : (ms < lo ? ms : lo);
        hi = k == 0 ? ms : (ms > hi ? ms : hi);
    }
    if (n_launches) *n_launches = h->prof_n;
    if (total_ms) *total_ms = tot;
    if (min_ms) *min_ms = lo;
    if (max_ms) *max_ms = hi;
    return DS_OK;
}

int ds_profile_split(ds_handle *h, int *n_half_sweeps, double *half_sweep_ms, int *n_fused, double *fused_ms)
{
    if (!h) return DS_ERR_ARG;
    if (h->prof) return fail(h, DS_ERR_STATE, "ds_profile_split before ds_profile_end");
    int rc = use_device(h);
    if (rc != DS_OK) return rc;
    int n[2] = {0, 0};
    double t[2] = {0, 0};
    for (int k = 0; k < h->prof_n; ++k) {
        float ms = 0;
        HIPCHK(h, hipEventElapsedTime(&ms, h->events[2 * k], h->events[2 * k + 1]));
        const int kind = h->prof_kind[k];
        if (kind > 1) continue;  // two-iteration launches: ds_profile_kind(h, 2, ...)
        ++n[kind];
        t[kind] += ms;
    }
    if (n_half_sweeps) *n_half_sweeps = n[0];
    if (half_sweep_ms) *half_sweep_ms = t[0];
    if (n_fused) *n_fused = n[1];
    if (fused_ms) *fused_ms = t[1];
    return DS_OK;
}

int ds_profile_kind(ds_handle *h, int kind, int *n_launches, double *ms)
{
    if (!h) return DS_ERR_ARG;
    if (kind < 0 || kind > 2) return fail(h, DS_ERR_ARG, "kind must be 0, 1 or 2");
    if (h->prof) return fail(h, DS_ERR_STATE, "ds_profile_kind before ds_profile_end");
    int rc = use_device(h);
    if (rc != DS_OK) return rc;
    int n = 0;
    double t = 0;
    for (int k = 0; k < h->prof_n; ++k) {
        if (h->prof_kind[k] != kind) continue;
        float one = 0;
        HIPCHK(h, hipEventElapsedTime(&one, h->events[2 * k], h->events[2 * k + 1]));
        ++n;
        t += one;
    }
    if (n_launches) *n_launches = n;
    if (ms) *ms = t;
    return DS_OK;
}

int ds_fused2_info(const ds_handle *h, int *available, int *col_begin, int *col_end, int *row_begin, int *row_end,
                   long long *steps)
{
    if (!h) return DS_ERR_ARG;
    if (available) *available = h->plan2.ok ? 1 : 0;
    if (col_begin) *col_begin = h->plan2.cL;
    if (col_end) *col_end = h->plan2.cR;
    if (row_begin) *row_begin = h->plan2.rT;
    if (row_end) *row_end = h->plan2.rB;
    if (steps) *steps = h->fused2_steps;
    return DS_OK;
}

int ds_info(const ds_handle *h, int *H, int *W, int *D, int *dtype, int *row_begin, int *row_end)
{
    if (!h) return DS_ERR_ARG;
    if (H) *H = h->g.H;
    if (W) *W = h->g.W;
    if (D) *D = h->g.D;
    if (dtype) *dtype = h->dtype;
    if (row_begin) *row_begin = own_begin(h);
    if (row_end) *row_end = own_end(h);
    return DS_OK;
}

// The crossing-out state of a handle (masks, pair minima, flags), allocated on first use and
// cleared: all labels dead, the mask's halo rows all alive.
static int co_prepare(ds_handle *h, const ds::Geometry &g, double epsilon)
{
    const size_t mask_bytes = (size_t)(2 * ds::mask_colour_stride(g));
    h->co_begun = false;
    if (!h->co_alive || !h->co_gmin || !h->co_flags || !h->co_dirty) {
        // all or none: a failure must not leave a half-allocated state behind that a
        // later call would take for complete
        void *alive = nullptr, *gmin = nullptr, *flags = nullptr, *dirty = nullptr;
        hipError_t e = hipMalloc(&alive, mask_bytes);
        if (e == hipSuccess) e = hipMalloc(&gmin, (size_t)2 * g.Hb * g.Wh * 4 * h->elem);
        if (e == hipSuccess) e = hipMalloc(&flags, 2 * sizeof(int));
        if (e == hipSuccess) e = hipMalloc(&dirty, (size_t)2 * g.Hb * h->grid_x);
        if (e != hipSuccess) {
            (void)hipFree(alive);
            (void)hipFree(gmin);
            (void)hipFree(flags);
            (void)hipFree(dirty);
            (void)hipGetLastError();
            return fail(h, e == hipErrorOutOfMemory ? DS_ERR_NOMEM : DS_ERR_HIP,
                        std::string("hipMalloc(crossing-out state): ") + hipGetErrorString(e));
        }
        (void)hipFree(h->co_alive);
        (void)hipFree(h->co_gmin);
        (void)hipFree(h->co_flags);
        (void)hipFree(h->co_dirty);
        h->co_alive = static_cast<uint8_t *>(alive);
        h->co_gmin = gmin;
        h->co_flags = static_cast<int *>(flags);
        h->co_dirty = static_cast<uint8_t *>(dirty);
    }
    h->co_epsilon = epsilon;
    h->co_fixed = false;
    h->co_dirty_all = true;  // the first pass looks at every pixel
    h->co_passes_done = 0;
    HIPCHK(h, hipMemsetAsync(h->co_alive, 0, mask_bytes, h->stream));
    // Halo rows start all-alive: until the neighbouring band's rows arrive they must not
    // take support away (deletions cannot be undone; a late deletion only costs a pass).
    const size_t row_bytes = (size_t)g.Wh * g.Dpad;
    for (int colour = 0; colour < 2; ++colour) {
        uint8_t *base = h->co_alive + colour * ds::mask_colour_stride(g);
        HIPCHK(h, hipMemsetAsync(base, 1, row_bytes, h->stream));
        HIPCHK(h, hipMemsetAsync(base + (size_t)(g.Hb + 1) * row_bytes, 1, row_bytes, h->stream));
    }
    return DS_OK;
}

int ds_co_begin(ds_handle *h, double epsilon)
{
    if (!h) return DS_ERR_ARG;
    if (!(epsilon >= 0.0)) return fail(h, DS_ERR_ARG, "epsilon must be >= 0");
    const ds::Geometry g = own_view(h);
    if (h->dtype == DS_F32) {
        // Vertices and edges are compared in the handle's precision.  An epsilon below the spacing of
        // f32 numbers at the size the potentials reach (costs up to 255 plus lambda * (D - 1)) cannot
        // be resolved: the masks would hang on rounding noise.  Such a run belongs on an f64 handle.
        const double reach = 255.0 + std::fabs(h->lambda) * (g.D - 1);
        const double resolution = reach * 1.1920928955078125e-07;  // 2^-23
        if (epsilon < resolution)
            return fail(h, DS_ERR_ARG,
                        "epsilon " + std::to_string(epsilon) + " is below what f32 potentials resolve here (" +
                            std::to_string(resolution) + "): use an f64 handle");
    }
    int rc = need_current(h, "ds_co_begin");
    if (rc == DS_OK) rc = use_device(h);
    if (rc != DS_OK) return rc;
    rc = co_prepare(h, g, epsilon);
    if (rc != DS_OK) return rc;
    const dim3 grid(h->grid_x, g.Hb), block(ds::kBlock);
    for (int colour = 0; colour < 2; ++colour) {
        rc = dispatch(h, [&](auto t, auto lpp) -> int {
            using T = decltype(t);
            hipLaunchKernelGGL((ds::co_init_kernel<T, decltype(lpp)::value>), grid, block, 0, h->stream, g,
                               static_cast<const T *>(h->phi), own_cost(h), (T)h->lambda, (T)epsilon, colour,
                               h->seg_pixels, h->co_alive, static_cast<T *>(h->co_gmin));
            return DS_OK;
        });
        if (rc != DS_OK) return rc;
        HIPCHK(h, hipGetLastError());
    }
    h->co_begun = true;
    return DS_OK;
}

// The dirty flags a whole-band relaxation pass reads and writes (co_block_dirty in kernels.hpp), or
// two null pointers where they do not apply: a band with neighbours gets mask rows from them between
// passes, which no flag of its own records (DS_CO_DIRTY=off: nowhere, for checking).
static int co_dirty_flags(ds_handle *h, const uint8_t **prev, uint8_t **cur)
{
    *prev = nullptr;
    *cur = nullptr;
    const ds::Geometry g = own_view(h);
    if (g.row0 != 0 || g.Hb != g.H) return DS_OK;
    if (const char *e = std::getenv("DS_CO_DIRTY"))
        if (std::strcmp(e, "off") == 0) return DS_OK;
    const size_t half = (size_t)g.Hb * h->grid_x;
    uint8_t *p = h->co_dirty + (size_t)h->co_prev * half, *c = h->co_dirty + (size_t)(1 - h->co_prev) * half;
    if (h->co_dirty_all) {
        HIPCHK(h, hipMemsetAsync(p, 1, half, h->stream));
        h->co_passes_done = 0;  // every block has work again
    }
    HIPCHK(h, hipMemsetAsync(c, 0, half, h->stream));
    h->co_dirty_all = false;
    h->co_prev = 1 - h->co_prev;  // what this pass writes is what the next one reads
    *prev = p;
    *cur = c;
    return DS_OK;
}

// One relaxation pass, queued on the handle's stream; the two flags (a vertex died / a pixel
// has no vertex left) end up in h->co_flags on the device.  The pass covers the band, or, with
// radius >= 0, the window of pixels within `radius` rows and columns of pixel (wi, wj) (rounded
// out to whole blocks).
static int co_launch_pass(ds_handle *h, int radius = -1, int wi = 0, int wj = 0)
{
    const ds::Geometry g = own_view(h);
    int row_first = g.row0, rows = g.Hb, seg_first = 0, segs = h->grid_x, col_lo = 0, col_hi = g.W - 1;
    if (radius >= 0) {
        const int r0 = wi - radius > g.row0 ? wi - radius : g.row0;
        const int r1 = wi + radius < g.row0 + g.Hb - 1 ? wi + radius : g.row0 + g.Hb - 1;
        const int c0 = wj - radius > 0 ? wj - radius : 0, c1 = wj + radius < g.W - 1 ? wj + radius : g.W - 1;
        row_first = r0;
        rows = r1 - r0 + 1;
        col_lo = c0;
        col_hi = c1;
        seg_first = (c0 >> 1) / h->seg_pixels;
        segs = (c1 >> 1) / h->seg_pixels - seg_first + 1;
    }
    HIPCHK(h, hipMemsetAsync(h->co_flags, 0, 2 * sizeof(int), h->stream));
    const uint8_t *dirty_prev = nullptr;
    uint8_t *dirty_cur = nullptr;
    if (radius >= 0) h->co_dirty_all = true;  // a windowed pass keeps no flags: the next whole pass looks everywhere
    else if (co_dirty_flags(h, &dirty_prev, &dirty_cur) != DS_OK) return DS_ERR_HIP;
    // with dirty flags most blocks of a late pass return at once: fewer, longer blocks then (32
    // segments each, their flags looked up by 32 lanes at once); the first passes, where every block
    // has work, keep one segment per block
    const int per_block = dirty_prev && h->co_passes_done >= 2 && (long long)segs * rows >= 1024 ? 32 : 1;
    if (radius < 0) ++h->co_passes_done;
    const dim3 grid((segs + per_block - 1) / per_block, rows), block(ds::kBlock);
    for (int colour = 0; colour < 2; ++colour) {
        int rc = dispatch(h, [&](auto t, auto lpp) -> int {
            using T = decltype(t);
            hipLaunchKernelGGL((ds::co_relax_kernel<T, decltype(lpp)::value>), grid, block, 0, h->stream, g,
                               static_cast<const T *>(h->phi), (T)h->lambda, (T)h->co_epsilon, colour,
                               h->seg_pixels, h->co_alive, static_cast<const T *>(h->co_gmin), h->co_flags, row_first,
                               seg_first, col_lo, col_hi, dirty_prev, dirty_cur, h->grid_x, per_block, seg_first + segs);
            return DS_OK;
        });
        if (rc != DS_OK) return rc;
        HIPCHK(h, hipGetLastError());
    }
    return DS_OK;
}

int ds_co_pass(ds_handle *h, int *changed, int *lost)
{
    if (!h) return DS_ERR_ARG;
    if (!changed || !lost) return fail(h, DS_ERR_ARG, "changed / lost is NULL");
    if (!h->co_begun) return fail(h, DS_ERR_STATE, "ds_co_pass before ds_co_begin");
    int rc = use_device(h);
    if (rc == DS_OK) rc = co_launch_pass(h);
    if (rc != DS_OK) return rc;
    int host_flags[2] = {0, 0};
    HIPCHK(h, hipMemcpyAsync(host_flags, h->co_flags, 2 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    *changed = host_flags[0];
    *lost = host_flags[1];
    h->co_fixed = !host_flags[0] || host_flags[1];
    return DS_OK;
}

int ds_co_halo(ds_handle *h, int colour, int side, int which, void **dev_ptr, size_t *bytes)
{
    if (!h) return DS_ERR_ARG;
    if ((colour | 1) != 1 || (side | 1) != 1 || (which | 1) != 1 || !dev_ptr || !bytes)
        return fail(h, DS_ERR_ARG, "bad ds_co_halo argument");
    if (!h->co_begun) return fail(h, DS_ERR_STATE, "ds_co_halo before ds_co_begin");
    const ds::Geometry g = own_view(h);
    // rows of the mask buffer: 0 = halo above, 1..Hb = the band, Hb + 1 = halo below
    const int il = side == 0 ? (which == 0 ? 1 : 0) : (which == 0 ? g.Hb : g.Hb + 1);
    *dev_ptr = h->co_alive + colour * ds::mask_colour_stride(g) + (long long)il * g.Wh * g.Dpad;
    *bytes = (size_t)g.Wh * g.Dpad;
    return DS_OK;
}

int ds_co_finish(ds_handle *h, int empty, uint32_t *labels, uint8_t *alive_out, int *not_empty)
{
    if (!h) return DS_ERR_ARG;
    if (!h->co_begun) return fail(h, DS_ERR_STATE, "ds_co_finish before ds_co_begin");
    const ds::Geometry g = own_view(h);
    int rc = use_device(h);
    if (rc != DS_OK) return rc;
    if (alive_out && !h->co_host_order) {
        hipError_t e = hipMalloc(&h->co_host_order, (size_t)g.Hb * g.W * g.D);
        if (e != hipSuccess)
            return fail(h, e == hipErrorOutOfMemory ? DS_ERR_NOMEM : DS_ERR_HIP,
                        std::string("hipMalloc(mask copy): ") + hipGetErrorString(e));
    }
    const dim3 grid(h->grid_x, g.Hb), block(ds::kBlock);
    if (empty) HIPCHK(h, hipMemsetAsync(h->co_alive, 0, (size_t)(2 * ds::mask_colour_stride(g)), h->stream));
    HIPCHK(h, hipMemsetAsync(h->co_flags, 0, 2 * sizeof(int), h->stream));
    for (int colour = 0; colour < 2; ++colour) {
        switch (h->lpp) {
        case 16: hipLaunchKernelGGL(ds::co_labels_kernel<16>, grid, block, 0, h->stream, g, h->co_alive, colour, h->seg_pixels, h->d_labels, h->co_flags + 1); break;
        case 32: hipLaunchKernelGGL(ds::co_labels_kernel<32>, grid, block, 0, h->stream, g, h->co_alive, colour, h->seg_pixels, h->d_labels, h->co_flags + 1); break;
        default: hipLaunchKernelGGL(ds::co_labels_kernel<64>, grid, block, 0, h->stream, g, h->co_alive, colour, h->seg_pixels, h->d_labels, h->co_flags + 1); break;
        }
        HIPCHK(h, hipGetLastError());
    }
    int host_flags[2] = {0, 0};
    HIPCHK(h, hipMemcpyAsync(host_flags, h->co_flags, 2 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    if (labels)
        HIPCHK(h, hipMemcpyAsync(labels, h->d_labels, (size_t)g.Hb * g.W * sizeof(uint32_t), hipMemcpyDeviceToHost,
                                 h->stream));
    if (alive_out) {
        hipLaunchKernelGGL(ds::co_unpack_kernel, dim3(4096), block, 0, h->stream, g, h->co_alive, h->co_host_order);
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipMemcpyAsync(alive_out, h->co_host_order, (size_t)g.Hb * g.W * g.D, hipMemcpyDeviceToHost,
                                 h->stream));
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (not_empty) *not_empty = host_flags[1] == 0;
    return DS_OK;
}

int ds_crossing_out(ds_handle *h, double epsilon, int max_passes, uint32_t *labels, uint8_t *alive_out,
                    int *not_empty, int *passes)
{
    if (!h) return DS_ERR_ARG;
    if (max_passes < 1) return fail(h, DS_ERR_ARG, "max_passes must be >= 1");
    const ds::Geometry &g = h->g;
    if (g.row0 != 0 || g.Hb != g.H)
        return fail(h, DS_ERR_STATE,
                    "ds_crossing_out needs the whole image; bands use ds_co_begin / ds_co_pass / ds_co_halo / ds_co_finish");
    int rc = ds_co_begin(h, epsilon);
    if (rc != DS_OK) return rc;
    int done = 0, changed = 1, lost = 0;
    while (done < max_passes && changed && !lost) {
        rc = ds_co_pass(h, &changed, &lost);
        if (rc != DS_OK) return rc;
        ++done;
    }
    if (changed && !lost) return fail(h, DS_ERR_STATE, "crossing out did not reach its fixed point within max_passes");
    // A pixel without alive vertices takes all support from its neighbours, they from theirs,
    // and the pixel grid is connected: the greatest consistent sub-graph is empty.  That ends
    // the loop early; ds_co_finish clears the masks instead of the passes that would have.
    rc = ds_co_finish(h, lost, labels, alive_out, not_empty);
    if (rc != DS_OK) return rc;
    if (passes) *passes = done;
    return DS_OK;
}

int ds_find_best_labeling(ds_handle *h, int max_passes, uint32_t *labels, int *not_empty, long long *choices)
{
    if (!h) return DS_ERR_ARG;
    if (!labels || !not_empty) return fail(h, DS_ERR_ARG, "labels / not_empty is NULL");
    if (max_passes < 1) return fail(h, DS_ERR_ARG, "max_passes must be >= 1");
    const ds::Geometry &g = h->g;
    if (g.row0 != 0 || g.Hb != g.H) return fail(h, DS_ERR_STATE, "ds_find_best_labeling needs the whole image");
    if (!h->co_begun)
        return fail(h, DS_ERR_STATE, "ds_find_best_labeling works on the masks of ds_crossing_out (or ds_co_begin + ds_co_pass)");
    int rc = use_device(h);
    if (rc != DS_OK) return rc;
    if (!h->co_state) {
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&h->co_state), 2 * sizeof(long long));
        if (e != hipSuccess) return fail_hip(h, "hipMalloc(labeling state)", e);
    }
    const long long total = (long long)g.H * g.W;
    long long chosen = 0;
    int ok = 1;
    // The windows below follow one choice's consequences outwards; that equals whole-image passes
    // only if nothing else is pending.  Masks that a caller stepped with ds_co_pass and left short
    // of their fixed point are relaxed over the whole image first.
    for (int done = 0; !h->co_fixed; ++done) {
        if (done >= max_passes)
            return fail(h, DS_ERR_STATE, "crossing out did not reach its fixed point within max_passes");
        int changed = 0, lost = 0;
        rc = ds_co_pass(h, &changed, &lost);
        if (rc != DS_OK) return rc;
    }
    const char *wenv = std::getenv("DS_LABELING_WINDOW");
    const bool whole_passes = wenv && std::strcmp(wenv, "off") == 0;  // checking knob: whole-image passes
    const dim3 grid(h->grid_x, g.Hb), block(ds::kBlock);
    // Masks only shrink, so the pixels that can still have a choice when their turn comes are among
    // those with several labels now: count once, and visit only those, in raster order.
    std::vector<uint32_t> counts((size_t)total);
    for (int colour = 0; colour < 2; ++colour) {
        switch (h->lpp) {
        case 16: hipLaunchKernelGGL(ds::co_count_kernel<16>, grid, block, 0, h->stream, g, h->co_alive, colour, h->seg_pixels, h->d_labels); break;
        case 32: hipLaunchKernelGGL(ds::co_count_kernel<32>, grid, block, 0, h->stream, g, h->co_alive, colour, h->seg_pixels, h->d_labels); break;
        default: hipLaunchKernelGGL(ds::co_count_kernel<64>, grid, block, 0, h->stream, g, h->co_alive, colour, h->seg_pixels, h->d_labels); break;
        }
        HIPCHK(h, hipGetLastError());
    }
    HIPCHK(h, hipMemcpyAsync(counts.data(), h->d_labels, (size_t)total * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (long long idx = 0; idx < total && ok; ++idx) {
        if (counts[(size_t)idx] == 0) ok = 0;  // the reference's "Disparity ... wasn't found" at its first pixel without labels
        if (counts[(size_t)idx] < 2 || !ok) continue;
        // this pixel alone: one label left by now -> nothing to do; several -> the one with the
        // smallest f_t(d) stays (co_fix_kernel)
        if (h->dtype == DS_F64)
            hipLaunchKernelGGL(ds::co_fix_kernel<double>, dim3(1), dim3(64), 0, h->stream, g,
                               static_cast<const double *>(h->phi), h->cost_px, h->co_alive, idx, 1LL, h->d_labels,
                               h->co_state);
        else
            hipLaunchKernelGGL(ds::co_fix_kernel<float>, dim3(1), dim3(64), 0, h->stream, g,
                               static_cast<const float *>(h->phi), h->cost_px, h->co_alive, idx, 1LL, h->d_labels,
                               h->co_state);
        HIPCHK(h, hipGetLastError());
        long long st[2] = {0, 0};
        HIPCHK(h, hipMemcpyAsync(st, h->co_state, sizeof st, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (st[0] == 2) {  // no label left at this pixel
            ok = 0;
            break;
        }
        if (st[0] == 1) {  // labels were crossed out: back to the fixed point (crossing_out, :135-183)
            ++chosen;
            h->co_dirty_all = true;  // a mask changed outside the relaxation passes
            // Only the chosen pixel's mask changed, and a deletion can only follow a deletion at a
            // neighbour: pass k need not look further than k pixels from the chosen one (and changes
            // nothing outside that window), and a pass that deletes nothing has reached the fixed
            // point: no pixel beyond its window has a changed neighbour either.  The same masks as
            // whole-image passes, at the cost of the window instead of the image.
            const int pi = (int)(idx / g.W), pj = (int)(idx % g.W);
            int changed = 1, lost = 0, done = 0;
            while (changed && !lost && done < max_passes) {
                rc = whole_passes ? co_launch_pass(h) : co_launch_pass(h, done + 1, pi, pj);
                if (rc != DS_OK) return rc;
                int flags[2] = {0, 0};
                HIPCHK(h, hipMemcpyAsync(flags, h->co_flags, sizeof flags, hipMemcpyDeviceToHost, h->stream));
                HIPCHK(h, hipStreamSynchronize(h->stream));
                changed = flags[0];
                lost = flags[1];
                ++done;
            }
            if (changed && !lost)
                return fail(h, DS_ERR_STATE, "crossing out did not reach its fixed point within max_passes");
            if (lost) {    // the reference's assert!(is_not_empty)
                ok = 0;
                break;
            }
        }
    }
    // every pixel has one label left (or the run ended early): the smallest alive label per pixel
    HIPCHK(h, hipMemsetAsync(h->co_flags, 0, 2 * sizeof(int), h->stream));
    for (int colour = 0; colour < 2; ++colour) {
        switch (h->lpp) {
        case 16: hipLaunchKernelGGL(ds::co_labels_kernel<16>, grid, block, 0, h->stream, g, h->co_alive, colour, h->seg_pixels, h->d_labels, h->co_flags + 1); break;
        case 32: hipLaunchKernelGGL(ds::co_labels_kernel<32>, grid, block, 0, h->stream, g, h->co_alive, colour, h->seg_pixels, h->d_labels, h->co_flags + 1); break;
        default: hipLaunchKernelGGL(ds::co_labels_kernel<64>, grid, block, 0, h->stream, g, h->co_alive, colour, h->seg_pixels, h->d_labels, h->co_flags + 1); break;
        }
        HIPCHK(h, hipGetLastError());
    }
    HIPCHK(h, hipMemcpyAsync(labels, h->d_labels, (size_t)total * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    *not_empty = ok;
    if (choices) *choices = chosen;
    return DS_OK;
}

static void eps_release(ds_handle *h)
{
    (void)hipFree(h->eps_buf[0]);
    (void)hipFree(h->eps_buf[1]);
    h->eps_buf[0] = h->eps_buf[1] = h->eps_sorted = nullptr;
    h->eps_count = 0;
}

int ds_epsilon_gaps(ds_handle *h, size_t *count, int *largest_repeated)
{
    if (!h) return DS_ERR_ARG;
    if (!count) return fail(h, DS_ERR_ARG, "count is NULL");
    const ds::Geometry &g = h->g;
    if (g.row0 != 0 || g.Hb != g.H) return fail(h, DS_ERR_STATE, "ds_epsilon_gaps needs the whole image");
    int rc = need_current(h, "ds_epsilon_gaps");
    if (rc == DS_OK) rc = use_device(h);
    if (rc != DS_OK) return rc;
    eps_release(h);
    const size_t n = (size_t)g.H * g.W * g.D * (1 + 4 * (size_t)g.D);
    size_t free_b = 0, total_b = 0;
    HIPCHK(h, hipMemGetInfo(&free_b, &total_b));
    if (n > free_b / (3 * sizeof(double)))  // two buffers + the sort's work space
        return fail(h, DS_ERR_NOMEM, "the array of gaps (H*W*D*(1+4D) doubles, three times over) does not fit this GPU");
    const size_t lds = (9 * (size_t)g.D + ds::kBlock) * sizeof(double);
    for (int b = 0; b < 2; ++b) {
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&h->eps_buf[b]), n * sizeof(double));
        if (e != hipSuccess) {
            eps_release(h);
            return fail_hip(h, "hipMalloc (epsilon gaps)", e);
        }
    }
    const dim3 grid((unsigned)((size_t)g.H * g.W)), block(ds::kBlock);
    if (h->dtype == DS_F64)
        hipLaunchKernelGGL(ds::eps_gaps_kernel<double>, grid, block, lds, h->stream, g,
                           static_cast<const double *>(h->phi), h->cost_px, h->lambda, h->eps_buf[0]);
    else
        hipLaunchKernelGGL(ds::eps_gaps_kernel<float>, grid, block, lds, h->stream, g,
                           static_cast<const float *>(h->phi), h->cost_px, h->lambda, h->eps_buf[0]);
    hipError_t e = hipGetLastError();
    size_t unique = 0;
    if (e == hipSuccess) e = ds::sort_unique_f64(h->eps_buf[0], h->eps_buf[1], n, &h->eps_sorted, &unique, h->stream);
    if (e != hipSuccess) {
        eps_release(h);
        return fail_hip(h, "epsilon gaps", e);
    }
    // entries of vertices / edges that do not exist were written as +inf: one survives, last
    if (unique > 0) {
        double last = 0.;
        HIPCHK(h, hipMemcpy(&last, h->eps_sorted + (unique - 1), sizeof last, hipMemcpyDeviceToHost));
        if (std::isinf(last)) --unique;
    }
    // Does the largest gap occur more than once?  (The reference's dedup repeats the last
    // element of its array in that case.)  The other buffer still holds all n values sorted:
    // find where the +inf entries begin, by bisection, and look at the two values before.
    int repeated = 0;
    {
        const double *sorted_all = h->eps_sorted == h->eps_buf[0] ? h->eps_buf[1] : h->eps_buf[0];
        size_t lo = 0, hi = n;  // first index holding +inf lies in [lo, hi]
        while (lo < hi) {
            const size_t mid = lo + (hi - lo) / 2;
            double v = 0.;
            HIPCHK(h, hipMemcpy(&v, sorted_all + mid, sizeof v, hipMemcpyDeviceToHost));
            if (std::isinf(v)) hi = mid;
            else lo = mid + 1;
        }
        if (lo >= 2) {
            double pair[2] = {0., 0.};
            HIPCHK(h, hipMemcpy(pair, sorted_all + (lo - 2), sizeof pair, hipMemcpyDeviceToHost));
            repeated = pair[0] == pair[1];
        }
    }
    if (largest_repeated) *largest_repeated = repeated;
    h->eps_count = unique;
    *count = unique;
    return DS_OK;
}

int ds_epsilon_gaps_fetch(ds_handle *h, double *out)
{
    if (!h) return DS_ERR_ARG;
    if (!h->eps_sorted) return fail(h, DS_ERR_STATE, "ds_epsilon_gaps_fetch without ds_epsilon_gaps");
    if (!out && h->eps_count) return fail(h, DS_ERR_ARG, "out is NULL");
    int rc = use_device(h);
    if (rc != DS_OK) return rc;
    if (h->eps_count)
        HIPCHK(h, hipMemcpy(out, h->eps_sorted, h->eps_count * sizeof(double), hipMemcpyDeviceToHost));
    eps_release(h);
    return DS_OK;
}

// One pass of eps_query_kernel over the whole image.  mins: H * W * 5 doubles (device), result: 3 x u64
// (device).  out[0] = smallest gap > lo (+inf: none), out[1] = largest gap, out[2] = gaps <= lo.
static int eps_query(ds_handle *h, double lo, double *mins, int have_mins, unsigned long long *result, double out[3])
{
    const ds::Geometry &g = h->g;
    static const unsigned long long init[3] = {0x7ff0000000000000ull, 0ull, 0ull};  // +inf, +0, 0
    HIPCHK(h, hipMemcpyAsync(result, init, sizeof init, hipMemcpyHostToDevice, h->stream));
    int shift = 0;
    while ((1 << shift) < g.D) ++shift;
    const size_t lds = 9 * (size_t)g.D * sizeof(double);
    const dim3 grid((unsigned)((size_t)g.H * g.W)), block(ds::kBlock);
    if (h->dtype == DS_F64)
        hipLaunchKernelGGL(ds::eps_query_kernel<double>, grid, block, lds, h->stream, g, static_cast<const double *>(h->phi),
                           h->cost_px, h->lambda, lo, shift, mins, have_mins, result);
    else
        hipLaunchKernelGGL(ds::eps_query_kernel<float>, grid, block, lds, h->stream, g, static_cast<const float *>(h->phi),
                           h->cost_px, h->lambda, lo, shift, mins, have_mins, result);
    HIPCHK(h, hipGetLastError());
    unsigned long long host[3] = {0, 0, 0};
    HIPCHK(h, hipMemcpyAsync(host, result, sizeof host, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    std::memcpy(&out[0], &host[0], sizeof(double));
    std::memcpy(&out[1], &host[1], sizeof(double));
    out[2] = (double)host[2];
    return DS_OK;
}

static int eps_query_buffers(ds_handle *h, double **mins, unsigned long long **result)
{
    const ds::Geometry &g = h->g;
    *mins = nullptr;
    *result = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(mins), (size_t)g.H * g.W * 5 * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(result), 3 * sizeof(unsigned long long));
    if (e != hipSuccess) {
        (void)hipFree(*mins);
        (void)hipFree(*result);
        (void)hipGetLastError();
        return fail_hip(h, "hipMalloc (epsilon query)", e);
    }
    return DS_OK;
}

// ds_crossing_out with the reference's own comparisons (co_exact_kernel); mins: the vertex and pair
// minima of the current phi (eps_query with have_mins = 0 fills them).
static int crossing_out_exact(ds_handle *h, double epsilon, const double *mins, int max_passes, uint32_t *labels,
                              uint8_t *alive_out, int *not_empty, int *passes)
{
    const ds::Geometry &g = h->g;
    int rc = co_prepare(h, g, epsilon);
    if (rc != DS_OK) return rc;
    // the O(D) kernels' pair minima as well, so that every entry point that continues on these
    // masks (ds_find_best_labeling, ds_co_pass) finds the state it expects
    const dim3 grid_o(h->grid_x, g.Hb), block(ds::kBlock);
    for (int colour = 0; colour < 2; ++colour) {
        rc = dispatch(h, [&](auto t, auto lpp) -> int {
            using T = decltype(t);
            hipLaunchKernelGGL((ds::co_init_kernel<T, decltype(lpp)::value>), grid_o, block, 0, h->stream, g,
                               static_cast<const T *>(h->phi), h->cost_px, (T)h->lambda, (T)epsilon, colour,
                               h->seg_pixels, h->co_alive, static_cast<T *>(h->co_gmin));
            return DS_OK;
        });
        if (rc != DS_OK) return rc;
        HIPCHK(h, hipGetLastError());
    }
    int shift = 0;
    while ((1 << shift) < g.D) ++shift;
    const size_t lds = (size_t)g.D * (8 * sizeof(double) + sizeof(int) + 5) + 8;
    // per pixel: the last pass in which it lost labels and the last pass in which it was looked at
    // (co_exact_kernel looks at a pixel again only if a neighbour lost labels in that pass or later)
    const size_t pixels = (size_t)g.H * g.W;
    if (!h->co_version) {
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&h->co_version), 2 * pixels * sizeof(int));
        if (e != hipSuccess) return fail_hip(h, "hipMalloc (crossing-out versions)", e);
    }
    HIPCHK(h, hipMemsetAsync(h->co_version, 0, 2 * pixels * sizeof(int), h->stream));
    const char *dirty_env = std::getenv("DS_CO_DIRTY");
    int *version = dirty_env && std::strcmp(dirty_env, "off") == 0 ? nullptr : h->co_version;
    int *seen = version ? version + pixels : nullptr;
    int passes_done = 0;
    auto launch = [&](int phase) -> int {
        const uint8_t *dirty_prev = nullptr;
        uint8_t *dirty_cur = nullptr;
        if (phase == 1 && co_dirty_flags(h, &dirty_prev, &dirty_cur) != DS_OK) return DS_ERR_HIP;
        // from the third pass on most segments have nothing to do: 32 of them per block then
        const int per_block = phase == 1 && dirty_prev && passes_done >= 2 ? 32 : 1;
        if (phase == 1) ++passes_done;
        const dim3 grid((h->grid_x + per_block - 1) / per_block, g.Hb);
        if (h->dtype == DS_F64)
            hipLaunchKernelGGL(ds::co_exact_kernel<double>, grid, block, lds, h->stream, g, static_cast<const double *>(h->phi),
                               h->cost_px, h->lambda, epsilon, shift, mins, h->co_alive, h->co_flags, phase, dirty_prev,
                               dirty_cur, h->seg_pixels, h->grid_x, version, seen, per_block, passes_done);
        else
            hipLaunchKernelGGL(ds::co_exact_kernel<float>, grid, block, lds, h->stream, g, static_cast<const float *>(h->phi),
                               h->cost_px, h->lambda, epsilon, shift, mins, h->co_alive, h->co_flags, phase, dirty_prev,
                               dirty_cur, h->seg_pixels, h->grid_x, version, seen, per_block, passes_done);
        HIPCHK(h, hipGetLastError());
        return DS_OK;
    };
    rc = launch(0);
    if (rc != DS_OK) return rc;
    h->co_begun = true;
    // A pixel that has lost every label settles the verdict, and a caller that wants no masks stops
    // there.  The masks of an empty result are something else: the reference relaxes on to its fixed
    // point, where -- an edge copy whose twin was never set outlives the twin's vertex -- some
    // vertices may still stand.  A caller that asks for the masks gets those.
    const bool to_fixed_point = alive_out != nullptr;
    int done = 0, changed = 1, lost = 0;
    while (done < max_passes && changed && !(lost && !to_fixed_point)) {
        HIPCHK(h, hipMemsetAsync(h->co_flags, 0, 2 * sizeof(int), h->stream));
        rc = launch(1);
        if (rc != DS_OK) return rc;
        int host_flags[2] = {0, 0};
        HIPCHK(h, hipMemcpyAsync(host_flags, h->co_flags, 2 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        changed = host_flags[0];
        lost |= host_flags[1];
        ++done;
    }
    const bool stopped_early = lost && !to_fixed_point;
    h->co_fixed = !changed || stopped_early;
    if (changed && !stopped_early)
        return fail(h, DS_ERR_STATE, "crossing out did not reach its fixed point within max_passes");
    rc = ds_co_finish(h, stopped_early, labels, alive_out, not_empty);
    if (rc != DS_OK) return rc;
    if (passes) *passes = done;
    return DS_OK;
}

int ds_crossing_out_exact(ds_handle *h, double epsilon, int max_passes, uint32_t *labels, uint8_t *alive_out,
                          int *not_empty, int *passes)
{
    if (!h) return DS_ERR_ARG;
    if (!(epsilon >= 0.0)) return fail(h, DS_ERR_ARG, "epsilon must be >= 0");
    if (max_passes < 1) return fail(h, DS_ERR_ARG, "max_passes must be >= 1");
    const ds::Geometry &g = h->g;
    if (g.row0 != 0 || g.Hb != g.H) return fail(h, DS_ERR_STATE, "ds_crossing_out_exact needs the whole image");
    int rc = need_current(h, "ds_crossing_out_exact");
    if (rc == DS_OK) rc = use_device(h);
    if (rc != DS_OK) return rc;
    double *mins = nullptr;
    unsigned long long *result = nullptr;
    rc = eps_query_buffers(h, &mins, &result);
    if (rc != DS_OK) return rc;
    double q[3];
    rc = eps_query(h, -1.0, mins, 0, result, q);
    if (rc == DS_OK) rc = crossing_out_exact(h, epsilon, mins, max_passes, labels, alive_out, not_empty, passes);
    (void)hipFree(mins);
    (void)hipFree(result);
    return rc;
}

int ds_epsilon_next(ds_handle *h, double lo, double *next, double *largest, unsigned long long *not_above)
{
    if (!h) return DS_ERR_ARG;
    const ds::Geometry &g = h->g;
    if (g.row0 != 0 || g.Hb != g.H) return fail(h, DS_ERR_STATE, "ds_epsilon_next needs the whole image");
    int rc = need_current(h, "ds_epsilon_next");
    if (rc == DS_OK) rc = use_device(h);
    if (rc != DS_OK) return rc;
    double *mins = nullptr;
    unsigned long long *result = nullptr;
    rc = eps_query_buffers(h, &mins, &result);
    if (rc != DS_OK) return rc;
    double out[3] = {0., 0., 0.};
    rc = eps_query(h, lo, mins, 0, result, out);
    (void)hipFree(mins);
    (void)hipFree(result);
    if (rc != DS_OK) return rc;
    if (next) *next = out[0];
    if (largest) *largest = out[1];
    if (not_above) *not_above = (unsigned long long)out[2];
    return DS_OK;
}

int ds_epsilon_search(ds_handle *h, int exact, int max_passes, double *epsilon, long long *stats)
{
    if (!h) return DS_ERR_ARG;
    if (!epsilon) return fail(h, DS_ERR_ARG, "epsilon is NULL");
    if (max_passes < 1) return fail(h, DS_ERR_ARG, "max_passes must be >= 1");
    const ds::Geometry &g = h->g;
    if (g.row0 != 0 || g.Hb != g.H) return fail(h, DS_ERR_STATE, "ds_epsilon_search needs the whole image");
    if (h->dtype != DS_F64 && !exact)
        return fail(h, DS_ERR_ARG, "ds_epsilon_search on an f32 handle needs exact = 1: its candidates go down to 0, which comparisons in f32 do not resolve");
    int rc = need_current(h, "ds_epsilon_search");
    if (rc == DS_OK) rc = use_device(h);
    if (rc != DS_OK) return rc;
    double *mins = nullptr;
    unsigned long long *result = nullptr;
    rc = eps_query_buffers(h, &mins, &result);
    if (rc != DS_OK) return rc;
    long long predicates = 0, queries = 0, passes_total = 0;
    auto done = [&](int code) {
        (void)hipFree(mins);
        (void)hipFree(result);
        if (stats) {
            stats[0] = predicates;
            stats[1] = queries;
            stats[2] = passes_total;
        }
        return code;
    };
    // the predicate of epsilon_search.rs:100-102: initialize_with_epsilon, crossing_out, is_not_empty
    const bool verbose = std::getenv("DS_EPSILON_SEARCH_LOG") != nullptr;  // one line per question, on stderr
    auto ask = [&](bool by_reference, double eps, bool &yes) {
        int not_empty = 0, passes = 0;
        const auto t0 = std::chrono::steady_clock::now();
        const int r = by_reference ? crossing_out_exact(h, eps, mins, max_passes, nullptr, nullptr, &not_empty, &passes)
                                   : ds_crossing_out(h, eps, max_passes, nullptr, nullptr, &not_empty, &passes);
        ++predicates;
        passes_total += passes;
        yes = not_empty != 0;
        if (verbose)
            std::fprintf(stderr, "ds_epsilon_search: epsilon %.17g (%s comparisons): %s after %d passes, %.3f s\n", eps,
                         by_reference ? "the reference's" : "O(D)", yes ? "not empty" : "empty", passes,
                         std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        return r;
    };
    auto consistent = [&](double eps, bool &yes) { return ask(exact != 0, eps, yes); };
    double q[3];
    rc = eps_query(h, -1.0, mins, 0, result, q);  // every gap is > -1: the smallest and the largest candidate
    ++queries;
    if (rc != DS_OK) return done(rc);
    const double smallest = q[0], largest = q[1];
    // The reference searches the sorted array for the first candidate that leaves a non-empty graph and
    // never tries the last one (first_index == last_index ends its loop): if no other candidate does,
    // the largest is the answer.
    if (!(smallest < largest)) {
        *epsilon = largest;
        return done(DS_OK);
    }
    bool yes = false;
    rc = consistent(smallest, yes);
    if (rc != DS_OK) return done(rc);
    if (yes) {
        *epsilon = smallest;
        return done(DS_OK);
    }
    // Invariant: every candidate <= lo fails; the answer is the smallest candidate in (lo, hi] that
    // passes, or `largest` if none does.  The predicate only changes at candidates, so halving the
    // interval of real numbers with the predicate narrows it to a few candidates without knowing any.
    double lo = smallest, hi = largest;
    auto halve = [&](bool by_reference) -> int {
        for (int it = 0; it < 80; ++it) {
            const double mid = lo + (hi - lo) * 0.5;
            if (!(lo < mid && mid < hi)) break;
            const int r = ask(by_reference, mid, yes);
            if (r != DS_OK) return r;
            if (yes) hi = mid;
            else lo = mid;
        }
        return DS_OK;
    };
    // With the reference's comparisons a crossing-out costs O(D^2) per pixel and pass.  The O(D) one
    // answers the same unless a penalty lies within rounding of its threshold, so on f64 handles it
    // does the halving first.  Where it changes its answer is a gap of the graph moved by that
    // rounding, so the interval it ends with is widened by a bound on the rounding (a few units in
    // the last place of the largest penalty: r), the reference's comparisons confirm the wider
    // interval's two ends (r grows until they do) and finish the halving inside it: about a dozen
    // crossing-outs of the expensive kind instead of fifty.
    const bool fast_first = exact && h->dtype == DS_F64;
    rc = halve(exact && !fast_first);
    if (rc != DS_OK) return done(rc);
    if (fast_first) {
        double r = 1e-12 * (255.0 + std::fabs(h->lambda) * (g.D - 1));
        double lo_known = smallest, hi_known = largest;  // the reference's comparisons: fails at lo_known, passes at hi_known (or it is `largest`)
        const double lo_fast = lo, hi_fast = hi;
        bool lo_set = false, hi_set = false;
        for (int grow = 0; grow < 40 && !(lo_set && hi_set); ++grow, r *= 16.0) {
            if (!lo_set) {
                const double x = lo_fast - r;
                if (!(x > lo_known)) lo_set = true;  // back at a point that is known to fail
                else {
                    rc = ask(true, x, yes);
                    if (rc != DS_OK) return done(rc);
                    if (yes) hi_known = std::min(hi_known, x), hi_set = true;  // the answer lies below
                    else lo_known = x, lo_set = true;
                }
            }
            if (!hi_set) {
                const double x = hi_fast + r;
                if (!(x < hi_known)) hi_set = true;
                else {
                    rc = ask(true, x, yes);
                    if (rc != DS_OK) return done(rc);
                    if (yes) hi_known = x, hi_set = true;
                    else lo_known = std::max(lo_known, x), lo_set = true;  // the answer lies above
                }
            }
            // an end that turned out to lie on the wrong side moved the other end's known point: look again
            if (lo_set && hi_set && !(lo_known < hi_known)) return done(fail(h, DS_ERR_STATE, "ds_epsilon_search: the predicate is not monotone"));
        }
        lo = lo_known;
        hi = hi_known;
        rc = halve(true);
        if (rc != DS_OK) return done(rc);
    }
    for (int it = 0; it < 64; ++it) {
        rc = eps_query(h, lo, mins, 1, result, q);
        ++queries;
        if (rc != DS_OK) return done(rc);
        const double candidate = q[0];
        if (!(candidate < largest)) {  // nothing between lo and the largest candidate: the untested last element
            *epsilon = largest;
            return done(DS_OK);
        }
        rc = consistent(candidate, yes);
        if (rc != DS_OK) return done(rc);
        if (yes) {
            *epsilon = candidate;
            return done(DS_OK);
        }
        lo = candidate;
    }
    return done(fail(h, DS_ERR_STATE, "ds_epsilon_search: more than 64 candidates inside the last interval of the bisection"));
}

int ds_dedup_gaps(const double *sorted, size_t n, double tolerance, int largest_repeated, double *out, size_t *n_out)
{
    if (!n_out || (n && (!sorted || !out))) return DS_ERR_ARG;
    size_t kept = 0, i = 0, run_start = 0;
    while (i < n) {
        out[kept++] = sorted[i];
        run_start = i;
        size_t j = i + 1;
        while (j < n && !(std::fabs(sorted[i] - sorted[j]) > tolerance)) ++j;
        i = j;
    }
    // the reference pushes the array's last element once more when its run is longer than one
    if (n && (run_start != n - 1 || largest_repeated)) out[kept++] = sorted[n - 1];
    *n_out = kept;
    return DS_OK;
}

int ds_box_blur(ds_handle *h, const uint32_t *disparity_map, uint64_t threshold, uint64_t filter_radius,
                uint64_t normalization_term, uint64_t *out)
{
    if (!h) return DS_ERR_ARG;
    if (!disparity_map || !out) return fail(h, DS_ERR_ARG, "disparity_map / out is NULL");
    if (normalization_term == 0) return fail(h, DS_ERR_ARG, "normalization_term is 0 (the reference divides by it)");
    const ds::Geometry &g = h->g;
    if (g.row0 != 0 || g.Hb != g.H) return fail(h, DS_ERR_STATE, "ds_box_blur needs the whole image");
    int rc = use_device(h);
    if (rc != DS_OK) return rc;
    const size_t n = (size_t)g.H * g.W;
    // the label buffer of ds_labels holds the input; the output (u64 like the reference's usize) is temporary
    unsigned long long *d_out = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&d_out), n * sizeof(unsigned long long));
    if (e != hipSuccess) return fail(h, e == hipErrorOutOfMemory ? DS_ERR_NOMEM : DS_ERR_HIP, std::string("hipMalloc(box blur): ") + hipGetErrorString(e));
    auto done = [&](int code) {
        (void)hipFree(d_out);
        return code;
    };
    e = hipMemcpyAsync(h->d_labels, disparity_map, n * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream);
    if (e != hipSuccess) return done(fail_hip(h, "hipMemcpyAsync(disparity map)", e));
    hipLaunchKernelGGL(ds::box_blur_kernel, dim3((unsigned)((n + ds::kBlock - 1) / ds::kBlock)), dim3(ds::kBlock), 0,
                       h->stream, h->d_labels, g.H, g.W, (unsigned long long)threshold, (unsigned long long)filter_radius,
                       (unsigned long long)normalization_term, d_out);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, n * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return done(fail_hip(h, "box blur", e));
    return done(DS_OK);
}

int ds_placement_info(const ds_handle *h, int *tries, double *probe_ms, int *split)
{
    if (!h) return DS_ERR_ARG;
    if (tries) *tries = h->placement_tries;
    if (probe_ms) *probe_ms = h->placement_ms;
    if (split) *split = h->phi_b ? 1 : 0;
    return DS_OK;
}

int ds_checksum(ds_handle *h, uint64_t *out)
{
    if (!h) return DS_ERR_ARG;
    if (!out) return fail(h, DS_ERR_ARG, "out is NULL");
    int rc = need_current(h, "ds_checksum");
    if (rc == DS_OK) rc = use_device(h);
    if (rc != DS_OK) return rc;
    HIPCHK(h, hipMemsetAsync(h->d_checksum, 0, sizeof(unsigned long long), h->stream));
    long long blocks = (8 * h->plane_elems + ds::kBlock - 1) / ds::kBlock;
    if (blocks > 16384) blocks = 16384;
    const ds::Geometry own = own_view(h);
    if (h->dtype == DS_F64)
        hipLaunchKernelGGL(ds::checksum_kernel<double>, dim3((unsigned)blocks), dim3(ds::kBlock), 0,
                           h->stream, own, static_cast<const double *>(h->phi), h->d_checksum);
    else
        hipLaunchKernelGGL(ds::checksum_kernel<float>, dim3((unsigned)blocks), dim3(ds::kBlock), 0,
                           h->stream, own, static_cast<const float *>(h->phi), h->d_checksum);
    HIPCHK(h, hipGetLastError());
    unsigned long long v = 0;
    HIPCHK(h, hipMemcpyAsync(&v, h->d_checksum, sizeof v, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    *out = v;
    return DS_OK;
}

}  // extern "C"

#include "bands.hpp"
