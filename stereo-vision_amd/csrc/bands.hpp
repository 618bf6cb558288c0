// Row bands over several ranks, one process per GPU: the protocol that keeps the bands' halo
// rows current while they iterate, the RCCL transport it uses by default, and the C entry
// points for both (include/diffstereo.h, "Row bands over the GPUs of a node").
//
// Included at the end of diffstereo.hip: it uses that file's handle and launch helpers.
//
// The reference is one process with one raster loop (diffusion_stereo/src/diffusion_graph.rs:
// 264-293).  Under the red-black schedule a pixel update reads only the four neighbours of
// the other colour, so after each colour a band needs exactly one row from each neighbouring
// band: the slot-3 (downward) messages of the row above it and the slot-1 (upward) messages
// of the row below it.
//
// Streams.  Everything a band does for its neighbours -- the half-sweeps of its boundary rows
// and the two exchanges of an iteration -- is queued on `side_stream`; the kernel that
// updates the interior rows runs on the handle's stream.  The two are ordered by two events
// only: the boundary chain of iteration n waits for the interior kernel of iteration n - 1
// (ev_main) and the interior kernel of iteration n + 1 waits for the chain of iteration n
// (ev_side).  A two-iteration step adds `frame_stream` for the frame's launches, which no
// exchange waits for (band_fused2_step), tied to the boundary stream by two more events and
// joined into the handle's stream at the end of the step.  All streams have the default
// priority.  RCCL's kernel is queued on the boundary stream itself; every wait is for work
// queued earlier in program order, so no cycle of waits exists.
#pragma once
#include <chrono>
#include <cstdio>
#include <dlfcn.h>
#include <rccl/rccl.h>

struct ds_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    std::string err;
};

namespace {

// ---- librccl, loaded on first use ------------------------------------------------------
struct RcclApi {
    void *dl = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                              hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};

std::mutex g_rccl_mutex;
RcclApi g_rccl;
thread_local std::string g_comm_error;

// A copy that the process has loaded already (a host that also uses RCCL through another
// library) is taken before a second one is mapped: two copies would each keep their own state.
RcclApi *rccl()
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.dl) return &g_rccl;
    // DS_RCCL_LIBRARY names the one file to load instead (another build of RCCL; a name that does
    // not exist makes every ds_comm_* call fail cleanly, which is how the tests reach that path)
    const char *override_name = std::getenv("DS_RCCL_LIBRARY");
    const char *defaults[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    std::vector<const char *> names;
    if (override_name && *override_name) names.push_back(override_name);
    else names.assign(std::begin(defaults), std::end(defaults));
    void *dl = nullptr;
    for (const char *n : names)
        if ((dl = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL))) break;
    if (!dl)
        for (const char *n : names)
            if ((dl = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!dl) {
        const char *why = dlerror();  // one call: it returns the text once and clears it
        g_rccl.err = std::string("librccl not found: ") + (why ? why : "dlopen failed");
        return nullptr;
    }
    auto sym = [&](const char *name) {
        void *p = dlsym(dl, name);
        if (!p) g_rccl.err = std::string("librccl lacks ") + name;
        return p;
    };
    RcclApi a;
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(sym("ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(sym("ncclCommInitRank"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
    a.Send = reinterpret_cast<decltype(a.Send)>(sym("ncclSend"));
    a.Recv = reinterpret_cast<decltype(a.Recv)>(sym("ncclRecv"));
    a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(sym("ncclAllReduce"));
    a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(sym("ncclGroupStart"));
    a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(sym("ncclGroupEnd"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.Send || !a.Recv || !a.AllReduce ||
        !a.GroupStart || !a.GroupEnd || !a.GetErrorString) {
        dlclose(dl);
        return nullptr;
    }
    a.dl = dl;
    g_rccl = a;
    return &g_rccl;
}

int comm_fail(ds_comm *c, const char *what, ncclResult_t r)
{
    RcclApi *api = rccl();
    const std::string msg = std::string(what) + ": " + (api ? api->GetErrorString(r) : "librccl not loaded");
    if (c) c->err = msg;
    else g_comm_error = msg;
    return 1;
}

// ---- the RCCL transport ----------------------------------------------------------------
int rccl_exchange(void *ctx, int n, const int *peers, void *const *send, void *const *recv, const size_t *bytes,
                  void *hip_stream)
{
    ds_comm *c = static_cast<ds_comm *>(ctx);
    RcclApi *api = rccl();
    if (!api || !c || !c->comm) return 1;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    ncclResult_t r = api->GroupStart();
    if (r != ncclSuccess) return comm_fail(c, "ncclGroupStart", r);
    ncclResult_t first = ncclSuccess;
    const char *what = "";
    for (int k = 0; k < n && first == ncclSuccess; ++k) {
        first = api->Send(send[k], bytes[k], ncclUint8, peers[k], c->comm, s);
        what = "ncclSend";
        if (first != ncclSuccess) break;
        first = api->Recv(recv[k], bytes[k], ncclUint8, peers[k], c->comm, s);
        what = "ncclRecv";
    }
    r = api->GroupEnd();  // always closed, also after a failure inside the group
    if (first != ncclSuccess) return comm_fail(c, what, first);
    if (r != ncclSuccess) return comm_fail(c, "ncclGroupEnd", r);
    return 0;
}

int rccl_all_reduce(void *ctx, void *dev, int count, int kind, void *hip_stream)
{
    ds_comm *c = static_cast<ds_comm *>(ctx);
    RcclApi *api = rccl();
    if (!api || !c || !c->comm) return 1;
    ncclDataType_t type = ncclFloat64;
    ncclRedOp_t op = ncclSum;
    switch (kind) {
    case DS_REDUCE_SUM_F64: break;
    case DS_REDUCE_MAX_F64: op = ncclMax; break;
    case DS_REDUCE_MAX_I32: type = ncclInt32; op = ncclMax; break;
    case DS_REDUCE_MIN_I32: type = ncclInt32; op = ncclMin; break;
    default: c->err = "unknown reduction kind"; return 1;
    }
    ncclResult_t r = api->AllReduce(dev, dev, (size_t)count, type, op, c->comm, static_cast<hipStream_t>(hip_stream));
    if (r != ncclSuccess) return comm_fail(c, "ncclAllReduce", r);
    return 0;
}

const char *rccl_last_error(void *ctx) { return ctx ? static_cast<ds_comm *>(ctx)->err.c_str() : ""; }

// ---- the protocol ------------------------------------------------------------------------
int transport_fail(ds_handle *h, const char *what)
{
    const char *text = h->transport.last_error ? h->transport.last_error(h->transport.ctx) : nullptr;
    return fail(h, DS_ERR_STATE, std::string(what) + " failed in the transport" + (text && *text ? std::string(": ") + text : ""));
}

int need_attached(ds_handle *h, const char *who)
{
    if (!h->attached) return fail(h, DS_ERR_STATE, std::string(who) + " before ds_band_attach");
    return DS_OK;
}

// Halo row of phi: device pointer and size (see ds_halo).  planes: the four plane offsets of
// the set the row belongs to (a colour's current planes, or the set an iteration is producing).
void halo_row(const ds_handle *h, const long long *planes, int side, int which, void **ptr, size_t *bytes)
{
    const ds::Geometry &g = h->g;
    const int own_rows = own_end(h) - own_begin(h);
    const int slot = side == 0 ? (which == 0 ? 1 : 3) : (which == 0 ? 3 : 1);
    // rows of the stored planes: 0 = the halo row above the stored rows, ext_up + 1 = the band's first own row
    const int il = h->ext_up + (side == 0 ? (which == 0 ? 1 : 0) : (which == 0 ? own_rows : own_rows + 1));
    *ptr = static_cast<char *>(h->phi) + (size_t)(planes[slot] + (long long)il * g.row_stride) * h->elem;
    *bytes = (size_t)g.Wh * g.Dpad * h->elem;
}

// The exchange that follows a colour's half-sweep, queued on `stream`: the band's first row's
// upward messages to the rank above, its last row's downward messages to the rank below, and
// the mirror images into the halo rows.
int exchange_planes(ds_handle *h, const long long *planes, hipStream_t stream)
{
    int peers[2], n = 0;
    void *send[2], *recv[2];
    size_t bytes[2];
    for (int side = 0; side < 2; ++side) {
        const int peer = side == 0 ? h->up : h->down;
        if (peer < 0) continue;
        peers[n] = peer;
        halo_row(h, planes, side, 0, &send[n], &bytes[n]);
        halo_row(h, planes, side, 1, &recv[n], &bytes[n]);
        ++n;
    }
    if (n == 0) return DS_OK;
    if (h->transport.exchange(h->transport.ctx, n, peers, send, recv, bytes, stream) != 0)
        return transport_fail(h, "the halo exchange");
    return DS_OK;
}

// upcoming = the colour-1 planes the running iteration is producing instead of the current ones
int exchange_phi(ds_handle *h, int colour, bool upcoming, hipStream_t stream)
{
    return exchange_planes(h, upcoming ? h->alt_off : h->g.plane_off[colour], stream);
}

// One colour of a band that sweeps colour by colour: boundary rows, their exchange beside the
// interior rows.
int band_half_sweep(ds_handle *h, int colour)
{
    const int rb = h->g.row0, re = h->g.row0 + h->g.Hb;
    const int top_end = h->up >= 0 ? (rb + 1 < re ? rb + 1 : re) : rb;
    const int bot_begin = h->down >= 0 ? (re - 1 > top_end ? re - 1 : top_end) : re;
    int rc = launch_sweep(h, colour, rb, top_end, nullptr, bot_begin, re);
    if (rc == DS_OK) rc = order_streams(h, h->stream, h->ev_main, h->side_stream);
    if (rc == DS_OK) rc = exchange_phi(h, colour, false, h->side_stream);
    if (rc == DS_OK) rc = launch_sweep(h, colour, top_end, bot_begin);
    if (rc == DS_OK) rc = order_streams(h, h->side_stream, h->ev_side, h->stream);
    return rc;
}

// One iteration of a band whose interior runs in the fused kernel.  The boundary chain and the
// interior kernel read only the planes of the previous iteration and write disjoint rows.
int band_fused_iteration(ds_handle *h, int fb, int fe)
{
    const int rb = h->g.row0, re = h->g.row0 + h->g.Hb;
    // the chain reads what the previous interior kernel wrote, and writes the planes it read
    int rc = order_streams(h, h->stream, h->ev_main, h->side_stream);
    if (rc != DS_OK) return rc;
    rc = on_side_stream(h, [&]() -> int {
        const int t0 = rb, t1 = h->up >= 0 ? rb + 2 : rb;      // first colour: two rows next to each neighbour
        const int b0 = h->down >= 0 ? re - 2 : re, b1 = re;
        int r = launch_sweep(h, 0, t0, t1, nullptr, b0, b1);
        if (r == DS_OK) r = exchange_phi(h, 0, false, h->side_stream);
        if (r != DS_OK) return r;
        ds::Geometry upcoming = h->g;                           // second colour: one row, into the planes
        for (int s = 0; s < 4; ++s) upcoming.plane_off[1][s] = h->alt_off[s];  // this iteration produces
        r = launch_sweep(h, 1, rb, h->up >= 0 ? rb + 1 : rb, &upcoming, h->down >= 0 ? re - 1 : re, re);
        if (r == DS_OK) r = exchange_phi(h, 1, true, h->side_stream);
        return r;
    });
    if (rc == DS_OK) rc = launch_fused(h, fb, fe);  // handle's stream; makes the produced planes current
    if (rc == DS_OK) rc = order_streams(h, h->side_stream, h->ev_side, h->stream);
    return rc;
}

// Two iterations of a band in one step (see fused2_step in diffstereo.hip for the whole image):
// the two-iteration kernel over the band's interior on the handle's stream; on the boundary
// stream, the frame around it through the one-iteration kernel twice (X -> Y -> Z), and for the
// band's first / last row, which that kernel cannot take next to a neighbour, the half-sweeps
// of the one-iteration protocol, twice, each followed by its exchange: first colour from X,
// exchange, second colour into Y, exchange, first colour from Y, exchange, second colour into
// Z, exchange -- the same messages in the same order as two single iterations, so a band that
// steps by two may have neighbours that do not.
int band_fused2_step(ds_handle *h)
{
    ds::Geometry &g = h->g;
    const ds_handle::Fused2Plan &p = h->plan2;
    const int rb = g.row0, re = g.row0 + g.Hb;
    const bool up = h->up >= 0, down = h->down >= 0;
    const long long *X = g.plane_off[1], *Y = h->y_off, *Z = h->alt_off;
    const int all = (g.W + p.strip1 - 1) / p.strip1;
    // rows the one-iteration kernel takes at the top / bottom of the band (a band's own first /
    // last row only where the image ends there)
    const int top0 = up ? rb + 1 : rb, bot1 = down ? re - 1 : re;
    auto sweep_boundary = [&](int colour, const long long *black) -> int {
        ds::Geometry geo = g;
        for (int s = 0; s < 4; ++s) geo.plane_off[1][s] = black[s];
        const int n = colour == 0 ? 2 : 1;  // the first colour's two rows feed the second colour's one
        return launch_sweep(h, colour, rb, up ? rb + n : rb, &geo, down ? re - n : re, re);
    };
    // Three queues beside each other.  Boundary stream: the four half-sweeps of the band's first /
    // last rows, each followed by its exchange -- nothing else, every launch on it waits for compute
    // units the interior kernel holds.  Frame stream: the frame's launches, which read X and Y
    // inside the band only.  Handle's stream: the interior.  Two events tie the first two together:
    // the second iteration's boundary rows read Y's rows next to them (frame stream, ev_rows), and
    // the frame's rows of the second iteration read Y's boundary and halo rows (boundary stream,
    // after the second exchange: ev_halo).  Every launch of the step writes its own part of Y or
    // Z; the step ends when all three queues have.
    int rc = need_frame_stream(h);
    if (rc == DS_OK) rc = order_streams(h, h->stream, h->ev_main, h->side_stream);
    if (rc == DS_OK) HIPCHK(h, hipStreamWaitEvent(h->frame_stream, h->ev_main, 0));
    if (rc != DS_OK) return rc;
    const FrameRegion rows1[2] = {{0, all, top0, p.rT + 2}, {0, all, p.rB - 2, bot1}};
    const FrameRegion cols1[2] = {{0, p.left1, p.rT + 2, p.rB - 2}, {p.right1_begin, p.right1, p.rT + 2, p.rB - 2}};
    const FrameRegion cols2[2] = {{0, p.left2, p.rT, p.rB, p.cL}, {p.cR, p.right2, p.rT, p.rB}};
    const FrameRegion rows2[2] = {{0, all, top0, p.rT}, {0, all, p.rB, bot1}};
    // iteration n + 1 on the widened frame, X -> Y: rows, then columns
    const bool last = h->store_first;  // the call's last step: its second pass and the interior store the first colour
    h->store_first = false;
    rc = on_stream(h, h->frame_stream, [&]() -> int {
        int r = launch_fused_regions(h, X, Y, rows1, 2, 1);
        if (r != DS_OK) return r;
        HIPCHK(h, hipEventRecord(h->ev_rows, h->frame_stream));
        return launch_fused_regions(h, X, Y, cols1, 2, 1);
    });
    h->store_first = last;
    if (rc != DS_OK) return rc;
    rc = on_side_stream(h, [&]() -> int {
        hipStream_t side = h->side_stream;
        int r = sweep_boundary(0, X);
        if (r == DS_OK) r = exchange_planes(h, g.plane_off[0], side);
        if (r == DS_OK) r = sweep_boundary(1, Y);
        if (r == DS_OK) r = exchange_planes(h, Y, side);
        if (r != DS_OK) return r;
        HIPCHK(h, hipEventRecord(h->ev_halo, side));
        // iteration n + 2: the first colour's boundary rows read Y's rows below / above them
        HIPCHK(h, hipStreamWaitEvent(side, h->ev_rows, 0));
        r = sweep_boundary(0, Y);
        if (r == DS_OK) r = exchange_planes(h, g.plane_off[0], side);
        if (r == DS_OK) r = sweep_boundary(1, Z);
        if (r == DS_OK) r = exchange_planes(h, Z, side);
        return r;
    });
    if (rc != DS_OK) return rc;
    // iteration n + 2 on the frame, Y -> Z: columns (rows rT - 2 .. rB + 2 of Y: the launches
    // above), then rows (Y's boundary and halo rows: the second exchange)
    rc = on_stream(h, h->frame_stream, [&]() -> int {
        int r = launch_fused_regions(h, Y, Z, cols2, 2, 1);
        if (r != DS_OK) return r;
        HIPCHK(h, hipStreamWaitEvent(h->frame_stream, h->ev_halo, 0));
        return launch_fused_regions(h, Y, Z, rows2, 2, 1);
    });
    if (rc != DS_OK) return rc;
    if (rc == DS_OK) rc = launch_fused2_interior(h);
    if (rc == DS_OK) rc = order_streams(h, h->side_stream, h->ev_side, h->stream);
    if (rc == DS_OK) rc = order_streams(h, h->frame_stream, h->ev_frame, h->stream);
    if (rc != DS_OK) return rc;
    for (int s = 0; s < 4; ++s) std::swap(g.plane_off[1][s], h->alt_off[s]);
    h->fused_pending = !h->store_first;
    return DS_OK;
}

// ---- deep halo: one exchange per two-iteration step -----------------------------------------
//
// Two iterations reach four rows: the second colour of iteration n + 2 at row i is a function of
// the second colour of iteration n at rows i - 4 .. i + 4 (and of nothing else: the first colour
// is a function of the second).  A deep-halo band stores three rows of each neighbouring band
// beyond its own (plus the usual halo row: four rows of the second colour's planes, all four
// slots), so the two-iteration kernel takes every own row of the band itself -- no half-sweeps
// of boundary rows, no exchange in the middle of a step -- and the only communication of a step
// is ONE exchange of those four rows with each neighbour: 4 slots x 4 rows per direction
// (31.5 MB at 4K x 256 f32), gathered into one buffer so that a neighbour gets one transfer,
// where the one-row protocol moves 4 x 1 row in four dependent exchanges.  The stored rows beyond the own rows are the stored band's frame: the
// kernels run over them as over any band and leave stale values there, which the next
// exchange overwrites with the neighbour's.
//
// Order inside a step (o0, o1 = the own rows; X the current planes, Z the ones produced):
//   handle's stream   the kernel over rows [o0 + 4, o1 - 4) in two halves: they read own rows of
//                     X only, so the first half starts at once -- the exchange of the step
//                     before is still in flight -- and the second when the boundary stream's
//                     launches of this step are done
//   frame stream      the column frame (one-iteration kernel, X -> Y -> Z) over the same rows,
//                     and at the image's top / bottom the rows the two-iteration kernel leaves out
//   boundary stream   behind the exchange of the step before (stream order): the column frame
//                     and the two-iteration kernel over rows [o0, o0 + 4) and [o1 - 4, o1), which
//                     read the halo rows (after the kernel's launches an event: the second half
//                     of the long launch waits for it); then an event that closes the step's
//                     compute on this stream; then the exchange of these very rows of Z
// The handle's stream never waits for an exchange: the second half starts when the band's first
// and last rows are done, and at the end of the step it joins the frame stream and the boundary
// stream's compute.  The exchange has the rest of the step and the next step's first half.  All waits are for
// work queued earlier; no priorities.
int exchange_deep(ds_handle *h, const long long *planes, hipStream_t stream)
{
    if (h->up < 0 && h->down < 0) return DS_OK;
    const ds::Geometry &g = h->g;
    const int o0 = own_begin(h), o1 = own_end(h);
    char *stage = static_cast<char *>(h->deep_stage);
    char *stage_recv = stage + (size_t)(32 * g.row_stride) * h->elem;
    auto rows_kernel = [&](const ds::HaloRows &rows, void *buffer, int pack) {
        const dim3 grid(64, 8), block(ds::kBlock);
        if (h->dtype == DS_F64)
            hipLaunchKernelGGL(ds::halo_rows_kernel<double>, grid, block, 0, stream, static_cast<double *>(h->phi), rows,
                               g.row_stride, static_cast<double *>(buffer), pack);
        else
            hipLaunchKernelGGL(ds::halo_rows_kernel<float>, grid, block, 0, stream, static_cast<float *>(h->phi), rows,
                               g.row_stride, static_cast<float *>(buffer), pack);
    };
    // the four rows of every slot's plane into one buffer per neighbour, that buffer in one
    // transfer, and out again on the other side (rows as rows of the stored planes: halo row = 0)
    ds::HaloRows out{}, in{};
    for (int slot = 0; slot < 4; ++slot) out.plane[slot] = in.plane[slot] = planes[slot];
    out.row[0] = h->up >= 0 ? o0 - g.row0 + 1 : -1;
    out.row[1] = h->down >= 0 ? o1 - 4 - g.row0 + 1 : -1;
    in.row[0] = h->up >= 0 ? o0 - 4 - g.row0 + 1 : -1;
    in.row[1] = h->down >= 0 ? o1 - g.row0 + 1 : -1;
    rows_kernel(out, stage, 1);
    HIPCHK(h, hipGetLastError());
    int peers[2], n = 0;
    void *send[2], *recv[2];
    size_t bytes[2];
    for (int side = 0; side < 2; ++side) {
        const int peer = side == 0 ? h->up : h->down;
        if (peer < 0) continue;
        peers[n] = peer;
        send[n] = stage + (size_t)((long long)side * 16 * g.row_stride) * h->elem;
        recv[n] = stage_recv + (size_t)((long long)side * 16 * g.row_stride) * h->elem;
        bytes[n] = (size_t)(16 * g.row_stride) * h->elem;
        ++n;
    }
    if (h->transport.exchange(h->transport.ctx, n, peers, send, recv, bytes, stream) != 0)
        return transport_fail(h, "the deep halo exchange");
    rows_kernel(in, stage_recv, 0);
    HIPCHK(h, hipGetLastError());
    return DS_OK;
}

int band_deep_step(ds_handle *h)
{
    ds::Geometry &g = h->g;
    const ds_handle::Fused2Plan &p = h->plan2;
    const int o0 = own_begin(h), o1 = own_end(h);
    const bool up = h->up >= 0, down = h->down >= 0;
    const long long *X = g.plane_off[1], *Y = h->y_off, *Z = h->alt_off;
    const int all = (g.W + p.strip1 - 1) / p.strip1;
    // rows of the long launch; next to a neighbour p.rT == o0 and p.rB == o1
    const int m0 = up ? o0 + 4 : p.rT, m1 = down ? o1 - 4 : p.rB;
    const int c0 = up ? m0 - 2 : p.rT + 2, c1 = down ? m1 + 2 : p.rB - 2;  // the column frame's first pass beside it
    const bool last = h->store_first;  // the call's last step: second passes and the kernel store the first colour
    int rc = need_frame_stream(h);
    if (rc == DS_OK) rc = order_streams(h, h->stream, h->ev_main, h->frame_stream);
    if (rc != DS_OK) return rc;
    HIPCHK(h, hipStreamWaitEvent(h->side_stream, h->ev_main, 0));
    // frame stream: what reads no halo row
    rc = on_stream(h, h->frame_stream, [&]() -> int {
        FrameRegion first[4] = {{0, p.left1, c0, c1}, {p.right1_begin, p.right1, c0, c1}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        FrameRegion second[4] = {{0, p.left2, m0, m1, p.cL}, {p.cR, p.right2, m0, m1}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        int n = 2;
        if (!up) {
            first[n] = {0, all, o0, p.rT + 2};
            second[n++] = {0, all, o0, p.rT};
        }
        if (!down) {
            first[n] = {0, all, p.rB - 2, o1};
            second[n++] = {0, all, p.rB, o1};
        }
        h->store_first = false;
        int r = launch_fused_regions(h, X, Y, first, n, 1);
        h->store_first = last;
        if (r != DS_OK) return r;
        HIPCHK(h, hipEventRecord(h->ev_rows, h->frame_stream));
        return launch_fused_regions(h, Y, Z, second, n, 1);
    });
    // The long launch in two halves.  The first starts at once; the second waits for the boundary
    // stream's launches of this step: those read the halo rows, so they come behind the exchange
    // of the step before and then get compute units only as blocks of the first half end -- if the
    // whole kernel were queued ahead of them they would trickle through the entire step and this
    // step's exchange would start when the step is over.  This way the exchange is queued when the
    // second half starts and has that half and the next step's first half to complete in.
    const int mid = m0 + (m1 - m0) / 2;
    if (rc == DS_OK) rc = launch_fused2_rows(h, m0, mid);
    if (rc != DS_OK) return rc;
    // boundary stream: what reads the halo rows, then the exchange of what it made
    rc = on_side_stream(h, [&]() -> int {
        FrameRegion first[4], second[4];
        int n = 0;
        if (up) {
            first[n] = {0, p.left1, o0 - 2, m0 - 2};
            second[n++] = {0, p.left2, o0, m0, p.cL};
            first[n] = {p.right1_begin, p.right1, o0 - 2, m0 - 2};
            second[n++] = {p.cR, p.right2, o0, m0};
        }
        if (down) {
            first[n] = {0, p.left1, m1 + 2, o1 + 2};
            second[n++] = {0, p.left2, m1, o1, p.cL};
            first[n] = {p.right1_begin, p.right1, m1 + 2, o1 + 2};
            second[n++] = {p.cR, p.right2, m1, o1};
        }
        h->store_first = false;
        int r = launch_fused_regions(h, X, Y, first, n, 1);
        h->store_first = last;
        if (r == DS_OK && up) r = launch_fused2_rows(h, o0, m0);
        if (r == DS_OK && down) r = launch_fused2_rows(h, m1, o1);
        if (r != DS_OK) return r;
        HIPCHK(h, hipEventRecord(h->ev_side, h->side_stream));         // the second half of the long launch waits for this
        HIPCHK(h, hipStreamWaitEvent(h->side_stream, h->ev_rows, 0));  // the second pass reads rows of the frame stream's first
        r = launch_fused_regions(h, Y, Z, second, n, 1);
        if (r != DS_OK) return r;
        HIPCHK(h, hipEventRecord(h->ev_halo, h->side_stream));         // all of the step's compute on this stream
        return exchange_deep(h, Z, h->side_stream);
    });
    if (rc != DS_OK) return rc;
    HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_side, 0));
    rc = launch_fused2_rows(h, mid, m1);
    if (rc == DS_OK) rc = order_streams(h, h->frame_stream, h->ev_frame, h->stream);
    if (rc != DS_OK) return rc;
    HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_halo, 0));  // the boundary stream's second pass, not the exchange
    for (int s = 0; s < 4; ++s) std::swap(g.plane_off[1][s], h->alt_off[s]);
    h->fused_pending = !h->store_first;
    return DS_OK;
}

// One iteration of a deep-halo band (the odd one of a call): the one-iteration kernel over all
// stored rows it can take; its reach is two rows, so every own row comes out right.  It reads the
// halo rows, so it waits for the exchange of the step before, and its own exchange follows it.
int band_deep_single(ds_handle *h)
{
    int fb = 0, fe = 0;
    fused_safe_rows(h, &fb, &fe);
    int rc = order_streams(h, h->side_stream, h->ev_side, h->stream);
    if (rc == DS_OK) rc = launch_fused(h, fb, fe);
    if (rc == DS_OK) rc = order_streams(h, h->stream, h->ev_main, h->side_stream);
    if (rc == DS_OK) rc = exchange_deep(h, h->g.plane_off[1], h->side_stream);
    return rc;
}

int reduce_scalar(ds_handle *h, void *dev, int count, int kind)
{
    if (h->up < 0 && h->down < 0) return DS_OK;  // a single band
    if (h->transport.all_reduce(h->transport.ctx, dev, count, kind, h->stream) != 0)
        return transport_fail(h, "the reduction over the ranks");
    return DS_OK;
}

// Mask halo rows of the crossing-out, both colours, to both neighbours, in one group.
int exchange_masks(ds_handle *h)
{
    const ds::Geometry g = own_view(h);
    int peers[4], n = 0;
    void *send[4], *recv[4];
    size_t bytes[4];
    const size_t row = (size_t)g.Wh * g.Dpad;
    for (int colour = 0; colour < 2; ++colour)
        for (int side = 0; side < 2; ++side) {
            const int peer = side == 0 ? h->up : h->down;
            if (peer < 0) continue;
            uint8_t *base = h->co_alive + colour * ds::mask_colour_stride(g);
            peers[n] = peer;
            send[n] = base + (size_t)(side == 0 ? 1 : g.Hb) * row;
            recv[n] = base + (size_t)(side == 0 ? 0 : g.Hb + 1) * row;
            bytes[n] = row;
            ++n;
        }
    if (n == 0) return DS_OK;
    if (h->transport.exchange(h->transport.ctx, n, peers, send, recv, bytes, h->stream) != 0)
        return transport_fail(h, "the mask halo exchange");
    return DS_OK;
}

}  // namespace

extern "C" {

int ds_comm_unique_id(uint8_t id[DS_COMM_ID_BYTES])
{
    if (!id) return DS_ERR_ARG;
    static_assert(sizeof(ncclUniqueId) == DS_COMM_ID_BYTES, "unique id size");
    RcclApi *api = rccl();
    if (!api) {
        g_comm_error = g_rccl.err;
        return DS_ERR_STATE;
    }
    ncclUniqueId u;
    ncclResult_t r = api->GetUniqueId(&u);
    if (r != ncclSuccess) {
        comm_fail(nullptr, "ncclGetUniqueId", r);
        return DS_ERR_HIP;
    }
    std::memcpy(id, u.internal, DS_COMM_ID_BYTES);
    return DS_OK;
}

int ds_comm_create(const uint8_t id[DS_COMM_ID_BYTES], int rank, int world, int device, ds_comm **out)
{
    if (!out) return DS_ERR_ARG;
    *out = nullptr;
    if (!id || world < 1 || rank < 0 || rank >= world) {
        g_comm_error = "ds_comm_create: need 0 <= rank < world and an id";
        return DS_ERR_ARG;
    }
    RcclApi *api = rccl();
    if (!api) {
        g_comm_error = g_rccl.err;
        return DS_ERR_STATE;
    }
    if (device >= 0) {
        hipError_t e = hipSetDevice(device);
        if (e != hipSuccess) {
            g_comm_error = std::string("hipSetDevice: ") + hipGetErrorString(e);
            return DS_ERR_HIP;
        }
    } else if (hipGetDevice(&device) != hipSuccess) {
        g_comm_error = "hipGetDevice failed";
        return DS_ERR_HIP;
    }
    ds_comm *c = new (std::nothrow) ds_comm();
    if (!c) return DS_ERR_NOMEM;
    c->rank = rank;
    c->world = world;
    c->device = device;
    ncclUniqueId u;
    std::memcpy(u.internal, id, DS_COMM_ID_BYTES);
    ncclResult_t r = api->CommInitRank(&c->comm, world, u, rank);
    if (r != ncclSuccess) {
        comm_fail(nullptr, "ncclCommInitRank", r);
        delete c;
        return DS_ERR_HIP;
    }
    *out = c;
    return DS_OK;
}

void ds_comm_destroy(ds_comm *c)
{
    if (!c) return;
    RcclApi *api = rccl();
    if (api && c->comm) {
        (void)hipSetDevice(c->device);
        (void)api->CommDestroy(c->comm);
    }
    delete c;
}

const char *ds_comm_last_error(const ds_comm *c) { return c ? c->err.c_str() : g_comm_error.c_str(); }

int ds_band_attach(ds_handle *h, const ds_transport *t, int up_rank, int down_rank)
{
    if (!h) return DS_ERR_ARG;
    if (!t || !t->exchange || !t->all_reduce) return fail(h, DS_ERR_ARG, "transport needs exchange and all_reduce");
    const ds::Geometry &g = h->g;
    if ((up_rank >= 0) != (g.row0 > 0) || (down_rank >= 0) != (g.row0 + g.Hb < g.H))
        return fail(h, DS_ERR_ARG, "a band has a neighbour above / below exactly where the image continues");
    int rc = use_device(h);
    if (rc != DS_OK) return rc;
    rc = need_side_stream(h);
    if (rc != DS_OK) return rc;
    if (!h->d_flag) {
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&h->d_flag), sizeof(int));
        if (e != hipSuccess) return fail_hip(h, "hipMalloc(flag)", e);
    }
    h->transport = *t;
    h->up = up_rank;
    h->down = down_rank;
    h->attached = true;
    // Neighbouring bands must run the same protocol (a deep-halo band exchanges four rows of four
    // planes per step, the others one row four times): the minimum and the maximum of "deep" over
    // all ranks must agree.
    if (up_rank >= 0 || down_rank >= 0) {
        int lo = h->deep ? 1 : 0, hi = lo;
        for (int pass = 0; pass < 2; ++pass) {
            int *v = pass == 0 ? &lo : &hi;
            HIPCHK(h, hipMemcpyAsync(h->d_flag, v, sizeof(int), hipMemcpyHostToDevice, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
            rc = reduce_scalar(h, h->d_flag, 1, pass == 0 ? DS_REDUCE_MIN_I32 : DS_REDUCE_MAX_I32);
            if (rc != DS_OK) return rc;
            HIPCHK(h, hipMemcpyAsync(v, h->d_flag, sizeof(int), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
        }
        if (lo != hi) {
            h->attached = false;
            return fail(h, DS_ERR_STATE,
                        "the bands disagree on DS_BAND_HALO=deep: every band must be created with it, be a band of 16 "
                        "rows or more, four rows or more away from the image's other edge, and have an interior for the "
                        "two-iteration kernel");
        }
    }
    return DS_OK;
}

int ds_band_attach_comm(ds_handle *h, ds_comm *c, int up_rank, int down_rank)
{
    if (!h) return DS_ERR_ARG;
    if (!c) return fail(h, DS_ERR_ARG, "communicator is NULL");
    if (up_rank >= c->world || down_rank >= c->world) return fail(h, DS_ERR_ARG, "neighbour rank outside the communicator");
    if (c->device != h->device) return fail(h, DS_ERR_ARG, "the communicator was created on another device than the handle");
    ds_transport t;
    t.ctx = c;
    t.exchange = rccl_exchange;
    t.all_reduce = rccl_all_reduce;
    t.last_error = rccl_last_error;
    return ds_band_attach(h, &t, up_rank, down_rank);
}

int ds_band_exchange(ds_handle *h)
{
    if (!h) return DS_ERR_ARG;
    int rc = need_attached(h, "ds_band_exchange");
    if (rc == DS_OK) rc = need_current(h, "ds_band_exchange");
    if (rc == DS_OK) rc = use_device(h);
    for (int colour = 0; colour < 2 && rc == DS_OK; ++colour) rc = exchange_phi(h, colour, false, h->stream);
    if (rc == DS_OK && h->deep) rc = exchange_deep(h, h->g.plane_off[1], h->stream);
    if (rc != DS_OK) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return DS_OK;
}

int ds_band_iterate(ds_handle *h, int n_iters)
{
    if (!h) return DS_ERR_ARG;
    if (n_iters < 0) return fail(h, DS_ERR_ARG, "n_iters must be >= 0");
    int rc = need_attached(h, "ds_band_iterate");
    if (rc == DS_OK) rc = need_current(h, "ds_band_iterate");
    if (rc == DS_OK) rc = use_device(h);
    if (rc != DS_OK) return rc;
    const ds::Geometry &g = h->g;
    if (h->up < 0 && h->down < 0) return ds_iterate(h, n_iters);  // one band: the whole image
    int fb = 0, fe = 0;
    if (h->fused_waves) fused_safe_rows(h, &fb, &fe);
    const bool fused = n_iters > 0 && fe - fb >= 4 && g.Hb >= 8;
    const auto host_begin = std::chrono::steady_clock::now();
    if (h->deep && n_iters > 0) {
        // two iterations per step with one exchange each; an odd iteration runs alone.  The call's
        // last launches store the first colour of the own rows; its one halo row per neighbour
        // follows in an exchange of its own, so that every other entry point finds both colours'
        // halo rows current, as after the one-row protocol.
        if (!prepare_fused2(h)) return fail(h, DS_ERR_NOMEM, "a deep-halo band needs the two-iteration kernel's third set of planes");
        int left = n_iters;
        for (; left >= 2 && rc == DS_OK; left -= 2) {
            h->store_first = left == 2;
            rc = band_deep_step(h);
            ++h->fused2_steps;
        }
        if (left == 1 && rc == DS_OK) {
            h->store_first = true;
            rc = band_deep_single(h);
        }
        h->store_first = false;
        const hipError_t d1 = hipStreamSynchronize(h->side_stream), d2 = hipStreamSynchronize(h->stream);
        const hipError_t d3 = h->frame_stream ? hipStreamSynchronize(h->frame_stream) : hipSuccess;
        if (rc != DS_OK) return rc;
        if (d1 != hipSuccess) return fail_hip(h, "hipStreamSynchronize(boundary stream)", d1);
        if (d2 != hipSuccess) return fail_hip(h, "hipStreamSynchronize", d2);
        if (d3 != hipSuccess) return fail_hip(h, "hipStreamSynchronize(frame stream)", d3);
        rc = exchange_phi(h, 0, false, h->stream);
        if (rc != DS_OK) return rc;
        HIPCHK(h, hipStreamSynchronize(h->stream));
        return DS_OK;
    }
    if (fused) {
        // two iterations per step while more than two remain (the last one runs alone: see ds_iterate)
        // as in ds_iterate: the call's last launches also store the first colour (the band's first /
        // last rows get theirs from the boundary half-sweeps in any case)
        const char *sf = std::getenv("DS_STORE_FIRST");
        const bool fold = !(sf && std::strcmp(sf, "off") == 0);
        const int keep = fold ? 2 : 3;
        int left = n_iters;
        if (left >= keep && prepare_fused2(h)) {
            for (; left >= keep && rc == DS_OK; left -= 2) {
                h->store_first = fold && left == 2;
                rc = band_fused2_step(h);
                ++h->fused2_steps;
            }
        }
        for (; left > 0 && rc == DS_OK; --left) {
            h->store_first = fold && left == 1;
            rc = band_fused_iteration(h, fb, fe);
        }
        h->store_first = false;
        if (rc == DS_OK) rc = materialise_first_colour(h, g.row0, g.row0 + g.Hb);
    } else {
        for (int it = 0; it < n_iters && rc == DS_OK; ++it)
            for (int colour = 0; colour < 2 && rc == DS_OK; ++colour) rc = band_half_sweep(h, colour);
    }
    const auto host_queued = std::chrono::steady_clock::now();
    // also after a failure: nothing of this call may still be running when it returns
    const hipError_t e1 = hipStreamSynchronize(h->side_stream), e2 = hipStreamSynchronize(h->stream);
    if (std::getenv("DS_BAND_HOST_TIME")) {  // diagnostic: is the host that queues the work the limit?
        const auto done = std::chrono::steady_clock::now();
        std::fprintf(stderr, "ds_band_iterate(%d): %.3f ms queueing, %.3f ms until the device was done\n", n_iters,
                     std::chrono::duration<double, std::milli>(host_queued - host_begin).count(),
                     std::chrono::duration<double, std::milli>(done - host_begin).count());
    }
    if (rc != DS_OK) return rc;
    if (e1 != hipSuccess) return fail_hip(h, "hipStreamSynchronize(boundary stream)", e1);
    if (e2 != hipSuccess) return fail_hip(h, "hipStreamSynchronize", e2);
    return DS_OK;
}

int ds_band_energy(ds_handle *h, double *energy, double *max_residual)
{
    if (!h) return DS_ERR_ARG;
    if (!energy) return fail(h, DS_ERR_ARG, "energy is NULL");
    int rc = need_attached(h, "ds_band_energy");
    if (rc == DS_OK) rc = need_current(h, "ds_band_energy");
    if (rc == DS_OK) rc = use_device(h);
    if (rc == DS_OK)
        rc = launch_energy(h, h->part_sum, h->part_max, h->part_sum + h->part_n, h->part_max + h->part_n, h->d_scalars);
    if (rc == DS_OK) rc = reduce_scalar(h, h->d_scalars, 1, DS_REDUCE_SUM_F64);
    if (rc == DS_OK) rc = reduce_scalar(h, h->d_scalars + 1, 1, DS_REDUCE_MAX_F64);
    if (rc != DS_OK) return rc;
    double host[2] = {0, 0};
    HIPCHK(h, hipMemcpyAsync(host, h->d_scalars, sizeof host, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    *energy = host[0];
    if (max_residual) *max_residual = host[1];
    return DS_OK;
}

int ds_band_crossing_out(ds_handle *h, double epsilon, int max_passes, uint32_t *labels, uint8_t *alive_out,
                         int *not_empty, int *passes)
{
    if (!h) return DS_ERR_ARG;
    if (max_passes < 1) return fail(h, DS_ERR_ARG, "max_passes must be >= 1");
    int rc = need_attached(h, "ds_band_crossing_out");
    if (rc == DS_OK) rc = ds_co_begin(h, epsilon);
    if (rc == DS_OK) rc = exchange_masks(h);
    if (rc != DS_OK) return rc;
    int done = 0, flags[2] = {1, 0};
    for (;;) {
        rc = co_launch_pass(h);
        if (rc == DS_OK) rc = reduce_scalar(h, h->co_flags, 2, DS_REDUCE_MAX_I32);
        if (rc != DS_OK) return rc;
        HIPCHK(h, hipMemcpyAsync(flags, h->co_flags, sizeof flags, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        ++done;
        h->co_fixed = flags[1] || !flags[0];
        if (h->co_fixed) break;  // some band lost a pixel, or no band changed: the same on every rank
        if (done >= max_passes) return fail(h, DS_ERR_STATE, "crossing out did not reach its fixed point within max_passes");
        rc = exchange_masks(h);
        if (rc != DS_OK) return rc;
    }
    int ok = 0;
    rc = ds_co_finish(h, flags[1], labels, alive_out, &ok);
    if (rc != DS_OK) return rc;
    HIPCHK(h, hipMemcpyAsync(h->d_flag, &ok, sizeof ok, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));  // `ok` is on this frame
    rc = reduce_scalar(h, h->d_flag, 1, DS_REDUCE_MIN_I32);
    if (rc != DS_OK) return rc;
    HIPCHK(h, hipMemcpyAsync(&ok, h->d_flag, sizeof ok, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (not_empty) *not_empty = ok;
    if (passes) *passes = done;
    return DS_OK;
}

}  // extern "C"
