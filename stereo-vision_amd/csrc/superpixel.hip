// Host side of the superpixel variant (ds_sp_* in include/diffstereo.h): handle, launches and
// the conversions between the reference's array layouts and the device's.  Device code lives in
// superpixel.hpp.  The graph is small (two nodes per window), so everything that is not the
// iteration itself favours plain code over speed.
#include "../../include/diffstereo.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <limits>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "superpixel.hpp"

struct ds_sp_handle {
    ds::SpGeometry g{};
    int dtype = DS_F64, device = 0, lpp = 64;
    double lambda = 0.;
    size_t elem = 8;
    uint8_t *d_left = nullptr, *d_right = nullptr, *superpixels = nullptr;
    int *Kt = nullptr;
    void *cost = nullptr, *weights = nullptr, *phi = nullptr;
    double *perim = nullptr;        // [colour][node][9], as the reference's edge_perimeters
    double *node_energy = nullptr;
    uint32_t *node_label = nullptr, *labels = nullptr;
    uint8_t *alive = nullptr;
    double *gmin = nullptr;
    int *flags = nullptr;
    hipStream_t stream = nullptr;
    std::string err;
};

namespace {

thread_local std::string g_sp_create_error;

int sp_fail(ds_sp_handle *h, int code, const std::string &msg)
{
    if (h) h->err = msg;
    else g_sp_create_error = msg;
    return code;
}

#define SPCHK(h, call)                                                                              \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return sp_fail(h, e_ == hipErrorOutOfMemory ? DS_ERR_NOMEM : DS_ERR_HIP,                \
                           std::string(#call) + ": " + hipGetErrorString(e_));                      \
    } while (0)

template <typename F> int sp_guarded(ds_sp_handle *h, F &&body)
{
    try {
        return body();
    } catch (const std::bad_alloc &) {
        return sp_fail(h, DS_ERR_NOMEM, "host allocation failed");
    } catch (const std::exception &e) {
        return sp_fail(h, DS_ERR_STATE, std::string("internal error: ") + e.what());
    } catch (...) {
        return sp_fail(h, DS_ERR_STATE, "internal error");
    }
}

template <typename F> int sp_dispatch(ds_sp_handle *h, F &&f)
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

void sp_release(ds_sp_handle *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    void *bufs[] = {h->d_left, h->d_right, h->superpixels, h->Kt, h->cost, h->weights, h->phi, h->perim,
                    h->node_energy, h->node_label, h->labels, h->alive, h->gmin, h->flags};
    for (void *p : bufs) (void)hipFree(p);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

// node (si, sj, s) -> index into the device's per-node arrays
long long node_of(const ds::SpGeometry &g, int si, int sj, int s)
{
    const int colour = 2 * ((si + sj) & 1) + s;
    return (long long)colour * g.colour_nodes + (long long)si * g.NJh + (sj >> 1);
}

// device weights (T) from the device's perimeters: w = (T)(perimeter * smoothing)
int upload_weights(ds_sp_handle *h, const std::vector<double> &perim_dev)
{
    const size_t n = perim_dev.size();
    SPCHK(h, hipMemcpyAsync(h->perim, perim_dev.data(), n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    if (h->dtype == DS_F64) {
        std::vector<double> w(n);
        for (size_t k = 0; k < n; ++k) w[k] = perim_dev[k] * h->lambda;
        SPCHK(h, hipMemcpyAsync(h->weights, w.data(), n * sizeof(double), hipMemcpyHostToDevice, h->stream));
        SPCHK(h, hipStreamSynchronize(h->stream));
    } else {
        std::vector<float> w(n);
        for (size_t k = 0; k < n; ++k) w[k] = (float)(perim_dev[k] * h->lambda);
        SPCHK(h, hipMemcpyAsync(h->weights, w.data(), n * sizeof(float), hipMemcpyHostToDevice, h->stream));
        SPCHK(h, hipStreamSynchronize(h->stream));
    }
    return DS_OK;
}

int launch_update(ds_sp_handle *h, int mode, int colour, int diag, long long count)
{
    if (count <= 0) return DS_OK;
    const int per_block = (ds::kBlock / 64) * (64 / h->lpp);
    const unsigned blocks = (unsigned)((count + per_block - 1) / per_block);
    int rc = sp_dispatch(h, [&](auto t, auto lpp) -> int {
        using T = decltype(t);
        hipLaunchKernelGGL((ds::sp_update_kernel<T, decltype(lpp)::value>), dim3(blocks), dim3(ds::kBlock), 0, h->stream,
                           h->g, static_cast<T *>(h->phi), static_cast<const T *>(h->cost), h->Kt,
                           static_cast<const T *>(h->weights), mode, colour, diag, count);
        return DS_OK;
    });
    if (rc != DS_OK) return rc;
    SPCHK(h, hipGetLastError());
    return DS_OK;
}

// node energies and argmin labels of all four colours
int launch_energy(ds_sp_handle *h)
{
    const int per_block = (ds::kBlock / 64) * (64 / h->lpp);
    const unsigned blocks = (unsigned)((h->g.colour_nodes + per_block - 1) / per_block);
    for (int colour = 0; colour < 4; ++colour) {
        int rc = sp_dispatch(h, [&](auto t, auto lpp) -> int {
            using T = decltype(t);
            hipLaunchKernelGGL((ds::sp_energy_kernel<T, decltype(lpp)::value>), dim3(blocks), dim3(ds::kBlock), 0,
                               h->stream, h->g, static_cast<const T *>(h->phi), static_cast<const T *>(h->cost), h->Kt,
                               static_cast<const T *>(h->weights), colour, h->node_energy, h->node_label);
            return DS_OK;
        });
        if (rc != DS_OK) return rc;
        SPCHK(h, hipGetLastError());
    }
    return DS_OK;
}

int expand_labels(ds_sp_handle *h, uint32_t *out)
{
    const long long px = (long long)h->g.H * h->g.W;
    hipLaunchKernelGGL(ds::sp_expand_kernel, dim3((unsigned)((px + 255) / 256)), dim3(256), 0, h->stream, h->g,
                       h->superpixels, h->node_label, h->labels);
    SPCHK(h, hipGetLastError());
    SPCHK(h, hipMemcpyAsync(out, h->labels, (size_t)px * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    SPCHK(h, hipStreamSynchronize(h->stream));
    return DS_OK;
}

template <typename T> int copy_phi(ds_sp_handle *h, T *out, const T *in)
{
    const ds::SpGeometry &g = h->g;
    std::vector<T> stage((size_t)(36 * g.plane_elems));
    if (out || in)
        SPCHK(h, hipMemcpyAsync(stage.data(), h->phi, stage.size() * sizeof(T), hipMemcpyDeviceToHost, h->stream));
    SPCHK(h, hipStreamSynchronize(h->stream));
    std::vector<int> K((size_t)(4 * g.colour_nodes));
    SPCHK(h, hipMemcpy(K.data(), h->Kt, K.size() * sizeof(int), hipMemcpyDeviceToHost));
    for (int si = 0; si < g.NI; ++si)
        for (int sj = 0; sj < g.NJ; ++sj)
            for (int s = 0; s < 2; ++s) {
                const int colour = 2 * ((si + sj) & 1) + s;
                const long long idx = (long long)si * g.NJh + (sj >> 1);
                const int Kn = K[(size_t)(colour * g.colour_nodes + idx)];
                for (int n = 0; n < 9; ++n) {
                    T *dev = stage.data() + ds::sp_plane(g, colour, n) + idx * g.Dpad;
                    const size_t host = ((((size_t)si * g.NJ + sj) * 2 + s) * 9 + n) * g.D;
                    if (out)
                        for (int d = 0; d < g.D; ++d) out[host + d] = d <= Kn ? dev[d] : T(0);
                    else
                        for (int d = 0; d <= Kn; ++d) dev[d] = in[host + d];
                }
            }
    if (in) {
        SPCHK(h, hipMemcpyAsync(h->phi, stage.data(), stage.size() * sizeof(T), hipMemcpyHostToDevice, h->stream));
        SPCHK(h, hipStreamSynchronize(h->stream));
    }
    return DS_OK;
}

int sp_create(const uint8_t *left, const uint8_t *right, int H, int W, int D, double lambda, int sh, int sw,
              int perimeters, int dtype, ds_sp_handle **out)
{
    if (!out) return sp_fail(nullptr, DS_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (!left || !right) return sp_fail(nullptr, DS_ERR_ARG, "image pointer is NULL");
    if (H < 1 || W < 1) return sp_fail(nullptr, DS_ERR_ARG, "H and W must be >= 1");
    if (D < 1 || D > 256) return sp_fail(nullptr, DS_ERR_ARG, "D must be in [1, 256]");
    if (sh < 1 || sw < 1 || H / sh < 1 || W / sw < 1)
        return sp_fail(nullptr, DS_ERR_ARG, "the window must be at least 1 x 1 and fit into the image");
    if (dtype != DS_F32 && dtype != DS_F64) return sp_fail(nullptr, DS_ERR_ARG, "dtype must be DS_F32 or DS_F64");
    if (!(lambda == lambda) || std::isinf(lambda)) return sp_fail(nullptr, DS_ERR_ARG, "lambda must be finite");
    const int NI = H / sh, NJ = W / sw;
    // calculate_edge_perimeters walks the last row of window row super_i over the columns
    // super_i * super_width .. (superpixels.rs:235: super_i where super_j is meant) and indexes the
    // image with them: a panic where those columns do not exist
    if (perimeters && NI >= 2 && (long long)(NI - 1) * sw > W)
        return sp_fail(nullptr, DS_ERR_ARG,
                       "calculate_edge_perimeters indexes columns up to (windows down - 1) * window width, beyond this "
                       "image's width: the reference panics here (superpixels.rs:235-237)");
    ds_sp_handle *h = new ds_sp_handle();
    auto bail = [&](int rc) {
        g_sp_create_error = h->err;
        sp_release(h);
        return rc;
    };
#define CREATE_CHK(call)                                                                          \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            h->err = std::string(#call) + ": " + hipGetErrorString(e_);                           \
            return bail(e_ == hipErrorOutOfMemory ? DS_ERR_NOMEM : DS_ERR_HIP);                   \
        }                                                                                         \
    } while (0)
    CREATE_CHK(hipGetDevice(&h->device));
    ds::SpGeometry &g = h->g;
    g.H = H; g.W = W; g.D = D; g.Dpad = (D + 3) & ~3;
    g.sh = sh; g.sw = sw; g.NI = NI; g.NJ = NJ; g.NJh = (NJ + 1) / 2;
    g.colour_nodes = (long long)NI * g.NJh;
    g.plane_elems = g.colour_nodes * g.Dpad;
    h->dtype = dtype;
    h->elem = dtype == DS_F64 ? 8 : 4;
    h->lambda = lambda;
    h->lpp = g.Dpad <= 64 ? 16 : g.Dpad <= 128 ? 32 : 64;
    const size_t nodes = (size_t)(4 * g.colour_nodes), px = (size_t)H * W;
    CREATE_CHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    CREATE_CHK(hipMalloc(&h->d_left, px));
    CREATE_CHK(hipMalloc(&h->d_right, px));
    CREATE_CHK(hipMalloc(&h->superpixels, px));
    CREATE_CHK(hipMalloc(&h->Kt, nodes * sizeof(int)));
    CREATE_CHK(hipMalloc(&h->cost, nodes * g.Dpad * h->elem));
    CREATE_CHK(hipMalloc(&h->weights, nodes * 9 * h->elem));
    CREATE_CHK(hipMalloc(&h->perim, nodes * 9 * sizeof(double)));
    CREATE_CHK(hipMalloc(&h->phi, (size_t)(36 * g.plane_elems) * h->elem));
    CREATE_CHK(hipMalloc(&h->node_energy, nodes * sizeof(double)));
    CREATE_CHK(hipMalloc(&h->node_label, nodes * sizeof(uint32_t)));
    CREATE_CHK(hipMalloc(&h->labels, px * sizeof(uint32_t)));
    CREATE_CHK(hipMalloc(&h->alive, nodes * g.Dpad));
    CREATE_CHK(hipMalloc(&h->gmin, nodes * 9 * sizeof(double)));
    CREATE_CHK(hipMalloc(&h->flags, 2 * sizeof(int)));
    CREATE_CHK(hipMemcpyAsync(h->d_left, left, px, hipMemcpyHostToDevice, h->stream));
    CREATE_CHK(hipMemcpyAsync(h->d_right, right, px, hipMemcpyHostToDevice, h->stream));
    CREATE_CHK(hipMemsetAsync(h->superpixels, 0, px, h->stream));
    CREATE_CHK(hipMemsetAsync(h->Kt, 0xff, nodes * sizeof(int), h->stream));  // -1: the padding nodes have no labels
    CREATE_CHK(hipMemsetAsync(h->cost, 0, nodes * g.Dpad * h->elem, h->stream));
    CREATE_CHK(hipMemsetAsync(h->node_label, 0, nodes * sizeof(uint32_t), h->stream));
    CREATE_CHK(hipMemsetAsync(h->node_energy, 0, nodes * sizeof(double), h->stream));
    if (dtype == DS_F64)
        hipLaunchKernelGGL(ds::sp_setup_kernel<double>, dim3(NJ, NI), dim3(ds::kBlock), 0, h->stream, g, h->d_left,
                           h->d_right, h->superpixels, h->Kt, static_cast<double *>(h->cost));
    else
        hipLaunchKernelGGL(ds::sp_setup_kernel<float>, dim3(NJ, NI), dim3(ds::kBlock), 0, h->stream, g, h->d_left,
                           h->d_right, h->superpixels, h->Kt, static_cast<float *>(h->cost));
    CREATE_CHK(hipGetLastError());
    // edge_perimeters: slot 0 of both nodes of a window = the boundary length inside the window;
    // every other slot 0, as the reference's loops over a window's last column / row leave them
    // (superpixels.rs:195-213, 234-255 test `image_j + 1 < window end` at the window's last
    // column, resp. row, which never holds).  perimeters == 0: all weights 0, the state before
    // calculate_edge_perimeters is called.
    std::vector<double> perim_dev(nodes * 9, 0.);
    if (perimeters) {
        double *inner = nullptr;
        CREATE_CHK(hipMalloc(reinterpret_cast<void **>(&inner), (size_t)NI * NJ * sizeof(double)));
        const double v = 1. - 1. / std::sqrt(2.), u = std::sqrt(2.) - 1.;
        hipLaunchKernelGGL(ds::sp_inner_perimeter_kernel, dim3((unsigned)((NI * NJ + 63) / 64)), dim3(64), 0, h->stream, g,
                           h->superpixels, u, v, inner);
        std::vector<double> host((size_t)NI * NJ);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(host.data(), inner, host.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        (void)hipFree(inner);
        CREATE_CHK(e);
        for (int si = 0; si < NI; ++si)
            for (int sj = 0; sj < NJ; ++sj)
                for (int s = 0; s < 2; ++s) perim_dev[(size_t)node_of(g, si, sj, s) * 9] = host[(size_t)si * NJ + sj];
    }
    int rc = upload_weights(h, perim_dev);
    if (rc != DS_OK) return bail(rc);
    if (dtype == DS_F64)
        hipLaunchKernelGGL(ds::sp_init_phi_kernel<double>, dim3(1024), dim3(256), 0, h->stream, g, h->Kt, static_cast<double *>(h->phi));
    else
        hipLaunchKernelGGL(ds::sp_init_phi_kernel<float>, dim3(1024), dim3(256), 0, h->stream, g, h->Kt, static_cast<float *>(h->phi));
    CREATE_CHK(hipGetLastError());
    CREATE_CHK(hipStreamSynchronize(h->stream));
#undef CREATE_CHK
    *out = h;
    return DS_OK;
}

}  // namespace

extern "C" {

int ds_sp_create(const uint8_t *left, const uint8_t *right, int H, int W, int D, double lambda, int super_height,
                 int super_width, int perimeters, int dtype, ds_sp_handle **out)
{
    return sp_guarded(nullptr, [&]() {
        return sp_create(left, right, H, W, D, lambda, super_height, super_width, perimeters, dtype, out);
    });
}

void ds_sp_destroy(ds_sp_handle *h) { sp_release(h); }

const char *ds_sp_last_error(const ds_sp_handle *h) { return h ? h->err.c_str() : g_sp_create_error.c_str(); }

int ds_sp_info(const ds_sp_handle *h, int *windows_down, int *windows_across, int *D, int *dtype)
{
    if (!h) return DS_ERR_ARG;
    if (windows_down) *windows_down = h->g.NI;
    if (windows_across) *windows_across = h->g.NJ;
    if (D) *D = h->g.D;
    if (dtype) *dtype = h->dtype;
    return DS_OK;
}

int ds_sp_get_superpixels(ds_sp_handle *h, uint8_t *out)
{
    if (!h) return DS_ERR_ARG;
    if (!out) return sp_fail(h, DS_ERR_ARG, "out is NULL");
    SPCHK(h, hipSetDevice(h->device));
    SPCHK(h, hipMemcpyAsync(out, h->superpixels, (size_t)h->g.H * h->g.W, hipMemcpyDeviceToHost, h->stream));
    SPCHK(h, hipStreamSynchronize(h->stream));
    return DS_OK;
}

int ds_sp_get_left_j(ds_sp_handle *h, int32_t *out)
{
    if (!h) return DS_ERR_ARG;
    if (!out) return sp_fail(h, DS_ERR_ARG, "out is NULL");
    SPCHK(h, hipSetDevice(h->device));
    return sp_guarded(h, [&]() -> int {
        const ds::SpGeometry &g = h->g;
        std::vector<int> K((size_t)(4 * g.colour_nodes));
        SPCHK(h, hipMemcpy(K.data(), h->Kt, K.size() * sizeof(int), hipMemcpyDeviceToHost));
        for (int si = 0; si < g.NI; ++si)
            for (int sj = 0; sj < g.NJ; ++sj)
                for (int s = 0; s < 2; ++s) out[((size_t)si * g.NJ + sj) * 2 + s] = K[(size_t)node_of(g, si, sj, s)];
        return DS_OK;
    });
}

int ds_sp_get_cost(ds_sp_handle *h, uint32_t *out)
{
    if (!h) return DS_ERR_ARG;
    if (!out) return sp_fail(h, DS_ERR_ARG, "out is NULL");
    SPCHK(h, hipSetDevice(h->device));
    return sp_guarded(h, [&]() -> int {
        const ds::SpGeometry &g = h->g;
        const size_t n = (size_t)(4 * g.colour_nodes) * g.Dpad;
        std::vector<double> c(n);
        if (h->dtype == DS_F64) {
            SPCHK(h, hipMemcpy(c.data(), h->cost, n * sizeof(double), hipMemcpyDeviceToHost));
        } else {
            std::vector<float> f(n);
            SPCHK(h, hipMemcpy(f.data(), h->cost, n * sizeof(float), hipMemcpyDeviceToHost));
            for (size_t k = 0; k < n; ++k) c[k] = f[k];
        }
        for (int si = 0; si < g.NI; ++si)
            for (int sj = 0; sj < g.NJ; ++sj)
                for (int s = 0; s < 2; ++s)
                    for (int d = 0; d < g.D; ++d)
                        out[(((size_t)si * g.NJ + sj) * 2 + s) * g.D + d] = (uint32_t)c[(size_t)node_of(g, si, sj, s) * g.Dpad + d];
        return DS_OK;
    });
}

int ds_sp_get_perimeters(ds_sp_handle *h, double *out)
{
    if (!h) return DS_ERR_ARG;
    if (!out) return sp_fail(h, DS_ERR_ARG, "out is NULL");
    SPCHK(h, hipSetDevice(h->device));
    return sp_guarded(h, [&]() -> int {
        const ds::SpGeometry &g = h->g;
        std::vector<double> p((size_t)(4 * g.colour_nodes) * 9);
        SPCHK(h, hipMemcpy(p.data(), h->perim, p.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (int si = 0; si < g.NI; ++si)
            for (int sj = 0; sj < g.NJ; ++sj)
                for (int s = 0; s < 2; ++s)
                    for (int n = 0; n < 9; ++n)
                        out[(((size_t)si * g.NJ + sj) * 2 + s) * 9 + n] = p[(size_t)node_of(g, si, sj, s) * 9 + n];
        return DS_OK;
    });
}

int ds_sp_set_perimeters(ds_sp_handle *h, const double *in)
{
    if (!h) return DS_ERR_ARG;
    if (!in) return sp_fail(h, DS_ERR_ARG, "in is NULL");
    SPCHK(h, hipSetDevice(h->device));
    return sp_guarded(h, [&]() -> int {
        const ds::SpGeometry &g = h->g;
        std::vector<double> p((size_t)(4 * g.colour_nodes) * 9, 0.);
        for (int si = 0; si < g.NI; ++si)
            for (int sj = 0; sj < g.NJ; ++sj)
                for (int s = 0; s < 2; ++s)
                    for (int n = 0; n < 9; ++n) {
                        const double v = in[(((size_t)si * g.NJ + sj) * 2 + s) * 9 + n];
                        if (!(v == v) || std::isinf(v)) return sp_fail(h, DS_ERR_ARG, "perimeters must be finite");
                        p[(size_t)node_of(g, si, sj, s) * 9 + n] = v;
                    }
        return upload_weights(h, p);
    });
}

int ds_sp_get_phi(ds_sp_handle *h, void *out)
{
    if (!h) return DS_ERR_ARG;
    if (!out) return sp_fail(h, DS_ERR_ARG, "out is NULL");
    SPCHK(h, hipSetDevice(h->device));
    return sp_guarded(h, [&]() -> int {
        if (h->dtype == DS_F64) return copy_phi<double>(h, static_cast<double *>(out), nullptr);
        return copy_phi<float>(h, static_cast<float *>(out), nullptr);
    });
}

int ds_sp_set_phi(ds_sp_handle *h, const void *in)
{
    if (!h) return DS_ERR_ARG;
    if (!in) return sp_fail(h, DS_ERR_ARG, "in is NULL");
    SPCHK(h, hipSetDevice(h->device));
    return sp_guarded(h, [&]() -> int {
        if (h->dtype == DS_F64) return copy_phi<double>(h, nullptr, static_cast<const double *>(in));
        return copy_phi<float>(h, nullptr, static_cast<const float *>(in));
    });
}

int ds_sp_iterate(ds_sp_handle *h, int n_iters, int schedule)
{
    if (!h) return DS_ERR_ARG;
    if (n_iters < 0) return sp_fail(h, DS_ERR_ARG, "n_iters must be >= 0");
    if (schedule != 0 && schedule != 1) return sp_fail(h, DS_ERR_ARG, "schedule must be 0 (four colours) or 1 (raster order)");
    SPCHK(h, hipSetDevice(h->device));
    const ds::SpGeometry &g = h->g;
    for (int it = 0; it < n_iters; ++it) {
        if (schedule == 0) {
            for (int colour = 0; colour < 4; ++colour) {
                const int rc = launch_update(h, 0, colour, 0, g.colour_nodes);
                if (rc != DS_OK) return rc;
            }
        } else {
            for (int diag = 0; diag <= g.NI + g.NJ - 2; ++diag) {
                const int first = diag - (g.NJ - 1) > 0 ? diag - (g.NJ - 1) : 0;
                const int last = diag < g.NI - 1 ? diag : g.NI - 1;
                for (int s = 0; s < 2; ++s) {
                    const int rc = launch_update(h, 1, s, diag, last - first + 1);
                    if (rc != DS_OK) return rc;
                }
            }
        }
    }
    SPCHK(h, hipStreamSynchronize(h->stream));
    return DS_OK;
}

int ds_sp_energy(ds_sp_handle *h, double *energy)
{
    if (!h) return DS_ERR_ARG;
    if (!energy) return sp_fail(h, DS_ERR_ARG, "energy is NULL");
    SPCHK(h, hipSetDevice(h->device));
    return sp_guarded(h, [&]() -> int {
        int rc = launch_energy(h);
        if (rc != DS_OK) return rc;
        const ds::SpGeometry &g = h->g;
        std::vector<double> e((size_t)(4 * g.colour_nodes));
        SPCHK(h, hipMemcpyAsync(e.data(), h->node_energy, e.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        SPCHK(h, hipStreamSynchronize(h->stream));
        double total = 0.;  // in the reference's order of nodes
        for (int si = 0; si < g.NI; ++si)
            for (int sj = 0; sj < g.NJ; ++sj)
                for (int s = 0; s < 2; ++s) total += e[(size_t)node_of(g, si, sj, s)];
        *energy = total;
        return DS_OK;
    });
}

int ds_sp_labels(ds_sp_handle *h, uint32_t *out)
{
    if (!h) return DS_ERR_ARG;
    if (!out) return sp_fail(h, DS_ERR_ARG, "out is NULL");
    SPCHK(h, hipSetDevice(h->device));
    int rc = launch_energy(h);
    if (rc != DS_OK) return rc;
    return expand_labels(h, out);
}

int ds_sp_crossing_out(ds_sp_handle *h, double epsilon, int max_passes, uint32_t *labels, uint8_t *alive_out,
                       int *not_empty, int *passes)
{
    if (!h) return DS_ERR_ARG;
    if (!(epsilon >= 0.)) return sp_fail(h, DS_ERR_ARG, "epsilon must be >= 0");
    if (max_passes < 1) return sp_fail(h, DS_ERR_ARG, "max_passes must be >= 1");
    SPCHK(h, hipSetDevice(h->device));
    return sp_guarded(h, [&]() -> int {
        const ds::SpGeometry &g = h->g;
        const size_t nodes = (size_t)(4 * g.colour_nodes);
        const size_t lds = (size_t)(9 * g.Dpad + ds::kBlock) * sizeof(double);
        SPCHK(h, hipMemsetAsync(h->alive, 0, nodes * g.Dpad, h->stream));
        auto pass = [&](int relax) -> int {
            SPCHK(h, hipMemsetAsync(h->flags, 0, 2 * sizeof(int), h->stream));
            if (h->dtype == DS_F64) {
                ds::SpCoArgs<double> a{static_cast<const double *>(h->phi), static_cast<const double *>(h->cost), h->Kt,
                                       h->perim, h->lambda, epsilon, h->alive, h->gmin, h->flags};
                hipLaunchKernelGGL(ds::sp_co_kernel<double>, dim3((unsigned)nodes), dim3(ds::kBlock), lds, h->stream, g, a, relax);
            } else {
                ds::SpCoArgs<float> a{static_cast<const float *>(h->phi), static_cast<const float *>(h->cost), h->Kt,
                                      h->perim, h->lambda, epsilon, h->alive, h->gmin, h->flags};
                hipLaunchKernelGGL(ds::sp_co_kernel<float>, dim3((unsigned)nodes), dim3(ds::kBlock), lds, h->stream, g, a, relax);
            }
            SPCHK(h, hipGetLastError());
            return DS_OK;
        };
        int rc = pass(0);
        if (rc != DS_OK) return rc;
        int done = 0, flags[2] = {1, 0};
        while (flags[0] && !flags[1]) {
            if (done >= max_passes) return sp_fail(h, DS_ERR_STATE, "crossing out did not reach its fixed point within max_passes");
            rc = pass(1);
            if (rc != DS_OK) return rc;
            SPCHK(h, hipMemcpyAsync(flags, h->flags, sizeof flags, hipMemcpyDeviceToHost, h->stream));
            SPCHK(h, hipStreamSynchronize(h->stream));
            ++done;
        }
        // a node without labels takes the support from all its neighbours, and the graph is
        // connected: the greatest consistent sub-graph is then empty
        if (flags[1]) SPCHK(h, hipMemsetAsync(h->alive, 0, nodes * g.Dpad, h->stream));
        SPCHK(h, hipMemsetAsync(h->flags, 0, 2 * sizeof(int), h->stream));
        hipLaunchKernelGGL(ds::sp_co_labels_kernel, dim3((unsigned)((nodes + 255) / 256)), dim3(256), 0, h->stream, g,
                           h->alive, h->Kt, h->node_label, h->flags + 1);
        SPCHK(h, hipGetLastError());
        SPCHK(h, hipMemcpyAsync(flags, h->flags, sizeof flags, hipMemcpyDeviceToHost, h->stream));
        if (labels) {
            rc = expand_labels(h, labels);
            if (rc != DS_OK) return rc;
        }
        SPCHK(h, hipStreamSynchronize(h->stream));
        if (alive_out) {
            std::vector<uint8_t> a(nodes * g.Dpad);
            SPCHK(h, hipMemcpy(a.data(), h->alive, a.size(), hipMemcpyDeviceToHost));
            for (int si = 0; si < g.NI; ++si)
                for (int sj = 0; sj < g.NJ; ++sj)
                    for (int s = 0; s < 2; ++s)
                        for (int d = 0; d < g.D; ++d)
                            alive_out[(((size_t)si * g.NJ + sj) * 2 + s) * g.D + d] = a[(size_t)node_of(g, si, sj, s) * g.Dpad + d];
        }
        if (not_empty) *not_empty = flags[1] == 0;
        if (passes) *passes = done;
        return DS_OK;
    });
}

}  // extern "C"
