#include "eps_sort.hpp"

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>

namespace ds {

hipError_t sort_unique_f64(double *keys, double *other, std::size_t n, double **result, std::size_t *n_unique,
                           hipStream_t stream)
{
    *result = keys;
    *n_unique = 0;
    if (n == 0) return hipSuccess;
    void *temp = nullptr;
    std::size_t *d_count = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&d_count), sizeof(std::size_t));
    if (e != hipSuccess) return e;
    auto done = [&](hipError_t code) {
        (void)hipFree(temp);
        (void)hipFree(d_count);
        return code;
    };
    std::size_t bytes_sort = 0, bytes_unique = 0;
    e = rocprim::radix_sort_keys(nullptr, bytes_sort, keys, other, n, 0, 64, stream);
    if (e != hipSuccess) return done(e);
    e = rocprim::unique(nullptr, bytes_unique, other, keys, d_count, n, rocprim::equal_to<double>(), stream);
    if (e != hipSuccess) return done(e);
    const std::size_t bytes = bytes_sort > bytes_unique ? bytes_sort : bytes_unique;
    e = hipMalloc(&temp, bytes ? bytes : 1);
    if (e != hipSuccess) return done(e);
    std::size_t b = bytes;
    e = rocprim::radix_sort_keys(temp, b, keys, other, n, 0, 64, stream);  // keys -> other
    if (e != hipSuccess) return done(e);
    b = bytes;
    e = rocprim::unique(temp, b, other, keys, d_count, n, rocprim::equal_to<double>(), stream);  // other -> keys
    if (e != hipSuccess) return done(e);
    std::size_t count = 0;
    e = hipMemcpyAsync(&count, d_count, sizeof count, hipMemcpyDeviceToHost, stream);
    if (e != hipSuccess) return done(e);
    e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return done(e);
    *result = keys;
    *n_unique = count;
    return done(hipSuccess);
}

}  // namespace ds
