// Device sort + exact-duplicate removal of f64 keys (for the epsilon array of
// diffusion_stereo/src/epsilon_search.rs:39-74).  Lives in its own translation unit because
// it is the only code that needs rocPRIM.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

namespace ds {

// Sorts keys[0..n) ascending and removes exact duplicates.  `other` is a second buffer of n
// doubles; the result lands in *result (one of the two buffers), its length in *n_unique.
// Work space is allocated and freed inside.  Returns hipSuccess or the first HIP error.
hipError_t sort_unique_f64(double *keys, double *other, std::size_t n, double **result, std::size_t *n_unique,
                           hipStream_t stream);

}  // namespace ds
