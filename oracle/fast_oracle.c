/*
 * TEST INFRASTRUCTURE ONLY -- O(D), OpenMP-parallel CPU restatement of the
 * checkerboard diffusion sweep, energy and label extraction, in f64 and f32.
 * Used by tests/ as the bit-exact checker for the HIP kernels at sizes the
 * brute-force oracle (diffusion_oracle.c) cannot reach, and by bench.py's
 * cpu_baseline leg.  Never linked or loaded by the product library.
 *
 * Parity status: validated against diffusion_oracle.c (itself pinned by the
 * reference's known-answer tests) in tests/test_fast_oracle.py.
 *
 * Build: gcc -O2 -fopenmp -ffp-contract=off -shared -fPIC (see oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline int og_fast_nn(int i, int j, int H, int W)
{
    return (j > 0) + (i > 0) + (j + 1 < W) + (i + 1 < H);
}

#define REAL double
#define SFX 64
#include "fast_oracle_impl.h"
#undef REAL
#undef SFX

#define REAL float
#define SFX 32
#include "fast_oracle_impl.h"
#undef REAL
#undef SFX
