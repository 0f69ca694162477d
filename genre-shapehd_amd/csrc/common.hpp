// common.hpp -- shared host/device helpers for libgenre_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include "genre_hip.h"

namespace genre {

// ---- error reporting (thread-local; surfaced by genre_last_error()) ---------
char *err_buf();
inline int fail(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return 0;
}

#define GENRE_REQUIRE(cond, ...) \
    do { if (!(cond)) return ::genre::fail(__VA_ARGS__); } while (0)

// Launch-error check: the reference's `cudaGetLastError()` after every launch
// (back_projection_kernel.cu:806-810).  No synchronisation.
#define GENRE_LAUNCH_CHECK(what) \
    do { hipError_t e_ = hipGetLastError(); \
         if (e_ != hipSuccess) return ::genre::fail("%s: %s", what, hipGetErrorString(e_)); } while (0)

// ---- tensor descriptor helpers ---------------------------------------------
inline bool is_f32(const genre_tensor *t, int ndim)
{
    return t && t->data && t->dtype == GENRE_F32 && t->ndim == ndim;
}
inline bool is_i32(const genre_tensor *t, int ndim)
{
    return t && t->data && t->dtype == GENRE_I32 && t->ndim == ndim;
}
inline int64_t numel(const genre_tensor *t)
{
    int64_t n = 1;
    for (int i = 0; i < t->ndim; i++) n *= t->size[i];
    return n;
}
inline bool is_contiguous(const genre_tensor *t)
{
    int64_t expect = 1;
    for (int i = t->ndim - 1; i >= 0; i--) {
        if (t->size[i] != 1 && t->stride[i] != expect) return false;
        expect *= t->size[i];
    }
    return true;
}
// dense: the elements occupy one gap-free block of memory in SOME dimension order (a permuted contiguous tensor,
// e.g. the batch-minor volumes of toolbox/_fused_render.py: empty_batch_minor) -- fills may treat it as flat
inline bool is_dense(const genre_tensor *t)
{
    int order[5], n = 0;
    for (int i = 0; i < t->ndim; i++)
        if (t->size[i] != 1) order[n++] = i;
    for (int i = 1; i < n; i++)                                          // insertion sort by stride
        for (int j = i; j > 0 && t->stride[order[j]] < t->stride[order[j - 1]]; j--) {
            const int tmp = order[j]; order[j] = order[j - 1]; order[j - 1] = tmp;
        }
    int64_t expect = 1;
    for (int i = 0; i < n; i++) {
        if (t->stride[order[i]] != expect) return false;
        expect *= t->size[order[i]];
    }
    return true;
}
inline bool same_shape(const genre_tensor *a, const genre_tensor *b)
{
    if (a->ndim != b->ndim) return false;
    for (int i = 0; i < a->ndim; i++) if (a->size[i] != b->size[i]) return false;
    return true;
}
inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Device-side strided views (element strides), passed by value to kernels.
struct View2 { float *p; int64_t s0, s1; };
struct View4 { float *p; int64_t s0, s1, s2, s3; };
struct View5 { float *p; int64_t s0, s1, s2, s3, s4; };

inline View2 view2(const genre_tensor *t) { return {(float *)t->data, t->stride[0], t->stride[1]}; }
inline View4 view4(const genre_tensor *t)
{
    return {(float *)t->data, t->stride[0], t->stride[1], t->stride[2], t->stride[3]};
}
inline View5 view5(const genre_tensor *t)
{
    return {(float *)t->data, t->stride[0], t->stride[1], t->stride[2], t->stride[3], t->stride[4]};
}

constexpr int kWave = 64;          // gfx950 wavefront
constexpr int kCUs = 256;          // MI355X

inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Kernels that take more than 64 KB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize, which is a
// property of the kernel ON ONE DEVICE.  `done` (one static per kernel instantiation, at the launch site) remembers
// the devices that have it, one bit each; devices >= 64 and failures are never cached (the call is cheap and a failure
// may be transient), so a second GPU in the same process gets its own attribute and a first failure is not permanent.
inline int reserve_lds(const char *op, const void *kernel, size_t bytes, std::atomic<uint64_t> &done)
{
    int dev = 0;
    GENRE_REQUIRE(hipGetDevice(&dev) == hipSuccess, "%s: hipGetDevice failed", op);
    const uint64_t bit = (dev >= 0 && dev < 64) ? (uint64_t)1 << dev : 0;
    if (bit && (done.load(std::memory_order_relaxed) & bit)) return 1;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    GENRE_REQUIRE(e == hipSuccess, "%s: cannot reserve %zu bytes of LDS on device %d (%s)", op, bytes, dev, hipGetErrorString(e));
    if (bit) done.fetch_or(bit, std::memory_order_relaxed);
    return 1;
}

}  // namespace genre
