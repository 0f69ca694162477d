/* Host shim that lets the reference's CUDA kernel BODIES (sliced from
 * /root/reference at build time by oracle/build_ref.py, never copied into the
 * repo) compile as ordinary C++: one "thread" (blockIdx = threadIdx = 0,
 * blockDim = gridDim = 1) runs every grid-stride loop over its whole range,
 * float atomicAdd becomes a plain serial add.  Test infrastructure only. */
#pragma once
#include <cmath>
#include <cstdio>
#include <algorithm>
struct shim_dim3 { int x, y, z; };
static shim_dim3 blockIdx = {0, 0, 0}, threadIdx = {0, 0, 0};
static shim_dim3 blockDim = {1, 1, 1}, gridDim = {1, 1, 1};
#define __global__
#define __device__
#define __launch_bounds__(x)
#define __shared__ static
#define __syncthreads() ((void)0)
const int CUDA_NUM_THREADS = 1024;
static inline float atomicAdd(float *a, float v) { float o = *a; *a = o + v; return o; }
using std::min;
