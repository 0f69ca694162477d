// Driver around the reference's nndistance CUDA kernel bodies (sliced into
// _ref/nnd_slice.inc by build_ref.py), run as one host "thread": the launcher
// pairs of nnd_cuda.cu:129-141 and :163-177 (memset + two kernel calls with the
// roles swapped).  Test infrastructure only.
#include <cstdint>
#include <cstddef>
#include <cstring>
#include "cuda_host_shim.h"
#include "nnd_slice.inc"

extern "C" {

void ref_nnd_forward_kernels(int b, int n, const float *xyz1, int m, const float *xyz2,
                             float *dist1, int *idx1, float *dist2, int *idx2)
{
    NmDistanceKernel(b, n, xyz1, m, xyz2, dist1, idx1);
    NmDistanceKernel(b, m, xyz2, n, xyz1, dist2, idx2);
}

void ref_nnd_backward_kernels(int b, int n, const float *xyz1, int m, const float *xyz2,
                              const float *gd1, const int *idx1, const float *gd2, const int *idx2,
                              float *gx1, float *gx2)
{
    std::memset(gx1, 0, (size_t)b * n * 3 * 4);
    std::memset(gx2, 0, (size_t)b * m * 3 * 4);
    NmDistanceGradKernel(b, n, xyz1, m, xyz2, gd1, idx1, gx1, gx2);
    NmDistanceGradKernel(b, m, xyz2, n, xyz1, gd2, idx2, gx2, gx1);
}

}  // extern "C"
