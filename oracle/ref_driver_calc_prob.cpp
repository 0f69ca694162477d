// Driver around the reference's calc_prob kernel bodies (sliced into
// _ref/calc_prob_slice.inc by build_ref.py).  Reproduces the fills and launch
// arguments of calc_prob_kernel.cu:191-266 for dense [R,Zr] ray bundles
// (N=R, NC=X=Y=1 gives the same per-ray arithmetic).  Test infrastructure only.
#include <cstdint>
#include <cstddef>
#include "cuda_host_shim.h"
#include "calc_prob_slice.inc"

extern "C" {

void ref_calc_prob_forward(float *prob_in, float *stop_prob, int N, int NC, int X, int Y, int Zr)
{
    size_t tot = (size_t)N * NC * X * Y * Zr;
    for (size_t i = 0; i < tot; i++) stop_prob[i] = 0.0f;
    int sn = NC * X * Y * Zr, sc = X * Y * Zr, sx = Y * Zr, sy = Zr;
    calc_stop_forward_kernel(prob_in, N, NC, X, Y, Zr, sn, sc, sx, sy, 1,
                             stop_prob, sn, sc, sx, sy, 1, N * NC * X * Y);
}

void ref_calc_prob_backward(float *prob_in, float *spw, float *grad_out, int N, int NC, int X, int Y, int Zr)
{
    size_t tot = (size_t)N * NC * X * Y * Zr;
    for (size_t i = 0; i < tot; i++) grad_out[i] = 0.0f;
    int sn = NC * X * Y * Zr, sc = X * Y * Zr, sx = Y * Zr, sy = Zr;
    calc_stop_backward_kernel(prob_in, N, NC, X, Y, Zr, sn, sc, sx, sy, 1,
                              spw, sn, sc, sx, sy, 1, grad_out, sn, sc, sx, sy, 1, N * NC * X * Y);
}

}  // extern "C"
