// wave_scan.hpp -- 64-lane fp64 scans on gfx950 with DPP (no LDS traffic, no ds_bpermute).
//
// A wave64 inclusive scan is 4 row-local steps (row_shr 1,2,4,8 inside each row of 16 lanes) plus two
// row broadcasts (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3).  An fp64 value moves as
// two 32-bit DPP movs; lanes without a source keep the operation's identity (`old`, bound_ctrl = 0).
// 18 VALU instructions per scan instead of ~40 (incl. 12 LDS-crossbar ds_bpermute) for the __shfl_up form.
#pragma once
#include <hip/hip_runtime.h>

namespace genre {

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double identity, double v)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(identity), __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(identity), __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}

constexpr int kRowShr1 = 0x111, kRowShr2 = 0x112, kRowShr4 = 0x114, kRowShr8 = 0x118;
constexpr int kRowBcast15 = 0x142, kRowBcast31 = 0x143, kWaveShr1 = 0x138;

// inclusive product over lanes 0..l
__device__ __forceinline__ double wave_incl_prod(double v)
{
    v *= dpp_f64<kRowShr1, 0xf>(1.0, v);
    v *= dpp_f64<kRowShr2, 0xf>(1.0, v);
    v *= dpp_f64<kRowShr4, 0xf>(1.0, v);
    v *= dpp_f64<kRowShr8, 0xf>(1.0, v);
    v *= dpp_f64<kRowBcast15, 0xa>(1.0, v);
    v *= dpp_f64<kRowBcast31, 0xc>(1.0, v);
    return v;
}

// inclusive sum over lanes 0..l
__device__ __forceinline__ double wave_incl_sum(double v)
{
    v += dpp_f64<kRowShr1, 0xf>(0.0, v);
    v += dpp_f64<kRowShr2, 0xf>(0.0, v);
    v += dpp_f64<kRowShr4, 0xf>(0.0, v);
    v += dpp_f64<kRowShr8, 0xf>(0.0, v);
    v += dpp_f64<kRowBcast15, 0xa>(0.0, v);
    v += dpp_f64<kRowBcast31, 0xc>(0.0, v);
    return v;
}

// value of the previous lane (identity in lane 0)
__device__ __forceinline__ double wave_prev(double identity, double v) { return dpp_f64<kWaveShr1, 0xf>(identity, v); }

// value held by lane 63 / lane 0, broadcast through SGPRs
__device__ __forceinline__ double wave_last(double v)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

}  // namespace genre
