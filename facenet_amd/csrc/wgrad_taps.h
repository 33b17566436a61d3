// Interface between the C ABI in conv_igemm.hip and the tap-sharing weight-gradient kernel in conv_wgrad_taps.hip.
#pragma once
#include "common.h"
#include "../../include/facenet_hip.h"

namespace fn {

// Where a grouped weight-gradient launch puts its result.  FIRST member of every per-layer argument record, so that
// wgrad_reduce_kernel can walk records of either kernel with a byte stride.
//   ws == nullptr : the layer is not split over pixels, tiles are stored straight into dw;
//   ws != nullptr : split z stores into slab z of ws ([splits][Cout*KTOT] fp32), the reduce kernel adds the slabs in order.
struct WgradOut {
    float* dw;
    float* ws;
    int Cout, KTOT, splits;
    int store;   // 1: plain stores (dw or a slab), 0: atomicAdd into dw (single-layer legacy launch)
};

enum { WGRAD_TAPS_VARIANT = 5000000 };   // fn_conv2d_variant(d, 2) of a layer the tap-sharing kernel takes: 5000000 + code

int wgrad_taps_variant(const fn_conv_desc* d);          // 0: not a layer for this kernel
size_t wgrad_taps_arg_bytes();
// plans one layer: fills `rec` (wgrad_taps_arg_bytes() bytes), returns its workgroup count (or a negative status) and adds the
// slab floats it needs to *ws_used (ws == nullptr: sizing call)
long wgrad_taps_plan(const fn_conv_desc* d, int variant, void* rec, float* ws, long* ws_used);
int wgrad_taps_launch(const void* dev_args, const int32_t* dev_prefix, int n, int total, int variant, int dtype, hipStream_t st);

}  // namespace fn
