// Internal helpers shared by the gfx950 kernels of libsnet_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "snet_hip.h"

namespace snet {

void set_error(const std::string &msg);

#define SNET_CHECK_LAUNCH(what)                                                    \
  do {                                                                             \
    hipError_t _e = hipGetLastError();                                             \
    if (_e != hipSuccess) {                                                        \
      snet::set_error(std::string(what) + ": " + hipGetErrorString(_e));           \
      return 1;                                                                    \
    }                                                                              \
  } while (0)

#define SNET_REQUIRE(cond, msg)                                                    \
  do {                                                                             \
    if (!(cond)) {                                                                 \
      snet::set_error(std::string(msg));                                           \
      return 2;                                                                    \
    }                                                                              \
  } while (0)

// ---- compiled tensor-product specialisations (filled by generated TUs) ------
struct ConvKernels {
  const char *tag;
  int dx, dout, nsh, wn, threads;
  void (*fwd)(const float *x, const float *sh, const float *w, const int32_t *row_ptr, const int32_t *src,
              int64_t n_dst, float scale, float *out, hipStream_t st);
  void (*bwd_edge)(const float *x, const float *sh, const float *w, const int32_t *row_ptr, const int32_t *src,
                   int64_t n_dst, float scale, const float *g_out, float *g_w, float *g_sh, hipStream_t st);
  void (*bwd_edge_vec)(const float *x, const float *sh, const float *dsh, const float *w, const int32_t *row_ptr,
                       const int32_t *src, int64_t n_dst, float scale, const float *g_out, float *g_w, float *g_vec,
                       hipStream_t st);
  void (*bwd_node)(const float *sh, const float *w, const int32_t *col_ptr, const int32_t *eperm,
                   const int32_t *dst, int64_t n_src, float scale, const float *g_out, float *g_x,
                   hipStream_t st);
};
void register_conv(const ConvKernels *k);

struct ConvRegistrar {
  explicit ConvRegistrar(const ConvKernels *k) { register_conv(k); }
};

// ---- device helpers -----------------------------------------------------------
// 64-lane wavefront sum; every lane returns the total.
__device__ __forceinline__ float wave_sum(float v) {
  v += __shfl_xor(v, 32, 64);
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 8, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 1, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float act_fwd(float z, int act) {
  if (act == 0) return z / (1.0f + expf(-z));  // silu
  return tanhf(z);
}
// derivative of the activation wrt its pre-activation
__device__ __forceinline__ float act_grad(float z, int act) {
  if (act == 0) {
    const float s = 1.0f / (1.0f + expf(-z));
    return s * (1.0f + z * (1.0f - s));
  }
  const float t = tanhf(z);
  return 1.0f - t * t;
}

}  // namespace snet
