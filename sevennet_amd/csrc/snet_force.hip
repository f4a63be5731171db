// Forces and virial from dE/d(edge_vec), deterministic: every atom
// gathers over its in-edges (CSR by center) and its out-edges (source-sorted permutation).
#include "snet_common.h"

namespace snet {
double *reduce_scratch(int64_t n_doubles, hipStream_t st);
void launch_final_sum(const double *partial, int n, int stride, int ncomp, double *out, hipStream_t st);
}  // namespace snet

namespace {
constexpr int RED_BLOCKS = 256;

__global__ __launch_bounds__(256) void force_kernel(const float *__restrict__ g, const float *__restrict__ rv,
                                                    const int32_t *__restrict__ row_ptr,
                                                    const int32_t *__restrict__ col_ptr,
                                                    const int32_t *__restrict__ eperm, int64_t n_nodes,
                                                    float *__restrict__ F, float *__restrict__ vir_atom,
                                                    double *__restrict__ partial) {
  __shared__ double sm[4][6];
  double va[6] = {0, 0, 0, 0, 0, 0};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_nodes;
       i += (int64_t)gridDim.x * blockDim.x) {
    float fx = 0.f, fy = 0.f, fz = 0.f;
    for (int e = row_ptr[i]; e < row_ptr[i + 1]; ++e) {
      fx += g[3 * (int64_t)e + 0];
      fy += g[3 * (int64_t)e + 1];
      fz += g[3 * (int64_t)e + 2];
    }
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f, v5 = 0.f;
    for (int k = col_ptr[i]; k < col_ptr[i + 1]; ++k) {
      const int64_t e = eperm[k];
      const float gx = g[3 * e + 0], gy = g[3 * e + 1], gz = g[3 * e + 2];
      const float rx = rv[3 * e + 0], ry = rv[3 * e + 1], rz = rv[3 * e + 2];
      fx -= gx;
      fy -= gy;
      fz -= gz;
      v0 += rx * gx; v1 += ry * gy; v2 += rz * gz;
      v3 += rx * gy; v4 += ry * gz; v5 += rz * gx;
    }
    F[3 * i + 0] = fx;
    F[3 * i + 1] = fy;
    F[3 * i + 2] = fz;
    if (vir_atom) {
      vir_atom[6 * i + 0] = -v0; vir_atom[6 * i + 1] = -v1; vir_atom[6 * i + 2] = -v2;
      vir_atom[6 * i + 3] = -v3; vir_atom[6 * i + 4] = -v4; vir_atom[6 * i + 5] = -v5;
    }
    va[0] -= v0; va[1] -= v1; va[2] -= v2; va[3] -= v3; va[4] -= v4; va[5] -= v5;
  }
  if (partial) {
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const double t = snet::wave_sum_d(va[c]);
      if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6][c] = t;
    }
    __syncthreads();
    if (threadIdx.x < 6) {
      const int c = threadIdx.x;
      partial[(int64_t)blockIdx.x * 6 + c] = sm[0][c] + sm[1][c] + sm[2][c] + sm[3][c];
    }
  }
}
}  // namespace

extern "C" int snet_edge_force(const float *g_vec, const float *edge_vec, const int32_t *row_ptr,
                               const int32_t *col_ptr, const int32_t *eperm, int64_t n_nodes, int64_t n_edges,
                               float *forces, float *virial_atom, double *virial_total, void *stream) {
  (void)n_edges;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n_nodes <= 0) return 0;
  double *partial = nullptr;
  if (virial_total) {
    partial = snet::reduce_scratch(RED_BLOCKS * 6, st);
    SNET_REQUIRE(partial != nullptr, "snet_edge_force: scratch allocation failed");
  }
  force_kernel<<<RED_BLOCKS, 256, 0, st>>>(g_vec, edge_vec, row_ptr, col_ptr, eperm, n_nodes, forces, virial_atom,
                                           partial);
  if (virial_total) snet::launch_final_sum(partial, RED_BLOCKS, 6, 6, virial_total, st);
  SNET_CHECK_LAUNCH("snet_edge_force");
  return 0;
}
