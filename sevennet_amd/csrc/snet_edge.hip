// Edge embedding: |r|, Bessel radial basis x cutoff envelope,
// real spherical harmonics -- forward and the analytic reverse pass wrt edge_vec.
// One lane per edge; purely HBM-bound (reads 12 B, writes 4*(n_basis+nsh) B per edge).
#include "snet_common.h"
// clang-format off
#include "generated/sh_generated.h"
// clang-format on

namespace {

struct EdgeP {
  float rc, r_on, pref;
  int nb, kind, p, lmax, normalize;
  float coeffs[16];
};

__device__ __forceinline__ void envelope(const EdgeP &P, float r, float &env, float &denv) {
  if (P.kind == 0) {  // poly_cut: 1 - a x^p + b x^(p+1) - c x^(p+2)   (edge_embedding.py:125-132)
    const float p = (float)P.p;
    const float a = (p + 1.f) * (p + 2.f) * 0.5f, b = p * (p + 2.f), c = p * (p + 1.f) * 0.5f;
    const float x = r / P.rc;
    float xp1 = 1.f;  // x^(p-1)
    for (int i = 0; i < P.p - 1; ++i) xp1 *= x;
    const float xp = xp1 * x;
    env = 1.f - a * xp + b * xp * x - c * xp * x * x;
    denv = (-a * p * xp1 + b * (p + 1.f) * xp - c * (p + 2.f) * xp * x) / P.rc;
  } else {  // XPLOR (edge_embedding.py:150-160)
    if (r < P.r_on) {
      env = 1.f;
      denv = 0.f;
    } else {
      const float r2 = r * r, c2 = P.rc * P.rc, o2 = P.r_on * P.r_on;
      const float D = (c2 - o2) * (c2 - o2) * (c2 - o2);
      env = (c2 - r2) * (c2 - r2) * (c2 + 2.f * r2 - 3.f * o2) / D;
      denv = 12.f * r * (c2 - r2) * (o2 - r2) / D;
    }
  }
}

template <int LMAX>
__device__ __forceinline__ void sh_eval(float x, float y, float z, float (&Y)[(LMAX + 1) * (LMAX + 1)]);
template <> __device__ __forceinline__ void sh_eval<0>(float x, float y, float z, float (&Y)[1]) { sh_eval_0(x, y, z, Y); }
template <> __device__ __forceinline__ void sh_eval<1>(float x, float y, float z, float (&Y)[4]) { sh_eval_1(x, y, z, Y); }
template <> __device__ __forceinline__ void sh_eval<2>(float x, float y, float z, float (&Y)[9]) { sh_eval_2(x, y, z, Y); }
template <> __device__ __forceinline__ void sh_eval<3>(float x, float y, float z, float (&Y)[16]) { sh_eval_3(x, y, z, Y); }
template <int LMAX>
__device__ __forceinline__ void sh_grad(float x, float y, float z, const float (&g)[(LMAX + 1) * (LMAX + 1)], float &gx, float &gy, float &gz);
template <> __device__ __forceinline__ void sh_grad<0>(float x, float y, float z, const float (&g)[1], float &gx, float &gy, float &gz) { sh_grad_0(x, y, z, g, gx, gy, gz); }
template <> __device__ __forceinline__ void sh_grad<1>(float x, float y, float z, const float (&g)[4], float &gx, float &gy, float &gz) { sh_grad_1(x, y, z, g, gx, gy, gz); }
template <> __device__ __forceinline__ void sh_grad<2>(float x, float y, float z, const float (&g)[9], float &gx, float &gy, float &gz) { sh_grad_2(x, y, z, g, gx, gy, gz); }
template <> __device__ __forceinline__ void sh_grad<3>(float x, float y, float z, const float (&g)[16], float &gx, float &gy, float &gz) { sh_grad_3(x, y, z, g, gx, gy, gz); }

template <int LMAX>
__device__ __forceinline__ void sh_jac(float x, float y, float z, float (&J)[(LMAX + 1) * (LMAX + 1)][3]);
template <> __device__ __forceinline__ void sh_jac<0>(float x, float y, float z, float (&J)[1][3]) { sh_jac_0(x, y, z, J); }
template <> __device__ __forceinline__ void sh_jac<1>(float x, float y, float z, float (&J)[4][3]) { sh_jac_1(x, y, z, J); }
template <> __device__ __forceinline__ void sh_jac<2>(float x, float y, float z, float (&J)[9][3]) { sh_jac_2(x, y, z, J); }
template <> __device__ __forceinline__ void sh_jac<3>(float x, float y, float z, float (&J)[16][3]) { sh_jac_3(x, y, z, J); }

template <int LMAX>
__global__ __launch_bounds__(256) void edge_fwd_kernel(EdgeP P, const float *__restrict__ vec, int64_t E,
                                                       float *__restrict__ emb, float *__restrict__ sh,
                                                       float *__restrict__ dsh) {
  constexpr int NSH = (LMAX + 1) * (LMAX + 1);
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const float x = vec[3 * e + 0], y = vec[3 * e + 1], z = vec[3 * e + 2];
  const float r = sqrtf(x * x + y * y + z * z);
  float env, denv;
  envelope(P, r, env, denv);
  const float s = P.pref * env / r;
  for (int k = 0; k < P.nb; ++k) emb[e * P.nb + k] = sinf(P.coeffs[k] * r) * s;
  float Y[NSH];
  if (P.normalize) {
    const float ir = 1.f / r;
    sh_eval<LMAX>(x * ir, y * ir, z * ir, Y);
  } else {
    sh_eval<LMAX>(x, y, z, Y);
  }
#pragma unroll
  for (int i = 0; i < NSH; ++i) sh[e * NSH + i] = Y[i];
  if (dsh) {
    float J[NSH][3];
    float *o = dsh + e * (NSH * 3);
    if (P.normalize) {
      const float ir = 1.f / r;
      const float ux = x * ir, uy = y * ir, uz = z * ir;
      sh_jac<LMAX>(ux, uy, uz, J);
#pragma unroll
      for (int i = 0; i < NSH; ++i) {  // chain rule through u = v/|v|: (I - u u^T)/r
        const float dot = ux * J[i][0] + uy * J[i][1] + uz * J[i][2];
        o[3 * i + 0] = (J[i][0] - ux * dot) * ir;
        o[3 * i + 1] = (J[i][1] - uy * dot) * ir;
        o[3 * i + 2] = (J[i][2] - uz * dot) * ir;
      }
    } else {
      sh_jac<LMAX>(x, y, z, J);
#pragma unroll
      for (int i = 0; i < NSH; ++i) {
        o[3 * i + 0] = J[i][0];
        o[3 * i + 1] = J[i][1];
        o[3 * i + 2] = J[i][2];
      }
    }
  }
}

template <int LMAX>
__global__ __launch_bounds__(256) void edge_bwd_kernel(EdgeP P, const float *__restrict__ vec, int64_t E,
                                                       const float *__restrict__ g_emb,
                                                       const float *__restrict__ g_sh, float *__restrict__ g_vec,
                                                       int accumulate) {
  constexpr int NSH = (LMAX + 1) * (LMAX + 1);
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const float x = vec[3 * e + 0], y = vec[3 * e + 1], z = vec[3 * e + 2];
  const float r = sqrtf(x * x + y * y + z * z);
  const float ir = 1.f / r;
  float env, denv;
  envelope(P, r, env, denv);
  // d/dr of pref * sin(c r)/r * env(r)
  float gr = 0.f;
  for (int k = 0; k < P.nb; ++k) {
    float sn, cs;
    sincosf(P.coeffs[k] * r, &sn, &cs);
    const float f = sn * ir;
    const float df = (P.coeffs[k] * cs - f) * ir;
    gr += g_emb[e * P.nb + k] * P.pref * (df * env + f * denv);
  }
  float gx = 0.f, gy = 0.f, gz = 0.f;
  if (g_sh) {
  float gY[NSH];
#pragma unroll
  for (int i = 0; i < NSH; ++i) gY[i] = g_sh[e * NSH + i];
  if (P.normalize) {
    const float ux = x * ir, uy = y * ir, uz = z * ir;
    sh_grad<LMAX>(ux, uy, uz, gY, gx, gy, gz);
    const float dot = ux * gx + uy * gy + uz * gz;  // project out the radial part
    gx = (gx - ux * dot) * ir;
    gy = (gy - uy * dot) * ir;
    gz = (gz - uz * dot) * ir;
  } else {
    sh_grad<LMAX>(x, y, z, gY, gx, gy, gz);
  }
  }
  gx += gr * x * ir;
  gy += gr * y * ir;
  gz += gr * z * ir;
  if (accumulate) {
    gx += g_vec[3 * e + 0];
    gy += g_vec[3 * e + 1];
    gz += g_vec[3 * e + 2];
  }
  g_vec[3 * e + 0] = gx;
  g_vec[3 * e + 1] = gy;
  g_vec[3 * e + 2] = gz;
}

int prepare(const snet_edge_params *p, const float *coeffs_host, EdgeP &P) {
  SNET_REQUIRE(p != nullptr && coeffs_host != nullptr, "snet_edge_embed: null params");
  SNET_REQUIRE(p->n_basis >= 1 && p->n_basis <= 16, "snet_edge_embed: n_basis must be in 1..16");
  SNET_REQUIRE(p->lmax >= 0 && p->lmax <= 3, "snet_edge_embed: lmax must be in 0..3");
  SNET_REQUIRE(p->cutoff_kind == 0 || p->cutoff_kind == 1, "snet_edge_embed: unknown cutoff kind");
  SNET_REQUIRE(p->cutoff > 0.f, "snet_edge_embed: cutoff must be positive");
  P.rc = p->cutoff;
  P.r_on = p->cutoff_on;
  P.pref = 2.0f / p->cutoff;
  P.nb = p->n_basis;
  P.kind = p->cutoff_kind;
  P.p = p->poly_p;
  P.lmax = p->lmax;
  P.normalize = p->normalize;
  for (int i = 0; i < 16; ++i) P.coeffs[i] = i < p->n_basis ? coeffs_host[i] : 0.f;
  return 0;
}

// edge_vec[e] = pos[src[e]] - pos[center[e]] + shift[e], differences in fp64 (positions of a 100-A cell carry an fp32
// ulp of 8e-6 A; the reference's hosts -- ASE, LAMMPS -- also subtract in double and only then go to fp32)
__global__ void edge_vectors_kernel(const double *__restrict__ pos, const int32_t *__restrict__ center,
                                    const int32_t *__restrict__ src, const double *__restrict__ shift, int64_t E,
                                    float *__restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const double *pj = pos + 3 * (int64_t)src[e], *pi = pos + 3 * (int64_t)center[e];
#pragma unroll
  for (int k = 0; k < 3; ++k) out[3 * e + k] = (float)(pj[k] - pi[k] + (shift ? shift[3 * e + k] : 0.0));
}

// edges in source-grouped order e' (edge eperm[e']): destination atom (upper bound of the edge index in row_ptr) and
// radial row
__global__ void edges_by_source_kernel(const int32_t *__restrict__ row_ptr, int32_t n_dst, const int32_t *__restrict__ eperm,
                                       const int32_t *__restrict__ w_row, int64_t E, int32_t *__restrict__ center_t,
                                       int32_t *__restrict__ w_row_t) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= E) return;
  const int32_t e = eperm[i];
  int32_t lo = 0, hi = n_dst;  // largest node with row_ptr[node] <= e
  while (hi - lo > 1) {
    const int32_t mid = (lo + hi) >> 1;
    if (row_ptr[mid] <= e) lo = mid; else hi = mid;
  }
  center_t[i] = lo;
  w_row_t[i] = w_row ? w_row[e] : e;
}

}  // namespace

extern "C" int snet_edges_by_source(const int32_t *row_ptr, int64_t n_dst, const int32_t *eperm, const int32_t *w_row,
                                    int64_t E, int32_t *center_t, int32_t *w_row_t, void *stream) {
  if (E <= 0) return 0;
  SNET_REQUIRE(row_ptr && eperm && center_t && w_row_t && n_dst > 0, "snet_edges_by_source: null argument");
  edges_by_source_kernel<<<(unsigned)((E + 255) / 256), 256, 0, static_cast<hipStream_t>(stream)>>>(
      row_ptr, (int32_t)n_dst, eperm, w_row, E, center_t, w_row_t);
  SNET_CHECK_LAUNCH("snet_edges_by_source");
  return 0;
}

extern "C" int snet_edge_vectors(const double *pos, const int32_t *center, const int32_t *src, const double *shift,
                                 int64_t E, float *edge_vec, void *stream) {
  if (E <= 0) return 0;
  SNET_REQUIRE(pos && center && src && edge_vec, "snet_edge_vectors: null argument");
  edge_vectors_kernel<<<(unsigned)((E + 255) / 256), 256, 0, static_cast<hipStream_t>(stream)>>>(pos, center, src, shift, E,
                                                                                              edge_vec);
  SNET_CHECK_LAUNCH("snet_edge_vectors");
  return 0;
}

extern "C" int snet_edge_embed_fwd(const snet_edge_params *p, const float *coeffs, const float *edge_vec,
                                   int64_t E, float *emb, float *sh, float *dsh, void *stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  EdgeP P;
  if (int rc = prepare(p, coeffs, P)) return rc;
  if (E <= 0) return 0;
  const unsigned grid = (unsigned)((E + 255) / 256);
  switch (P.lmax) {
    case 0: edge_fwd_kernel<0><<<grid, 256, 0, st>>>(P, edge_vec, E, emb, sh, dsh); break;
    case 1: edge_fwd_kernel<1><<<grid, 256, 0, st>>>(P, edge_vec, E, emb, sh, dsh); break;
    case 2: edge_fwd_kernel<2><<<grid, 256, 0, st>>>(P, edge_vec, E, emb, sh, dsh); break;
    default: edge_fwd_kernel<3><<<grid, 256, 0, st>>>(P, edge_vec, E, emb, sh, dsh); break;
  }
  SNET_CHECK_LAUNCH("snet_edge_embed_fwd");
  return 0;
}

extern "C" int snet_edge_embed_bwd(const snet_edge_params *p, const float *coeffs, const float *edge_vec,
                                   int64_t E, const float *g_emb, const float *g_sh, float *g_vec,
                                   int32_t accumulate, void *stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  EdgeP P;
  if (int rc = prepare(p, coeffs, P)) return rc;
  if (E <= 0) return 0;
  const unsigned grid = (unsigned)((E + 255) / 256);
  switch (P.lmax) {
    case 0: edge_bwd_kernel<0><<<grid, 256, 0, st>>>(P, edge_vec, E, g_emb, g_sh, g_vec, accumulate); break;
    case 1: edge_bwd_kernel<1><<<grid, 256, 0, st>>>(P, edge_vec, E, g_emb, g_sh, g_vec, accumulate); break;
    case 2: edge_bwd_kernel<2><<<grid, 256, 0, st>>>(P, edge_vec, E, g_emb, g_sh, g_vec, accumulate); break;
    default: edge_bwd_kernel<3><<<grid, 256, 0, st>>>(P, edge_vec, E, g_emb, g_sh, g_vec, accumulate); break;
  }
  SNET_CHECK_LAUNCH("snet_edge_embed_bwd");
  return 0;
}
