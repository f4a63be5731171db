// Node-level elementwise kernels of the SevenNet step
// plus the radial-MLP activation.  All HBM-bound; features are `ir_mul` rows.
#include "snet_common.h"

namespace {

struct GateTable {
  int n;
  int total_ch;  // sum of mul over all segments = lanes needed per node
  snet_gate_seg seg[SNET_MAX_GATE_SEGS];
  int ch0[SNET_MAX_GATE_SEGS];
};

// One lane per (node, segment channel u); a gated lane handles its 2l+1 components so the
// gate scalar's gradient needs no cross-lane reduction.
// A workgroup of max(256, total_ch) lanes takes npb = lanes / total_ch whole nodes.  `addend` (nullable, same shape as y) is the
// self-connection term: y += addend is done here, in place (every y element belongs to exactly one lane),
// so the separate add pass over y disappears and the reverse pass finds the summed gate input in y.
__global__ __launch_bounds__(1024) void gate_fwd_kernel(GateTable T, float *__restrict__ y,
                                                       const float *__restrict__ addend, float *__restrict__ out,
                                                       int64_t n_nodes, int dim_in, int dim_out, int npb) {
  const int q = threadIdx.x / T.total_ch;  // node inside the workgroup
  const int c = threadIdx.x - q * T.total_ch;
  const int64_t node = (int64_t)blockIdx.x * npb + q;
  if (q >= npb || node >= n_nodes) return;
  int s = 0;
  while (s + 1 < T.n && c >= T.ch0[s + 1]) ++s;
  const snet_gate_seg sg = T.seg[s];
  const int u = c - T.ch0[s];
  float *yr = y + node * dim_in;
  const float *ar = addend ? addend + node * dim_in : nullptr;
  float *orow = out + node * dim_out;
  if (sg.kind == 0) {
    float v = yr[sg.in_off + u];
    if (ar) { v += ar[sg.in_off + u]; yr[sg.in_off + u] = v; }
    orow[sg.out_off + u] = snet::act_fwd_gate(v, sg.act) * sg.cst;
  } else {
    float z = yr[sg.gate_off + u];
    if (ar) { z += ar[sg.gate_off + u]; yr[sg.gate_off + u] = z; }
    const float g = snet::act_fwd_gate(z, sg.act) * sg.cst;
    const int d = 2 * sg.l + 1;
    for (int m = 0; m < d; ++m) {
      const int k = sg.in_off + m * sg.mul + u;
      float v = yr[k];
      if (ar) { v += ar[k]; yr[k] = v; }
      orow[sg.out_off + m * sg.mul + u] = v * g;
    }
  }
}

__global__ __launch_bounds__(1024) void gate_bwd_kernel(GateTable T, const float *__restrict__ y,
                                                       const float *__restrict__ g_out, float *__restrict__ g_y,
                                                       int64_t n_nodes, int dim_in, int dim_out, int npb) {
  const int q = threadIdx.x / T.total_ch;
  const int c = threadIdx.x - q * T.total_ch;
  const int64_t node = (int64_t)blockIdx.x * npb + q;
  if (q >= npb || node >= n_nodes) return;
  int s = 0;
  while (s + 1 < T.n && c >= T.ch0[s + 1]) ++s;
  const snet_gate_seg sg = T.seg[s];
  const int u = c - T.ch0[s];
  const float *yr = y + node * dim_in;
  const float *gor = g_out + node * dim_out;
  float *gyr = g_y + node * dim_in;
  if (sg.kind == 0) {
    gyr[sg.in_off + u] = gor[sg.out_off + u] * sg.cst * snet::act_grad_gate(yr[sg.in_off + u], sg.act);
  } else {
    const float z = yr[sg.gate_off + u];
    const float g = snet::act_fwd_gate(z, sg.act) * sg.cst;
    const int d = 2 * sg.l + 1;
    float acc = 0.f;
    for (int m = 0; m < d; ++m) {
      const float go = gor[sg.out_off + m * sg.mul + u];
      acc += go * yr[sg.in_off + m * sg.mul + u];
      gyr[sg.in_off + m * sg.mul + u] = go * g;
    }
    gyr[sg.gate_off + u] = acc * sg.cst * snet::act_grad_gate(z, sg.act);
  }
}

// 16-byte form: one WAVE per node, a lane owns 4 consecutive channels of a segment (every mul and offset a multiple of 4):
// a quarter of the memory instructions and four times the bytes in flight per lane; 4 nodes per 256-thread workgroup.
// row_norm (reverse kernel, nullable): row_norm[node] = norm_mult * ||g_y[node]||_2 * 1.0001, the bound snet_row_norm2 gives.
using f4 = __attribute__((ext_vector_type(4))) float;
#ifndef SNET_NT_GATE
#define SNET_NT_GATE 1   // streaming loads of y / addend / g_out (read once): step 41.0 -> 40.7 ms (round 4, same box)
#endif
__device__ __forceinline__ f4 ld4(const float *p) {
#if SNET_NT_GATE
  return __builtin_nontemporal_load(reinterpret_cast<const f4 *>(p));
#else
  return *reinterpret_cast<const f4 *>(p);
#endif
}
__device__ __forceinline__ void st4(float *p, f4 v) { *reinterpret_cast<f4 *>(p) = v; }

__global__ __launch_bounds__(256) void gate_fwd_vec_kernel(GateTable T, float *__restrict__ y, const float *__restrict__ addend,
                                                          float *__restrict__ out, int64_t n_nodes, int dim_in, int dim_out) {
  const int lane = threadIdx.x & 63;
  const int64_t node = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (node >= n_nodes) return;
  float *yr = y + node * dim_in;
  const float *ar = addend ? addend + node * dim_in : nullptr;
  float *orow = out + node * dim_out;
  for (int c = 4 * lane; c < T.total_ch; c += 256) {
    int s = 0;
    while (s + 1 < T.n && c >= T.ch0[s + 1]) ++s;
    const snet_gate_seg sg = T.seg[s];
    const int u = c - T.ch0[s];
    if (sg.kind == 0) {
      f4 v = ld4(yr + sg.in_off + u);
      if (ar) { v += ld4(ar + sg.in_off + u); st4(yr + sg.in_off + u, v); }
      f4 o;
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = snet::act_fwd_gate(v[i], sg.act) * sg.cst;
      st4(orow + sg.out_off + u, o);
    } else {
      f4 z = ld4(yr + sg.gate_off + u);
      if (ar) { z += ld4(ar + sg.gate_off + u); st4(yr + sg.gate_off + u, z); }
      f4 g;
#pragma unroll
      for (int i = 0; i < 4; ++i) g[i] = snet::act_fwd_gate(z[i], sg.act) * sg.cst;
      const int d = 2 * sg.l + 1;
      for (int m = 0; m < d; ++m) {
        const int k = sg.in_off + m * sg.mul + u;
        f4 v = ld4(yr + k);
        if (ar) { v += ld4(ar + k); st4(yr + k, v); }
        st4(orow + sg.out_off + m * sg.mul + u, v * g);
      }
    }
  }
}

__global__ __launch_bounds__(256) void gate_bwd_vec_kernel(GateTable T, const float *__restrict__ y, const float *__restrict__ g_out,
                                                          float *__restrict__ g_y, int64_t n_nodes, int dim_in, int dim_out,
                                                          float norm_mult, float *__restrict__ row_norm) {
  const int lane = threadIdx.x & 63;
  const int64_t node = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (node >= n_nodes) return;
  const float *yr = y + node * dim_in;
  const float *gor = g_out + node * dim_out;
  float *gyr = g_y + node * dim_in;
  float sq = 0.f;
  auto put = [&](int k, f4 v) {
    st4(gyr + k, v);
    sq = fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], fmaf(v[3], v[3], sq))));
  };
  for (int c = 4 * lane; c < T.total_ch; c += 256) {
    int s = 0;
    while (s + 1 < T.n && c >= T.ch0[s + 1]) ++s;
    const snet_gate_seg sg = T.seg[s];
    const int u = c - T.ch0[s];
    if (sg.kind == 0) {
      const f4 go = ld4(gor + sg.out_off + u), yv = ld4(yr + sg.in_off + u);
      f4 r;
#pragma unroll
      for (int i = 0; i < 4; ++i) r[i] = go[i] * sg.cst * snet::act_grad_gate(yv[i], sg.act);
      put(sg.in_off + u, r);
    } else {
      const f4 z = ld4(yr + sg.gate_off + u);
      f4 g, dg, acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float f_, g_;
        snet::act_pair_gate(z[i], sg.act, f_, g_);
        g[i] = f_ * sg.cst;
        dg[i] = g_;
      }
      const int d = 2 * sg.l + 1;
      for (int m = 0; m < d; ++m) {
        const f4 go = ld4(gor + sg.out_off + m * sg.mul + u);
        acc += go * ld4(yr + sg.in_off + m * sg.mul + u);
        put(sg.in_off + m * sg.mul + u, go * g);
      }
      f4 r;
#pragma unroll
      for (int i = 0; i < 4; ++i) r[i] = acc[i] * sg.cst * dg[i];
      put(sg.gate_off + u, r);
    }
  }
  if (row_norm) {  // uniform branch
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
    if (lane == 0) row_norm[node] = norm_mult * sqrtf(sq) * 1.0001f;
  }
}

bool gate_vec_ok(const GateTable &T, int dim_in, int dim_out, const void *a, const void *b, const void *c, const void *d) {
  bool ok = (dim_in & 3) == 0 && (dim_out & 3) == 0;
  for (const void *p : {a, b, c, d}) ok = ok && (reinterpret_cast<uintptr_t>(p) & 15) == 0;
  for (int i = 0; i < T.n; ++i) {
    const snet_gate_seg &g = T.seg[i];
    ok = ok && (g.mul & 3) == 0 && (g.in_off & 3) == 0 && (g.out_off & 3) == 0 && (g.kind == 0 || (g.gate_off & 3) == 0);
  }
  return ok;
}

int build_gate_table(const snet_gate_seg *segs, int n, GateTable &T) {
  SNET_REQUIRE(segs != nullptr && n >= 1 && n <= SNET_MAX_GATE_SEGS, "snet_gate: 1..16 segments required");
  T.n = n;
  int c = 0;
  for (int i = 0; i < n; ++i) {
    SNET_REQUIRE(segs[i].mul > 0 && segs[i].l >= 0 && (segs[i].kind == 0 || segs[i].kind == 1),
                 "snet_gate: malformed segment");
    T.seg[i] = segs[i];
    T.ch0[i] = c;
    c += segs[i].mul;
  }
  T.total_ch = c;
  return 0;
}

__global__ __launch_bounds__(256) void act_fwd_kernel(const float *__restrict__ z, float *__restrict__ a, int64_t n,
                                                      int act, float cst) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    a[i] = snet::act_fwd(z[i], act) * cst;
}
__global__ __launch_bounds__(256) void act_bwd_kernel(const float *__restrict__ z, const float *__restrict__ ga,
                                                      float *__restrict__ gz, int64_t n, int act, float cst) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    gz[i] = ga[i] * cst * snet::act_grad(z[i], act);
}
__global__ __launch_bounds__(256) void add_kernel(float *__restrict__ y, const float *__restrict__ x, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] += x[i];
}

// out[r, c] = x[idx_r(r), idx_c(c)] family ------------------------------------------------
__global__ __launch_bounds__(256) void embed_kernel(const float *__restrict__ table, const int32_t *__restrict__ types,
                                                    float *__restrict__ out, int64_t n, int dim) {
  const int64_t total = n * dim;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / dim;
    const int c = (int)(i - r * dim);
    out[i] = table[(int64_t)types[r] * dim + c];
  }
}
__global__ __launch_bounds__(256) void permute_cols_kernel(const float *__restrict__ x, const int32_t *__restrict__ ci,
                                                           float *__restrict__ out, int64_t n, int dim) {
  const int64_t total = n * dim;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / dim;
    const int c = (int)(i - r * dim);
    out[i] = x[r * dim + ci[c]];
  }
}
__global__ __launch_bounds__(256) void gather_rows_kernel(const float *__restrict__ x, const int32_t *__restrict__ idx,
                                                          float *__restrict__ out, int64_t n, int dim) {
  const int64_t total = n * dim;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / dim;
    const int c = (int)(i - r * dim);
    out[i] = x[(int64_t)idx[r] * dim + c];
  }
}
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(const float *__restrict__ x, const int32_t *__restrict__ idx,
                                                               float *__restrict__ y, int64_t n, int dim) {
  const int64_t total = n * dim;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / dim;
    const int c = (int)(i - r * dim);
    y[(int64_t)idx[r] * dim + c] += x[i];
  }
}

// max_k |row[k]| over one wavefront (16-byte loads when the rows allow it); every lane returns the result
__device__ __forceinline__ float row_absmax_wave(const float *__restrict__ row, int dim, int lane) {
  float m = 0.f;
  if ((dim & 3) == 0) {
    for (int k = 4 * lane; k < dim; k += 256) {
      const float4 v = *reinterpret_cast<const float4 *>(row + k);
      m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
  } else {
    for (int k = lane; k < dim; k += 64) m = fmaxf(m, fabsf(row[k]));
  }
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  return m;
}

// out[r] = max_k |x[r, k]|: one wavefront per row
__global__ void row_absmax_kernel(const float *__restrict__ x, int64_t n_rows, int dim, float *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (r >= n_rows) return;
  const float m = row_absmax_wave(x + r * dim, dim, lane);
  if (lane == 0) out[r] = m;
}

// the same for up to 8 matrices in one launch (the source-row bounds of every interaction layer at the start of the reverse
// pass: five launches of a 50-us kernel, latency-bound at brick sizes): workgroups [blk0[j], blk0[j+1]) take matrix j
constexpr int MAX_ABSMAX_JOBS = 8;
struct AbsmaxJobs {
  const float *x[MAX_ABSMAX_JOBS];
  float *out[MAX_ABSMAX_JOBS];
  long long rows[MAX_ABSMAX_JOBS];
  long long blk0[MAX_ABSMAX_JOBS + 1];
  int dim[MAX_ABSMAX_JOBS];
  int n;
};
__global__ void row_absmax_multi_kernel(AbsmaxJobs J) {
  const int lane = threadIdx.x & 63;
  int j = 0;
  while (j + 1 < J.n && (long long)blockIdx.x >= J.blk0[j + 1]) ++j;
  const int64_t r = ((int64_t)blockIdx.x - J.blk0[j]) * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (r >= J.rows[j]) return;
  const int dim = J.dim[j];
  const float m = row_absmax_wave(J.x[j] + r * dim, dim, lane);
  if (lane == 0) J.out[j][r] = m;
}

// out[r] = mult * ||x[r, :]||_2 (fp32 sum of squares: a bound needs no more)
__global__ void row_norm2_kernel(const float *__restrict__ x, int64_t n_rows, int dim, float mult, float *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (r >= n_rows) return;
  const float *row = x + r * dim;
  float m = 0.f;
  if ((dim & 3) == 0) {
    for (int k = 4 * lane; k < dim; k += 256) {
      const float4 v = *reinterpret_cast<const float4 *>(row + k);
      m = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, fmaf(v.w, v.w, m))));
    }
  } else {
    for (int k = lane; k < dim; k += 64) m = fmaf(row[k], row[k], m);
  }
  for (int o = 32; o > 0; o >>= 1) m += __shfl_xor(m, o, 64);
  if (lane == 0) out[r] = mult * sqrtf(m) * 1.0001f;
}

// out[s, c] = sum over the segment's rows; lanes over columns (coalesced row reads), 4 rows in flight.
// chunk_pos (nullable): the rows of x keep their 16-column chunks in another order (chunk c / 16 sits at position
// chunk_pos[c / 16]: the fused reverse kernel's g_xe); out is in standard order.
// VEC = 4: 16-byte loads (dim % 4 == 0, 16-byte aligned rows): a quarter of the load instructions of the scalar form
template <int VEC>
__global__ __launch_bounds__(256) void segment_sum_rows_kernel(const float *__restrict__ x, const int32_t *__restrict__ seg,
                                                               const int32_t *__restrict__ perm, int dim,
                                                               const int32_t *__restrict__ chunk_pos,
                                                               float *__restrict__ out) {
  const int s = blockIdx.x;
  const int k0 = seg[s], k1 = seg[s + 1];
  if constexpr (VEC == 4) {
    using f4 = __attribute__((ext_vector_type(4))) float;
    for (int c = 4 * threadIdx.x; c < dim; c += 4 * blockDim.x) {
      const int cx = chunk_pos ? chunk_pos[c >> 4] * 16 + (c & 15) : c;
      f4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
      int k = k0;
      // (streaming loads were measured, round 4: 1.02 -> 0.925 ms stand-alone on cold rows, but 3.01 -> 3.22 ms per step behind the reverse
      // kernel that has just written them -- the tail of g_xe is still in the cache hierarchy; plain loads stay)
      for (; k + 3 < k1; k += 4) {
        a0 += *reinterpret_cast<const f4 *>(x + (size_t)perm[k] * dim + cx);
        a1 += *reinterpret_cast<const f4 *>(x + (size_t)perm[k + 1] * dim + cx);
        a2 += *reinterpret_cast<const f4 *>(x + (size_t)perm[k + 2] * dim + cx);
        a3 += *reinterpret_cast<const f4 *>(x + (size_t)perm[k + 3] * dim + cx);
      }
      for (; k < k1; ++k) a0 += *reinterpret_cast<const f4 *>(x + (size_t)perm[k] * dim + cx);
      *reinterpret_cast<f4 *>(out + (size_t)s * dim + c) = (a0 + a1) + (a2 + a3);
    }
  } else {
    for (int c = threadIdx.x; c < dim; c += blockDim.x) {
      const int cx = chunk_pos ? chunk_pos[c >> 4] * 16 + (c & 15) : c;
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      int k = k0;
      for (; k + 3 < k1; k += 4) {
        a0 += x[(size_t)perm[k] * dim + cx];
        a1 += x[(size_t)perm[k + 1] * dim + cx];
        a2 += x[(size_t)perm[k + 2] * dim + cx];
        a3 += x[(size_t)perm[k + 3] * dim + cx];
      }
      for (; k < k1; ++k) a0 += x[(size_t)perm[k] * dim + cx];
      out[(size_t)s * dim + c] = (a0 + a1) + (a2 + a3);
    }
  }
}

void launch_segment_sum(const float *x, const int32_t *seg_ptr, const int32_t *perm, int64_t n_seg, int dim,
                        const int32_t *chunk_pos, float *out, hipStream_t st) {
  const bool vec = (dim & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
  if (vec) {  // 4 columns per thread: 480-wide rows -> 120 threads busy of 128
    const int threads = dim >= 768 ? 256 : (dim > 256 ? 128 : 64);
    segment_sum_rows_kernel<4><<<(unsigned)n_seg, threads, 0, st>>>(x, seg_ptr, perm, dim, chunk_pos, out);
  } else {
    const int threads = dim >= 256 ? 256 : (dim > 128 ? 256 : (dim > 64 ? 128 : 64));
    segment_sum_rows_kernel<1><<<(unsigned)n_seg, threads, 0, st>>>(x, seg_ptr, perm, dim, chunk_pos, out);
  }
}

// e_atom = e*scale[t]+shift[t]; deterministic two-stage sum (double)
constexpr int RED_BLOCKS = 256;
__global__ __launch_bounds__(256) void rescale_partial_kernel(const float *__restrict__ e, const int32_t *__restrict__ types,
                                                              const float *__restrict__ scale, const float *__restrict__ shift,
                                                              int n_scale, int64_t n, float *__restrict__ e_atom,
                                                              double *__restrict__ partial) {
  __shared__ double sm[4];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int t = n_scale > 1 ? types[i] : 0;
    const float v = e[i] * scale[t] + shift[t];
    e_atom[i] = v;
    acc += (double)v;
  }
  acc = snet::wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}
// Folded readout: the two readout linears have no nonlinearity between them (reduce_input_to_hidden /
// reduce_hidden_to_energy, model_build.py), so they are ONE vector v (product taken in fp64 at load time).
// One wave per atom: e = x_i . v + c accumulated in fp64, rescaled, atomic energy stored, fp64 two-stage sum.
__global__ __launch_bounds__(256) void readout_energy_kernel(const float *__restrict__ x, int64_t n, int dim,
                                                             const double *__restrict__ v, double c,
                                                             const int32_t *__restrict__ types, const float *__restrict__ scale,
                                                             const float *__restrict__ shift, int n_scale,
                                                             float *__restrict__ e_atom, double *__restrict__ partial) {
  __shared__ double sm[4];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 4 + wave; i < n; i += (int64_t)gridDim.x * 4) {
    const float *row = x + i * dim;
    double d = 0.0;
    for (int k = lane; k < dim; k += 64) d += (double)row[k] * v[k];
    d = snet::wave_sum_d(d) + c;
    const int t = n_scale > 1 ? types[i] : 0;
    const double e = d * (double)scale[t] + (double)shift[t];
    if (lane == 0) {
      e_atom[i] = (float)e;
      acc += e;
    }
  }
  if (lane == 0) sm[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}
// reverse of the folded readout: g_x[i, k] = scale[type_i] * v[k]
__global__ __launch_bounds__(256) void readout_grad_kernel(const double *__restrict__ v, int dim, const int32_t *__restrict__ types,
                                                           const float *__restrict__ scale, int n_scale, int64_t total,
                                                           float *__restrict__ g_x) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = idx / dim;
    const int k = (int)(idx - i * dim);
    g_x[idx] = (float)((double)scale[n_scale > 1 ? types[i] : 0] * v[k]);
  }
}
__global__ __launch_bounds__(64) void final_sum_kernel(const double *__restrict__ partial, int n, int stride, int ncomp,
                                                       double *__restrict__ out) {
  // out[c] = sum_b partial[b*stride + c], fixed order
  for (int c = 0; c < ncomp; ++c) {
    double acc = 0.0;
    for (int b = threadIdx.x; b < n; b += 64) acc += partial[(int64_t)b * stride + c];
    acc = snet::wave_sum_d(acc);
    if (threadIdx.x == 0) out[c] = acc;
  }
}

inline unsigned grid_for(int64_t n) {
  int64_t g = (n + 255) / 256;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (unsigned)g;
}

}  // namespace

namespace snet {
double *reduce_scratch(int64_t n_doubles, hipStream_t st);  // snet_api.cpp
void launch_final_sum(const double *partial, int n, int stride, int ncomp, double *out, hipStream_t st) {
  final_sum_kernel<<<1, 64, 0, st>>>(partial, n, stride, ncomp, out);
}
}  // namespace snet

extern "C" int snet_gate_fwd(float *y, const float *addend, float *out, int64_t n_nodes, int32_t dim_in,
                             int32_t dim_out, const snet_gate_seg *segs, int32_t n_segs, void *stream) {
  GateTable T;
  if (int rc = build_gate_table(segs, n_segs, T)) return rc;
  if (n_nodes <= 0) return 0;
  if (gate_vec_ok(T, dim_in, dim_out, y, addend, out, nullptr)) {
    gate_fwd_vec_kernel<<<(unsigned)((n_nodes + 3) / 4), 256, 0, static_cast<hipStream_t>(stream)>>>(T, y, addend, out, n_nodes,
                                                                                                  dim_in, dim_out);
    SNET_CHECK_LAUNCH("snet_gate_fwd");
    return 0;
  }
  SNET_REQUIRE(T.total_ch <= 1024, "snet_gate_fwd: more than 1024 gate channels per node");
  const int tb = T.total_ch <= 256 ? 256 : (T.total_ch + 63) / 64 * 64;
  const int npb = tb / T.total_ch;
  gate_fwd_kernel<<<(unsigned)((n_nodes + npb - 1) / npb), tb, 0, static_cast<hipStream_t>(stream)>>>(
      T, y, addend, out, n_nodes, dim_in, dim_out, npb);
  SNET_CHECK_LAUNCH("snet_gate_fwd");
  return 0;
}
extern "C" int snet_gate_bwd(const float *y, const float *g_out, float *g_y, int64_t n_nodes, int32_t dim_in,
                             int32_t dim_out, const snet_gate_seg *segs, int32_t n_segs, void *stream) {
  return snet_gate_bwd_norm(y, g_out, g_y, n_nodes, dim_in, dim_out, segs, n_segs, 0.f, nullptr, stream);
}
extern "C" int snet_gate_bwd_norm(const float *y, const float *g_out, float *g_y, int64_t n_nodes, int32_t dim_in,
                                  int32_t dim_out, const snet_gate_seg *segs, int32_t n_segs, float norm_mult,
                                  float *row_norm, void *stream) {
  GateTable T;
  if (int rc = build_gate_table(segs, n_segs, T)) return rc;
  if (n_nodes <= 0) return 0;
  if (gate_vec_ok(T, dim_in, dim_out, y, g_out, g_y, nullptr)) {
    gate_bwd_vec_kernel<<<(unsigned)((n_nodes + 3) / 4), 256, 0, static_cast<hipStream_t>(stream)>>>(
        T, y, g_out, g_y, n_nodes, dim_in, dim_out, norm_mult, row_norm);
    SNET_CHECK_LAUNCH("snet_gate_bwd");
    return 0;
  }
  SNET_REQUIRE(T.total_ch <= 1024, "snet_gate_bwd: more than 1024 gate channels per node");
  const int tb = T.total_ch <= 256 ? 256 : (T.total_ch + 63) / 64 * 64;
  const int npb = tb / T.total_ch;
  gate_bwd_kernel<<<(unsigned)((n_nodes + npb - 1) / npb), tb, 0, static_cast<hipStream_t>(stream)>>>(
      T, y, g_out, g_y, n_nodes, dim_in, dim_out, npb);
  SNET_CHECK_LAUNCH("snet_gate_bwd");
  if (row_norm) return snet_row_norm2(g_y, n_nodes, dim_in, norm_mult, row_norm, stream);
  return 0;
}
extern "C" int snet_act_fwd(const float *z, float *a, int64_t n, int32_t act, float cst, void *stream) {
  SNET_REQUIRE(act >= 0 && act < snet::N_ACT, "snet_act_fwd: unknown activation");
  if (n <= 0) return 0;
  act_fwd_kernel<<<grid_for(n), 256, 0, static_cast<hipStream_t>(stream)>>>(z, a, n, act, cst);
  SNET_CHECK_LAUNCH("snet_act_fwd");
  return 0;
}
extern "C" int snet_act_bwd(const float *z, const float *g_a, float *g_z, int64_t n, int32_t act, float cst,
                            void *stream) {
  SNET_REQUIRE(act >= 0 && act < snet::N_ACT, "snet_act_bwd: unknown activation");
  if (n <= 0) return 0;
  act_bwd_kernel<<<grid_for(n), 256, 0, static_cast<hipStream_t>(stream)>>>(z, g_a, g_z, n, act, cst);
  SNET_CHECK_LAUNCH("snet_act_bwd");
  return 0;
}
namespace {
__global__ void add_row_bias_kernel(float *__restrict__ y, const float *__restrict__ bias, int64_t n, int dim) {
  const int64_t total = n * dim;  // grid_for() caps the grid: stride over the rest
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (int64_t)gridDim.x * blockDim.x)
    y[k] += bias[k % dim];
}
}  // namespace
extern "C" int snet_add_row_bias(float *y, const float *bias, int64_t n_rows, int32_t dim, void *stream) {
  SNET_REQUIRE(dim >= 1, "snet_add_row_bias: bad dim");
  if (n_rows <= 0) return 0;
  add_row_bias_kernel<<<grid_for(n_rows * dim), 256, 0, static_cast<hipStream_t>(stream)>>>(y, bias, n_rows, dim);
  SNET_CHECK_LAUNCH("snet_add_row_bias");
  return 0;
}
extern "C" int snet_add_inplace(float *y, const float *x, int64_t n, void *stream) {
  if (n <= 0) return 0;
  add_kernel<<<grid_for(n), 256, 0, static_cast<hipStream_t>(stream)>>>(y, x, n);
  SNET_CHECK_LAUNCH("snet_add_inplace");
  return 0;
}
extern "C" int snet_embed_rows(const float *table, const int32_t *types, float *out, int64_t n, int32_t dim,
                               void *stream) {
  if (n <= 0) return 0;
  embed_kernel<<<grid_for(n * dim), 256, 0, static_cast<hipStream_t>(stream)>>>(table, types, out, n, dim);
  SNET_CHECK_LAUNCH("snet_embed_rows");
  return 0;
}
extern "C" int snet_permute_cols(const float *x, const int32_t *ci, float *out, int64_t n, int32_t dim,
                                 void *stream) {
  if (n <= 0) return 0;
  permute_cols_kernel<<<grid_for(n * dim), 256, 0, static_cast<hipStream_t>(stream)>>>(x, ci, out, n, dim);
  SNET_CHECK_LAUNCH("snet_permute_cols");
  return 0;
}
extern "C" int snet_gather_rows(const float *x, const int32_t *idx, float *out, int64_t n, int32_t dim,
                                void *stream) {
  if (n <= 0) return 0;
  gather_rows_kernel<<<grid_for(n * dim), 256, 0, static_cast<hipStream_t>(stream)>>>(x, idx, out, n, dim);
  SNET_CHECK_LAUNCH("snet_gather_rows");
  return 0;
}
extern "C" int snet_scatter_add_rows(const float *x, const int32_t *idx, float *y, int64_t n, int32_t dim,
                                     void *stream) {
  if (n <= 0) return 0;
  scatter_add_rows_kernel<<<grid_for(n * dim), 256, 0, static_cast<hipStream_t>(stream)>>>(x, idx, y, n, dim);
  SNET_CHECK_LAUNCH("snet_scatter_add_rows");
  return 0;
}
extern "C" int snet_row_absmax(const float *x, int64_t n_rows, int32_t dim, float *out, void *stream) {
  SNET_REQUIRE(dim >= 1 && n_rows < (1ll << 33), "snet_row_absmax: bad shape");
  if (n_rows <= 0) return 0;
  SNET_REQUIRE(x != nullptr && out != nullptr, "snet_row_absmax: null argument");
  row_absmax_kernel<<<(unsigned)((n_rows + 3) / 4), 256, 0, static_cast<hipStream_t>(stream)>>>(x, n_rows, dim, out);
  SNET_CHECK_LAUNCH("snet_row_absmax");
  return 0;
}
extern "C" int snet_row_absmax_multi(const float *const *x, const int64_t *n_rows, const int32_t *dims, float *const *out, int32_t n,
                                     void *stream) {
  SNET_REQUIRE(n >= 1 && n <= MAX_ABSMAX_JOBS, "snet_row_absmax_multi: 1 .. 8 matrices per call");
  SNET_REQUIRE(x != nullptr && n_rows != nullptr && dims != nullptr && out != nullptr, "snet_row_absmax_multi: null argument");
  AbsmaxJobs J{};
  long long blocks = 0;
  int k = 0;
  for (int j = 0; j < n; ++j) {
    SNET_REQUIRE(dims[j] >= 1 && n_rows[j] < (1ll << 33), "snet_row_absmax_multi: bad shape");
    if (n_rows[j] <= 0) continue;
    SNET_REQUIRE(x[j] != nullptr && out[j] != nullptr, "snet_row_absmax_multi: null matrix");
    J.x[k] = x[j]; J.out[k] = out[j]; J.rows[k] = n_rows[j]; J.dim[k] = dims[j]; J.blk0[k] = blocks;
    blocks += (n_rows[j] + 3) / 4;
    ++k;
  }
  if (k == 0) return 0;
  J.blk0[k] = blocks;
  J.n = k;
  SNET_REQUIRE(blocks < (1ll << 31), "snet_row_absmax_multi: too many rows");
  row_absmax_multi_kernel<<<(unsigned)blocks, 256, 0, static_cast<hipStream_t>(stream)>>>(J);
  SNET_CHECK_LAUNCH("snet_row_absmax_multi");
  return 0;
}
extern "C" int snet_segment_sum_rows_chunked(const float *x, const int32_t *seg_ptr, const int32_t *perm, int64_t n_seg,
                                             int32_t dim, const int32_t *chunk_pos, float *out, void *stream) {
  SNET_REQUIRE(dim >= 16 && dim % 16 == 0 && n_seg < (1ll << 31), "snet_segment_sum_rows_chunked: dim must be a multiple of 16");
  SNET_REQUIRE(chunk_pos != nullptr, "snet_segment_sum_rows_chunked: null chunk table");
  if (n_seg <= 0) return 0;
  launch_segment_sum(x, seg_ptr, perm, n_seg, dim, chunk_pos, out, static_cast<hipStream_t>(stream));
  SNET_CHECK_LAUNCH("snet_segment_sum_rows_chunked");
  return 0;
}
extern "C" int snet_row_norm2(const float *x, int64_t n_rows, int32_t dim, float mult, float *out, void *stream) {
  SNET_REQUIRE(dim >= 1 && n_rows < (1ll << 33), "snet_row_norm2: bad shape");
  if (n_rows <= 0) return 0;
  SNET_REQUIRE(x != nullptr && out != nullptr, "snet_row_norm2: null argument");
  row_norm2_kernel<<<(unsigned)((n_rows + 3) / 4), 256, 0, static_cast<hipStream_t>(stream)>>>(x, n_rows, dim, mult, out);
  SNET_CHECK_LAUNCH("snet_row_norm2");
  return 0;
}
extern "C" int snet_segment_sum_rows(const float *x, const int32_t *seg_ptr, const int32_t *perm, int64_t n_seg,
                                     int32_t dim, float *out, void *stream) {
  SNET_REQUIRE(dim >= 1 && n_seg < (1ll << 31), "snet_segment_sum_rows: bad shape");
  if (n_seg <= 0) return 0;
  launch_segment_sum(x, seg_ptr, perm, n_seg, dim, nullptr, out, static_cast<hipStream_t>(stream));
  SNET_CHECK_LAUNCH("snet_segment_sum_rows");
  return 0;
}
__global__ __launch_bounds__(256) void i32_shift_kernel(const int32_t *__restrict__ in, int32_t delta, int32_t *__restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = in[i] + delta;
}
// out[i] = in[i] + delta (the reverse kernels' tile pointers re-based on a sub-list of tiles)
extern "C" int snet_i32_shift(const int32_t *in, int32_t delta, int32_t *out, int64_t n, void *stream) {
  SNET_REQUIRE(in != nullptr && out != nullptr, "snet_i32_shift: null argument");
  if (n <= 0) return 0;
  i32_shift_kernel<<<grid_for(n), 256, 0, static_cast<hipStream_t>(stream)>>>(in, delta, out, n);
  SNET_CHECK_LAUNCH("snet_i32_shift");
  return 0;
}
extern "C" int snet_readout_energy(const float *x, int64_t n, int32_t dim, const double *v, double c, const int32_t *types,
                                   const float *scale, const float *shift, int32_t n_scale, float *e_atom, double *energy,
                                   void *stream) {
  SNET_REQUIRE(n_scale >= 1 && dim > 0, "snet_readout_energy: n_scale >= 1 and dim > 0 required");
  SNET_REQUIRE(n_scale == 1 || types != nullptr, "snet_readout_energy: species-wise scale needs types");
  hipStream_t st = static_cast<hipStream_t>(stream);
  double *partial = snet::reduce_scratch(RED_BLOCKS, st);
  SNET_REQUIRE(partial != nullptr, "snet_readout_energy: scratch allocation failed");
  readout_energy_kernel<<<RED_BLOCKS, 256, 0, st>>>(x, n < 0 ? 0 : n, dim, v, c, types, scale, shift, n_scale, e_atom, partial);
  snet::launch_final_sum(partial, RED_BLOCKS, 1, 1, energy, st);
  SNET_CHECK_LAUNCH("snet_readout_energy");
  return 0;
}
extern "C" int snet_readout_grad(const double *v, int32_t dim, const int32_t *types, const float *scale, int32_t n_scale,
                                 int64_t n, float *g_x, void *stream) {
  SNET_REQUIRE(n_scale >= 1 && dim > 0, "snet_readout_grad: n_scale >= 1 and dim > 0 required");
  SNET_REQUIRE(n_scale == 1 || types != nullptr, "snet_readout_grad: species-wise scale needs types");
  if (n <= 0) return 0;
  readout_grad_kernel<<<grid_for(n * dim), 256, 0, static_cast<hipStream_t>(stream)>>>(v, dim, types, scale, n_scale, n * dim, g_x);
  SNET_CHECK_LAUNCH("snet_readout_grad");
  return 0;
}
extern "C" int snet_rescale_reduce(const float *e_scaled, const int32_t *types, const float *scale,
                                   const float *shift, int32_t n_scale, int64_t n, float *e_atom, double *energy,
                                   void *stream) {
  SNET_REQUIRE(n_scale >= 1, "snet_rescale_reduce: n_scale >= 1 required");
  hipStream_t st = static_cast<hipStream_t>(stream);
  double *partial = snet::reduce_scratch(RED_BLOCKS, st);
  SNET_REQUIRE(partial != nullptr, "snet_rescale_reduce: scratch allocation failed");
  rescale_partial_kernel<<<RED_BLOCKS, 256, 0, st>>>(e_scaled, types, scale, shift, n_scale, n < 0 ? 0 : n, e_atom,
                                                     partial);
  snet::launch_final_sum(partial, RED_BLOCKS, 1, 1, energy, st);
  SNET_CHECK_LAUNCH("snet_rescale_reduce");
  return 0;
}
