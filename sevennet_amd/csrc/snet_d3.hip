// DFT-D3 dispersion on gfx950.  Own HIP implementation of what the reference's CUDA library
// computes (sevenn/pair_e3gnn/pair_d3_for_ase.cu: coordination numbers :1004-1057, C6 interpolation and dC6/dCN
// :765-845, two-body energy / forces / virial for zero and Becke-Johnson damping :1263-1694, CN-gradient forces
// :1797-1961), behind a C-ABI shaped like its ten pair_* functions (:2034-2082).
//
// Layout instead of the reference's scheme (one thread per UNORDERED pair looping over all lattice translations,
// float arithmetic, float atomics into double accumulators, managed pointer-to-pointer tables):
//   * one workgroup per atom i; its 256 threads stride over the flattened (j, translation) list of ROW i -- every
//     ordered pair is visited once from each side, so each atom's energy share, force, dE/dCN and virial share are
//     plain block reductions in a fixed order: no atomics, bit-reproducible;
//   * fp64 throughout (the reference's known answers carry float32 summation error of a few 1e-5 relative on lattice
//     sums of 1e5 translations; MI355X vector fp64 runs at half the fp32 rate);
//   * flat device tables: wrapped positions [n,3] (bohr), translations [T,3], per-pair C6 and dC6/dCN_i [n,n], the
//     reference-C6 grid of the elements present [nt,nt,5,5,3].
// Units inside: bohr / hartree (0.52917726 A, 27.21138505 eV); outputs eV, eV/A, eV/A^3.
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "snet_common.h"

namespace {
using namespace snet;

constexpr double AU_TO_ANG = 0.52917726, AU_TO_EV = 27.21138505, K1 = 16.0, K3 = -4.0;
constexpr int NTH = 256, MAXREF = 5;

struct Func {
  double s6, a1, s8, a2, alp6, alp8;
  int damping;  // 0 zero, 1 Becke-Johnson
};

__device__ __forceinline__ double block_sum(double v, double *sh) {
  // fixed-order tree over the workgroup's 256 threads (deterministic)
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// CN_i = sum over (j, tau) != (i, 0) with r^2 <= cn_cut of 1 / (1 + exp(-K1 ((rcov_i + rcov_j) / r - 1)))
__global__ __launch_bounds__(NTH) void d3_cn_kernel(const double *__restrict__ x, const double *__restrict__ tau, int n,
                                                    int T, int t_zero, const double *__restrict__ rcov, double cn_cut,
                                                    double *__restrict__ cn) {
  __shared__ double sh[4];
  const int i = blockIdx.x;
  const double xi = x[3 * i], yi = x[3 * i + 1], zi = x[3 * i + 2], rci = rcov[i];
  double acc = 0.0;
  const int64_t total = (int64_t)n * T;
  for (int64_t k = threadIdx.x; k < total; k += NTH) {
    const int j = (int)(k / T), t = (int)(k - (int64_t)j * T);
    if (j == i && t == t_zero) continue;
    const double dx = x[3 * j] - xi + tau[3 * t], dy = x[3 * j + 1] - yi + tau[3 * t + 1], dz = x[3 * j + 2] - zi + tau[3 * t + 2];
    const double r2 = dx * dx + dy * dy + dz * dz;
    if (r2 <= cn_cut) acc += 1.0 / (1.0 + exp(-K1 * ((rci + rcov[j]) / sqrt(r2) - 1.0)));
  }
  acc = block_sum(acc, sh);
  if (threadIdx.x == 0) cn[i] = acc;
}

// C6_ij(CN_i, CN_j) and dC6_ij / dCN_i: Gaussian-weighted average over the reference grid (L = exp(K3 ((CN_i - a)^2 + (CN_j - b)^2))).
// Evaluated where it is used (d3_pair_kernel), once per (i, j, chunk of images): no [n, n] tables in memory (the reference keeps
// n (n + 1) / 2 fp32 entries, pair_d3.cu; the 16 n^2 bytes of fp64 tables this file kept until round 4 would have been 160 GB at 100 k atoms)
__device__ __forceinline__ void d3_c6(double cni, double cnj, const double *__restrict__ g, int mxi, int mxj, double &c6, double &dc6) {
  double num = 0.0, den = 0.0, dnum = 0.0, dden = 0.0, rmin = 1e300, cmin = 0.0;
  for (int a = 0; a < mxi; ++a)
    for (int b = 0; b < mxj; ++b) {
      const double *e = g + (a * MAXREF + b) * 3;
      if (e[0] <= 0.0) continue;
      const double rr = (e[1] - cni) * (e[1] - cni) + (e[2] - cnj) * (e[2] - cnj);
      if (rr < rmin) { rmin = rr; cmin = e[0]; }
      const double w = exp(K3 * rr);
      num += e[0] * w;
      den += w;
      const double dw = w * 2.0 * K3 * (cni - e[1]);
      dnum += e[0] * dw;
      dden += dw;
    }
  if (den > 1e-99) {
    c6 = num / den;
    dc6 = (dnum - c6 * dden) / den;
  } else {  // all weights underflow: the nearest reference value, no CN dependence (reference :833-837)
    c6 = cmin;
    dc6 = 0.0;
  }
}

// phi(r) with E_pair = -C6 phi, and phi'(r)
__device__ __forceinline__ void d3_phi(const Func &F, double r2, double r42, double r0, double &phi, double &dphi) {
  const double r = sqrt(r2);
  if (F.damping == 1) {
    const double R0 = F.a1 * sqrt(3.0 * r42) + F.a2, R2 = R0 * R0, R6 = R2 * R2 * R2, R8 = R6 * R2;
    const double r6 = r2 * r2 * r2, r8 = r6 * r2;
    const double t6 = 1.0 / (r6 + R6), t8 = 1.0 / (r8 + R8);
    phi = F.s6 * t6 + 3.0 * F.s8 * r42 * t8;
    dphi = -(6.0 * F.s6 * r6 * t6 * t6 + 24.0 * F.s8 * r42 * r8 * t8 * t8) / r;
  } else {
    const double ir = 1.0 / r, ir2 = ir * ir, ir6 = ir2 * ir2 * ir2, ir8 = ir6 * ir2;
    const double t6 = pow(F.a1 * r0 * ir, F.alp6), t8 = pow(F.a2 * r0 * ir, F.alp8);
    const double f6 = 1.0 / (1.0 + 6.0 * t6), f8 = 1.0 / (1.0 + 6.0 * t8);
    phi = F.s6 * f6 * ir6 + 3.0 * F.s8 * r42 * f8 * ir8;
    const double df6 = 6.0 * F.alp6 * t6 * f6 * f6 * ir, df8 = 6.0 * F.alp8 * t8 * f8 * f8 * ir;
    dphi = F.s6 * ir6 * (df6 - 6.0 * f6 * ir) + 3.0 * F.s8 * r42 * ir8 * (df8 - 8.0 * f8 * ir);
  }
}

// row i of the two-body sum at fixed C6: e_i = -1/2 sum C6 phi; f_i = -sum C6 phi' d / r; dE/dCN_i = -sum phi dC6_ij/dCN_i;
// strain derivative share s_i[ab] = -1/2 sum C6 phi' d_a d_b / r
__global__ __launch_bounds__(NTH) void d3_pair_kernel(const double *__restrict__ x, const double *__restrict__ tau, int n, int T,
                                                      int t_zero, const double *__restrict__ r2r4, const double *__restrict__ r0ab,
                                                      const int32_t *__restrict__ type, int nt, const double *__restrict__ cn,
                                                      const int32_t *__restrict__ mxc, const double *__restrict__ ref, int t_chunks,
                                                      double vdw_cut, Func F,
                                                      double *__restrict__ e_atom, double *__restrict__ f, double *__restrict__ dedcn,
                                                      double *__restrict__ s_atom) {
  __shared__ double sh[4];
  const int i = blockIdx.x;
  const double xi = x[3 * i], yi = x[3 * i + 1], zi = x[3 * i + 2], q_i = r2r4[i];
  double e = 0.0, fx = 0.0, fy = 0.0, fz = 0.0, dc = 0.0, s[6] = {0, 0, 0, 0, 0, 0};
  // work item = (atom j, chunk of the T images): C6_ij is evaluated once per item; t_chunks = 1 for large n, more for small cells whose
  // many images would otherwise leave most of the block's threads without an atom j
  const int ti = type[i], ch = (T + t_chunks - 1) / t_chunks;
  const double cni = cn[i];
  const int64_t total = (int64_t)n * t_chunks;
  for (int64_t k = threadIdx.x; k < total; k += NTH) {
    const int j = (int)(k / t_chunks), t_beg = (int)(k - (int64_t)j * t_chunks) * ch, t_end = min(T, t_beg + ch);
    const int tj = type[j];
    const double xj = x[3 * j] - xi, yj = x[3 * j + 1] - yi, zj = x[3 * j + 2] - zi;
    const double r42 = q_i * r2r4[j], r0 = r0ab[(size_t)ti * nt + tj];
    double c = 0.0, dcv = 0.0;
    bool have_c6 = false;
    for (int t = t_beg; t < t_end; ++t) {
      if (j == i && t == t_zero) continue;
      const double dx = xj + tau[3 * t], dy = yj + tau[3 * t + 1], dz = zj + tau[3 * t + 2];
      const double r2 = dx * dx + dy * dy + dz * dz;
      if (r2 > vdw_cut) continue;
      if (!have_c6) {
        d3_c6(cni, cn[j], ref + ((size_t)ti * nt + tj) * (MAXREF * MAXREF * 3), mxc[ti], mxc[tj], c, dcv);
        have_c6 = true;
      }
      double phi, dphi;
      d3_phi(F, r2, r42, r0, phi, dphi);
      e -= 0.5 * c * phi;
      dc -= phi * dcv;
      const double g = c * dphi / sqrt(r2);   // C6 phi' / r
      if (j != i) { fx -= g * dx; fy -= g * dy; fz -= g * dz; }
      s[0] -= 0.5 * g * dx * dx; s[1] -= 0.5 * g * dy * dy; s[2] -= 0.5 * g * dz * dz;
      s[3] -= 0.5 * g * dx * dy; s[4] -= 0.5 * g * dx * dz; s[5] -= 0.5 * g * dy * dz;
    }
  }
  e = block_sum(e, sh); fx = block_sum(fx, sh); fy = block_sum(fy, sh); fz = block_sum(fz, sh); dc = block_sum(dc, sh);
  for (int q = 0; q < 6; ++q) s[q] = block_sum(s[q], sh);
  if (threadIdx.x == 0) {
    e_atom[i] = e; f[3 * i] = fx; f[3 * i + 1] = fy; f[3 * i + 2] = fz; dedcn[i] = dc;
    for (int q = 0; q < 6; ++q) s_atom[6 * i + q] = s[q];
  }
}

// CN-gradient part: f_i += sum (dE/dCN_i + dE/dCN_j) cnt'(r) d / r ;  s_i[ab] += sum dE/dCN_i cnt'(r) d_a d_b / r
__global__ __launch_bounds__(NTH) void d3_cn_force_kernel(const double *__restrict__ x, const double *__restrict__ tau, int n, int T,
                                                          int t_zero, const double *__restrict__ rcov, double cn_cut,
                                                          const double *__restrict__ dedcn, double *__restrict__ f,
                                                          double *__restrict__ s_atom) {
  __shared__ double sh[4];
  const int i = blockIdx.x;
  const double xi = x[3 * i], yi = x[3 * i + 1], zi = x[3 * i + 2], rci = rcov[i], di = dedcn[i];
  double fx = 0.0, fy = 0.0, fz = 0.0, s[6] = {0, 0, 0, 0, 0, 0};
  const int64_t total = (int64_t)n * T;
  for (int64_t k = threadIdx.x; k < total; k += NTH) {
    const int j = (int)(k / T), t = (int)(k - (int64_t)j * T);
    if (j == i && t == t_zero) continue;
    const double dx = x[3 * j] - xi + tau[3 * t], dy = x[3 * j + 1] - yi + tau[3 * t + 1], dz = x[3 * j + 2] - zi + tau[3 * t + 2];
    const double r2 = dx * dx + dy * dy + dz * dz;
    if (r2 > cn_cut) continue;
    const double r = sqrt(r2), rc = rci + rcov[j];
    const double ex = exp(-K1 * (rc / r - 1.0));
    const double dcnt = -K1 * rc * ex / (r2 * (1.0 + ex) * (1.0 + ex));   // d cnt / d r
    const double gi = di * dcnt / r;
    if (j != i) {
      const double g = (di + dedcn[j]) * dcnt / r;
      fx += g * dx; fy += g * dy; fz += g * dz;
    }
    s[0] += gi * dx * dx; s[1] += gi * dy * dy; s[2] += gi * dz * dz;
    s[3] += gi * dx * dy; s[4] += gi * dx * dz; s[5] += gi * dy * dz;
  }
  fx = block_sum(fx, sh); fy = block_sum(fy, sh); fz = block_sum(fz, sh);
  for (int q = 0; q < 6; ++q) s[q] = block_sum(s[q], sh);
  if (threadIdx.x == 0) {
    f[3 * i] += fx; f[3 * i + 1] += fy; f[3 * i + 2] += fz;
    for (int q = 0; q < 6; ++q) s_atom[6 * i + q] += s[q];
  }
}

template <class T>
struct Dev {
  T *p = nullptr;
  size_t cap = 0;
  bool ensure(size_t n) {
    if (n <= cap) return true;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    if (hipMalloc((void **)&p, (n ? n : 1) * sizeof(T)) != hipSuccess) return false;
    cap = n ? n : 1;
    return true;
  }
  bool put(const std::vector<T> &h, hipStream_t st) {
    return ensure(h.size()) && (h.empty() || hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, st) == hipSuccess);
  }
  ~Dev() { if (p) (void)hipFree(p); }
};

}  // namespace

struct snet_d3 {
  // published tables (from the parameter blob)
  std::vector<double> r0ab, r2r4, rcov;       // [94*94] (A), [94], [94] (bohr)
  std::vector<double> c6ref;                   // [95][95][5][5][3]
  std::vector<int> mxc;                        // [95]
  bool have_tables = false;
  Func func{1.0, 0.4289, 0.7875, 4.4407, 14.0, 16.0, 1};
  bool have_func = false;
  double vdw_cut = 9000.0, cn_cut = 1600.0;    // bohr^2
  std::vector<int32_t> z;
  std::vector<double> pos;                     // A
  double cell[9] = {0};
  int pbc[3] = {0, 0, 0};
  bool have_cell = false;
  // results
  double energy = 0.0, stress[9] = {0};
  std::vector<double> forces, cn;
  Dev<double> d_x, d_tv, d_tc, d_rcov, d_r2r4, d_r0, d_ref, d_cn, d_e, d_f, d_dedcn, d_s;
  Dev<int32_t> d_type, d_mxc;
};

extern "C" {

int snet_d3_create(snet_d3 **out) {
  SNET_REQUIRE(out != nullptr, "snet_d3_create: null argument");
  *out = new snet_d3;
  return 0;
}
void snet_d3_destroy(snet_d3 *d) { delete d; }

int snet_d3_set_tables(snet_d3 *d, const double *r0ab, const double *c6ab, int64_t n_c6, const double *r2r4, const double *rcov) {
  SNET_REQUIRE(d && r0ab && c6ab && r2r4 && rcov && n_c6 > 0, "snet_d3_set_tables: null argument");
  d->r0ab.assign(r0ab, r0ab + 94 * 94);
  d->r2r4.assign(r2r4, r2r4 + 94);
  d->rcov.assign(rcov, rcov + 94);
  d->c6ref.assign((size_t)95 * 95 * MAXREF * MAXREF * 3, 0.0);
  d->mxc.assign(95, 0);
  auto at = [&](int zi, int zj, int a, int b) { return &d->c6ref[((((size_t)zi * 95 + zj) * MAXREF + a) * MAXREF + b) * 3]; };
  for (int64_t k = 0; k < n_c6; ++k) {   // row: C6, Z_i + 100 ref_i, Z_j + 100 ref_j, CN_i, CN_j (reference :361-391)
    const double *row = c6ab + 5 * k;
    const int a1 = (int)row[1], a2 = (int)row[2];
    const int zi = (a1 - 1) % 100 + 1, ri = (a1 - 1) / 100, zj = (a2 - 1) % 100 + 1, rj = (a2 - 1) / 100;
    SNET_REQUIRE(zi >= 1 && zi <= 94 && zj >= 1 && zj <= 94 && ri < MAXREF && rj < MAXREF, "snet_d3_set_tables: bad C6 table row");
    double *p = at(zi, zj, ri, rj), *q = at(zj, zi, rj, ri);
    p[0] = row[0]; p[1] = row[3]; p[2] = row[4];
    q[0] = row[0]; q[1] = row[4]; q[2] = row[3];
    d->mxc[zi] = std::max(d->mxc[zi], ri + 1);
    d->mxc[zj] = std::max(d->mxc[zj], rj + 1);
  }
  d->have_tables = true;
  return 0;
}

int snet_d3_settings(snet_d3 *d, double vdw_cutoff_au2, double cn_cutoff_au2, int32_t damping, const double *func5) {
  SNET_REQUIRE(d && func5, "snet_d3_settings: null argument");
  SNET_REQUIRE(damping == 0 || damping == 1, "snet_d3_settings: damping must be 0 (damp_zero) or 1 (damp_bj); the reference's "
                                               "damp_zerom / damp_bjm compute nothing either (pair_d3_for_ase.cu:1783-1784)");
  SNET_REQUIRE(vdw_cutoff_au2 > 0 && cn_cutoff_au2 > 0, "snet_d3_settings: cutoffs (bohr^2) must be positive");
  d->vdw_cut = vdw_cutoff_au2;
  d->cn_cut = cn_cutoff_au2;
  // func5 = (s6, rs6, s18, rs18, alp) of the functional; a1 = rs6, a2 = rs8 = rs18, s8 = s18, alp8 = alp + 2 (:608-628)
  d->func = Func{func5[0], func5[1], func5[2], func5[3], func5[4], func5[4] + 2.0, damping};
  d->have_func = true;
  return 0;
}

int snet_d3_set_atoms(snet_d3 *d, int32_t n, const int32_t *atomic_numbers, const double *positions) {
  SNET_REQUIRE(d && n > 0 && atomic_numbers && positions, "snet_d3_set_atoms: need n > 0 atoms");
  for (int i = 0; i < n; ++i)
    SNET_REQUIRE(atomic_numbers[i] >= 1 && atomic_numbers[i] <= 94, "snet_d3_set_atoms: D3 parameters exist for Z = 1 .. 94");
  d->z.assign(atomic_numbers, atomic_numbers + n);
  d->pos.assign(positions, positions + 3 * (size_t)n);
  return 0;
}

int snet_d3_set_cell(snet_d3 *d, const double *cell9, const int32_t *pbc3) {
  SNET_REQUIRE(d && cell9 && pbc3, "snet_d3_set_cell: null argument");
  std::memcpy(d->cell, cell9, sizeof(d->cell));
  for (int k = 0; k < 3; ++k) d->pbc[k] = pbc3[k] != 0;
  d->have_cell = true;
  return 0;
}

static void translations(const double a[9], const int pbc[3], double r2_cut, std::vector<double> &tau, int &t_zero) {
  // |n_k| <= int(r_cut / height_k) + 1 along periodic axes (set_lattice_repetition_criteria, :979-1001)
  const double rc = std::sqrt(r2_cut);
  int rep[3];
  for (int k = 0; k < 3; ++k) {
    const double *u = a + 3 * ((k + 1) % 3), *v = a + 3 * ((k + 2) % 3), *w = a + 3 * k;
    const double cp[3] = {u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]};
    const double h = std::fabs((cp[0] * w[0] + cp[1] * w[1] + cp[2] * w[2]) / std::sqrt(cp[0] * cp[0] + cp[1] * cp[1] + cp[2] * cp[2]));
    rep[k] = pbc[k] ? (int)std::fabs(rc / h) + 1 : 0;
  }
  tau.clear();
  t_zero = -1;
  for (int p = -rep[0]; p <= rep[0]; ++p)
    for (int q = -rep[1]; q <= rep[1]; ++q)
      for (int r = -rep[2]; r <= rep[2]; ++r) {
        if (p == 0 && q == 0 && r == 0) t_zero = (int)(tau.size() / 3);
        for (int c = 0; c < 3; ++c) tau.push_back(p * a[c] + q * a[3 + c] + r * a[6 + c]);
      }
}

int snet_d3_compute(snet_d3 *d, void *stream) {
  SNET_REQUIRE(d != nullptr, "snet_d3_compute: null handle");
  SNET_REQUIRE(d->have_tables && d->have_func && d->have_cell && !d->z.empty(),
               "snet_d3_compute: tables, settings, atoms and cell must be set first");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int n = (int)d->z.size();
  double a[9];
  for (int k = 0; k < 9; ++k) a[k] = d->cell[k] / AU_TO_ANG;
  const double det = a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) + a[2] * (a[3] * a[7] - a[4] * a[6]);
  SNET_REQUIRE(std::fabs(det) > 1e-12, "snet_d3_compute: singular cell (give molecules a box: the reference's calculator does, calculator.py:533-548)");
  double inv[9] = {(a[4] * a[8] - a[5] * a[7]) / det, (a[2] * a[7] - a[1] * a[8]) / det, (a[1] * a[5] - a[2] * a[4]) / det,
                   (a[5] * a[6] - a[3] * a[8]) / det, (a[0] * a[8] - a[2] * a[6]) / det, (a[2] * a[3] - a[0] * a[5]) / det,
                   (a[3] * a[7] - a[4] * a[6]) / det, (a[1] * a[6] - a[0] * a[7]) / det, (a[0] * a[4] - a[1] * a[3]) / det};
  // wrap into the cell (load_atom_info, :1170-1219): fractional = x inv(cell), rows of `a` are the lattice vectors
  std::vector<double> x(3 * (size_t)n);
  for (int i = 0; i < n; ++i) {
    double fr[3];
    for (int c = 0; c < 3; ++c) {
      fr[c] = 0.0;
      for (int k = 0; k < 3; ++k) fr[c] += d->pos[3 * i + k] / AU_TO_ANG * inv[3 * k + c];
      fr[c] -= std::floor(fr[c]);
    }
    for (int c = 0; c < 3; ++c) x[3 * i + c] = fr[0] * a[c] + fr[1] * a[3 + c] + fr[2] * a[6 + c];
  }
  std::vector<double> tv, tc;
  int zv = -1, zc = -1;
  translations(a, d->pbc, d->vdw_cut, tv, zv);
  translations(a, d->pbc, d->cn_cut, tc, zc);
  const int Tv = (int)(tv.size() / 3), Tc = (int)(tc.size() / 3);
  // elements present -> dense type index and their slice of the reference-C6 grid
  std::vector<int> type_of(95, -1), elems;
  std::vector<int32_t> type(n);
  for (int i = 0; i < n; ++i) {
    if (type_of[d->z[i]] < 0) { type_of[d->z[i]] = (int)elems.size(); elems.push_back(d->z[i]); }
    type[i] = type_of[d->z[i]];
  }
  const int nt = (int)elems.size();
  std::vector<double> ref((size_t)nt * nt * MAXREF * MAXREF * 3), r0((size_t)nt * nt), rcov(n), r2r4(n);
  std::vector<int32_t> mxc(nt);
  for (int p = 0; p < nt; ++p) {
    mxc[p] = d->mxc[elems[p]];
    for (int q = 0; q < nt; ++q) {
      std::memcpy(&ref[((size_t)p * nt + q) * MAXREF * MAXREF * 3], &d->c6ref[((size_t)elems[p] * 95 + elems[q]) * MAXREF * MAXREF * 3],
                  sizeof(double) * MAXREF * MAXREF * 3);
      r0[(size_t)p * nt + q] = d->r0ab[(size_t)(elems[p] - 1) * 94 + elems[q] - 1] / AU_TO_ANG;
    }
  }
  for (int i = 0; i < n; ++i) { rcov[i] = d->rcov[d->z[i] - 1]; r2r4[i] = d->r2r4[d->z[i] - 1]; }
  bool ok = d->d_x.put(x, st) && d->d_tv.put(tv, st) && d->d_tc.put(tc, st) && d->d_rcov.put(rcov, st) && d->d_r2r4.put(r2r4, st) &&
            d->d_r0.put(r0, st) && d->d_ref.put(ref, st) && d->d_type.put(type, st) && d->d_mxc.put(mxc, st) &&
            d->d_cn.ensure(n) && d->d_e.ensure(n) &&
            d->d_f.ensure(3 * (size_t)n) && d->d_dedcn.ensure(n) && d->d_s.ensure(6 * (size_t)n);
  SNET_REQUIRE(ok, "snet_d3_compute: device allocation / upload failed");
  d3_cn_kernel<<<n, NTH, 0, st>>>(d->d_x.p, d->d_tc.p, n, Tc, zc, d->d_rcov.p, d->cn_cut, d->d_cn.p);
  const int t_chunks = std::max(1, std::min(Tv, (4 * NTH + n - 1) / n));   // >= 4 work items per thread where the images allow it
  d3_pair_kernel<<<n, NTH, 0, st>>>(d->d_x.p, d->d_tv.p, n, Tv, zv, d->d_r2r4.p, d->d_r0.p, d->d_type.p, nt, d->d_cn.p, d->d_mxc.p,
                                    d->d_ref.p, t_chunks, d->vdw_cut, d->func, d->d_e.p, d->d_f.p, d->d_dedcn.p, d->d_s.p);
  d3_cn_force_kernel<<<n, NTH, 0, st>>>(d->d_x.p, d->d_tc.p, n, Tc, zc, d->d_rcov.p, d->cn_cut, d->d_dedcn.p, d->d_f.p, d->d_s.p);
  SNET_CHECK_LAUNCH("snet_d3_compute");
  std::vector<double> e(n), s(6 * (size_t)n);
  d->forces.resize(3 * (size_t)n);
  d->cn.resize(n);
  ok = hipMemcpyAsync(e.data(), d->d_e.p, sizeof(double) * n, hipMemcpyDeviceToHost, st) == hipSuccess &&
       hipMemcpyAsync(s.data(), d->d_s.p, sizeof(double) * 6 * n, hipMemcpyDeviceToHost, st) == hipSuccess &&
       hipMemcpyAsync(d->forces.data(), d->d_f.p, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, st) == hipSuccess &&
       hipMemcpyAsync(d->cn.data(), d->d_cn.p, sizeof(double) * n, hipMemcpyDeviceToHost, st) == hipSuccess &&
       hipStreamSynchronize(st) == hipSuccess;
  SNET_REQUIRE(ok, "snet_d3_compute: readback failed");
  double et = 0.0, sv[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {   // fixed order: reproducible totals
    et += e[i];
    for (int q = 0; q < 6; ++q) sv[q] += s[6 * (size_t)i + q];
  }
  d->energy = et * AU_TO_EV;
  for (auto &v : d->forces) v *= AU_TO_EV / AU_TO_ANG;
  const double vol = std::fabs(det) * AU_TO_ANG * AU_TO_ANG * AU_TO_ANG, k = AU_TO_EV / vol;
  const double full[9] = {sv[0], sv[3], sv[4], sv[3], sv[1], sv[5], sv[4], sv[5], sv[2]};
  for (int q = 0; q < 9; ++q) d->stress[q] = full[q] * k;
  return 0;
}

double snet_d3_energy(const snet_d3 *d) { return d ? d->energy : 0.0; }
const double *snet_d3_forces(const snet_d3 *d) { return (d && !d->forces.empty()) ? d->forces.data() : nullptr; }
const double *snet_d3_stress(const snet_d3 *d) { return d ? d->stress : nullptr; }
const double *snet_d3_coordination_numbers(const snet_d3 *d) { return (d && !d->cn.empty()) ? d->cn.data() : nullptr; }

}  // extern "C"
