// GPU neighbor list: the graph build that precedes the hot path.  Any mix of periodic and open axes,
// any cell height (round 4: thin cells and molecules no longer fall back to the host list).  Same conventions as the reference's host graph build (sevenn/train/dataload.py:32-79, ASE
// 'ijDS'): every ordered pair within the cutoff, no self edge, edge_vec = r_j - r_i + S.cell, computed
// in fp64 and stored as fp32 (sevenn/util.py:174-196).  Output is the CSR-by-center layout the
// convolution kernels consume, so no sort of the edge list is needed afterwards.
//
// Cell list in fractional coordinates: nb_k = max(1, floor(h_k / rc)) bins along lattice direction k (h_k =
// distance between opposite cell faces), searched over bin offsets -R_k .. R_k with R_k = ceil(rc / (h_k / nb_k)):
// R_k = 1 (the 27 surrounding bins cover the cutoff sphere, also for triclinic cells) whenever h_k >= rc; a cell
// THINNER than the cutoff has one bin and R_k = ceil(rc / h_k) > 1, i.e. it meets itself through several images.
// Every bin offset carries its own image shift floor((b + d) / nb), so images are distinct by construction.
// An OPEN (non-periodic) axis is binned over the fractional range the atoms span ([lo, hi], given by the caller),
// its atoms are not wrapped and offsets that leave the range are skipped (reference: the padded cell of
// sevenn/train/dataload.py:37-48 has the same effect).
#include "snet_common.h"

namespace {

struct Cell {
  double a[9];    // row-major lattice vectors
  double inv[9];  // inverse
  int nb[3];
  int R[3];       // bin offsets searched on each side
  int per[3];     // periodic axis?
  double lo[3], inv_ext[3];  // open axes: fractional lower bound and 1 / extent of the occupied range
  double rc2;
};

__global__ __launch_bounds__(256) void nl_bin_kernel(Cell C, const double *__restrict__ pos, int64_t n,
                                                     double *__restrict__ wpos, int32_t *__restrict__ wrap,
                                                     int32_t *__restrict__ cell_id) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x = pos[3 * i], y = pos[3 * i + 1], z = pos[3 * i + 2];
  double f[3];
  f[0] = x * C.inv[0] + y * C.inv[3] + z * C.inv[6];
  f[1] = x * C.inv[1] + y * C.inv[4] + z * C.inv[7];
  f[2] = x * C.inv[2] + y * C.inv[5] + z * C.inv[8];
  int b[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    int q;
    if (C.per[k]) {
      const double fl = floor(f[k]);
      wrap[3 * i + k] = (int)fl;
      f[k] -= fl;
      q = (int)(f[k] * C.nb[k]);
    } else {  // open axis: not wrapped, binned over the occupied range
      wrap[3 * i + k] = 0;
      q = (int)((f[k] - C.lo[k]) * C.inv_ext[k] * C.nb[k]);
      q = q < 0 ? 0 : q;
    }
    b[k] = q < C.nb[k] ? q : C.nb[k] - 1;
  }
  wpos[3 * i + 0] = f[0] * C.a[0] + f[1] * C.a[3] + f[2] * C.a[6];
  wpos[3 * i + 1] = f[0] * C.a[1] + f[1] * C.a[4] + f[2] * C.a[7];
  wpos[3 * i + 2] = f[0] * C.a[2] + f[1] * C.a[5] + f[2] * C.a[8];
  cell_id[i] = (b[0] * C.nb[1] + b[1]) * C.nb[2] + b[2];
}

// one lane per center atom; FILL = false counts, FILL = true writes src / edge_vec / shifts
template <bool FILL>
__global__ __launch_bounds__(128) void nl_pair_kernel(Cell C, const double *__restrict__ wpos,
                                                      const int32_t *__restrict__ wrap,
                                                      const int32_t *__restrict__ cell_id,
                                                      const int32_t *__restrict__ order,      // atoms sorted by bin
                                                      const int32_t *__restrict__ bin_start,  // [nbins+1]
                                                      int64_t n, int32_t *__restrict__ count,
                                                      const int32_t *__restrict__ row_ptr, int32_t *__restrict__ src,
                                                      int32_t *__restrict__ center, float *__restrict__ edge_vec,
                                                      int32_t *__restrict__ shifts) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double xi = wpos[3 * i], yi = wpos[3 * i + 1], zi = wpos[3 * i + 2];
  const int cid = cell_id[i];
  const int bz = cid % C.nb[2], by = (cid / C.nb[2]) % C.nb[1], bx = cid / (C.nb[2] * C.nb[1]);
  int cnt = 0;
  int64_t out = FILL ? row_ptr[i] : 0;
  auto fold = [](int c, int nb, int per, int &s) -> bool {  // bin index into [0, nb) with its image shift; false: outside an open axis
    s = 0;
    if (c >= 0 && c < nb) return true;
    if (!per) return false;
    s = c >= 0 ? c / nb : -((-c + nb - 1) / nb);
    return true;
  };
  for (int dx = -C.R[0]; dx <= C.R[0]; ++dx) {
    int cx = bx + dx, sx;
    if (!fold(cx, C.nb[0], C.per[0], sx)) continue;
    cx -= sx * C.nb[0];
    for (int dy = -C.R[1]; dy <= C.R[1]; ++dy) {
      int cy = by + dy, sy;
      if (!fold(cy, C.nb[1], C.per[1], sy)) continue;
      cy -= sy * C.nb[1];
      for (int dz = -C.R[2]; dz <= C.R[2]; ++dz) {
        int cz = bz + dz, sz;
        if (!fold(cz, C.nb[2], C.per[2], sz)) continue;
        cz -= sz * C.nb[2];
        const double ox = sx * C.a[0] + sy * C.a[3] + sz * C.a[6];
        const double oy = sx * C.a[1] + sy * C.a[4] + sz * C.a[7];
        const double oz = sx * C.a[2] + sy * C.a[5] + sz * C.a[8];
        const int nbin = (cx * C.nb[1] + cy) * C.nb[2] + cz;
        for (int k = bin_start[nbin]; k < bin_start[nbin + 1]; ++k) {
          const int j = order[k];
          const double ddx = wpos[3 * (int64_t)j] + ox - xi;
          const double ddy = wpos[3 * (int64_t)j + 1] + oy - yi;
          const double ddz = wpos[3 * (int64_t)j + 2] + oz - zi;
          const double d2 = ddx * ddx + ddy * ddy + ddz * ddz;
          if (d2 < C.rc2 && !(j == i && sx == 0 && sy == 0 && sz == 0)) {
            if (FILL) {
              src[out] = j;
              center[out] = (int)i;
              edge_vec[3 * out + 0] = (float)ddx;
              edge_vec[3 * out + 1] = (float)ddy;
              edge_vec[3 * out + 2] = (float)ddz;
              if (shifts) {  // relative to the caller's (unwrapped) positions
                shifts[3 * out + 0] = sx + wrap[3 * i + 0] - wrap[3 * (int64_t)j + 0];
                shifts[3 * out + 1] = sy + wrap[3 * i + 1] - wrap[3 * (int64_t)j + 1];
                shifts[3 * out + 2] = sz + wrap[3 * i + 2] - wrap[3 * (int64_t)j + 2];
              }
              ++out;
            } else {
              ++cnt;
            }
          }
        }
      }
    }
  }
  if (!FILL) count[i] = cnt;
}

int make_cell(const double *cell_host, double cutoff, const int32_t *pbc_host, const double *frac_range_host, Cell &C) {
  SNET_REQUIRE(cell_host != nullptr && cutoff > 0, "snet_nl: null cell / bad cutoff");
  const double *a = cell_host;
  for (int k = 0; k < 9; ++k) C.a[k] = a[k];
  const double det = a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) +
                     a[2] * (a[3] * a[7] - a[4] * a[6]);
  SNET_REQUIRE(fabs(det) > 1e-12, "snet_nl: singular cell");
  C.inv[0] = (a[4] * a[8] - a[5] * a[7]) / det;
  C.inv[1] = (a[2] * a[7] - a[1] * a[8]) / det;
  C.inv[2] = (a[1] * a[5] - a[2] * a[4]) / det;
  C.inv[3] = (a[5] * a[6] - a[3] * a[8]) / det;
  C.inv[4] = (a[0] * a[8] - a[2] * a[6]) / det;
  C.inv[5] = (a[2] * a[3] - a[0] * a[5]) / det;
  C.inv[6] = (a[3] * a[7] - a[4] * a[6]) / det;
  C.inv[7] = (a[1] * a[6] - a[0] * a[7]) / det;
  C.inv[8] = (a[0] * a[4] - a[1] * a[3]) / det;
  for (int k = 0; k < 3; ++k) {
    const double *u = a + 3 * ((k + 1) % 3), *v = a + 3 * ((k + 2) % 3);
    const double cx = u[1] * v[2] - u[2] * v[1], cy = u[2] * v[0] - u[0] * v[2], cz = u[0] * v[1] - u[1] * v[0];
    double h = fabs(det) / sqrt(cx * cx + cy * cy + cz * cz);   // distance between the opposite cell faces of axis k
    C.per[k] = pbc_host ? (pbc_host[k] != 0) : 1;
    C.lo[k] = 0.0;
    C.inv_ext[k] = 1.0;
    if (!C.per[k]) {
      SNET_REQUIRE(frac_range_host != nullptr, "snet_nl: an open axis needs the fractional range the atoms span");
      const double lo = frac_range_host[k], ext = frac_range_host[3 + k] - lo;
      SNET_REQUIRE(ext >= 0.0, "snet_nl: empty fractional range");
      C.lo[k] = lo;
      C.inv_ext[k] = ext > 1e-300 ? 1.0 / ext : 0.0;
      h *= ext;   // height of the occupied slab
    }
    int nb = (int)floor(h / cutoff);
    nb = nb < 1 ? 1 : (nb > 1024 ? 1024 : nb);
    C.nb[k] = nb;
    // bins on each side that cover the cutoff: 1 while the bin is at least one cutoff wide; a periodic cell thinner than the
    // cutoff (one bin) reaches ceil(rc / h) images of itself; an open axis has no images
    int R = 1;
    if (C.per[k] && h / nb < cutoff) {
      R = (int)ceil(cutoff / (h / nb));
      SNET_REQUIRE(R <= 64, "snet_nl: a periodic cell height is smaller than 1/64 of the cutoff");
    }
    C.R[k] = R;
  }
  C.rc2 = cutoff * cutoff;
  return 0;
}

}  // namespace

extern "C" int snet_nl_grid(const double *cell_host, double cutoff, const int32_t *pbc_host, const double *frac_range_host,
                             int32_t *nbins_host) {
  Cell C;
  if (int rc = make_cell(cell_host, cutoff, pbc_host, frac_range_host, C)) return rc;
  SNET_REQUIRE(nbins_host != nullptr, "snet_nl_grid: null output");
  nbins_host[0] = C.nb[0];
  nbins_host[1] = C.nb[1];
  nbins_host[2] = C.nb[2];
  return 0;
}

extern "C" int snet_nl_bin(const double *cell_host, double cutoff, const int32_t *pbc_host, const double *frac_range_host,
                           const double *pos, int64_t n, double *wpos, int32_t *wrap, int32_t *cell_id, void *stream) {
  Cell C;
  if (int rc = make_cell(cell_host, cutoff, pbc_host, frac_range_host, C)) return rc;
  if (n <= 0) return 0;
  nl_bin_kernel<<<(unsigned)((n + 255) / 256), 256, 0, static_cast<hipStream_t>(stream)>>>(C, pos, n, wpos, wrap,
                                                                                          cell_id);
  SNET_CHECK_LAUNCH("snet_nl_bin");
  return 0;
}

extern "C" int snet_nl_count(const double *cell_host, double cutoff, const int32_t *pbc_host, const double *frac_range_host,
                             const double *wpos, const int32_t *cell_id, const int32_t *order, const int32_t *bin_start,
                             int64_t n, int32_t *count, void *stream) {
  Cell C;
  if (int rc = make_cell(cell_host, cutoff, pbc_host, frac_range_host, C)) return rc;
  if (n <= 0) return 0;
  nl_pair_kernel<false><<<(unsigned)((n + 127) / 128), 128, 0, static_cast<hipStream_t>(stream)>>>(
      C, wpos, nullptr, cell_id, order, bin_start, n, count, nullptr, nullptr, nullptr, nullptr, nullptr);
  SNET_CHECK_LAUNCH("snet_nl_count");
  return 0;
}

extern "C" int snet_nl_fill(const double *cell_host, double cutoff, const int32_t *pbc_host, const double *frac_range_host,
                            const double *wpos, const int32_t *wrap, const int32_t *cell_id, const int32_t *order,
                            const int32_t *bin_start, int64_t n, const int32_t *row_ptr, int32_t *src, int32_t *center,
                            float *edge_vec, int32_t *shifts, void *stream) {
  Cell C;
  if (int rc = make_cell(cell_host, cutoff, pbc_host, frac_range_host, C)) return rc;
  if (n <= 0) return 0;
  nl_pair_kernel<true><<<(unsigned)((n + 127) / 128), 128, 0, static_cast<hipStream_t>(stream)>>>(
      C, wpos, wrap, cell_id, order, bin_start, n, nullptr, row_ptr, src, center, edge_vec, shifts);
  SNET_CHECK_LAUNCH("snet_nl_fill");
  return 0;
}
