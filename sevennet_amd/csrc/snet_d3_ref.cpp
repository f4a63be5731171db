// The reference's own D3 binding, verbatim: the ten `pair_*` entry points its ctypes stub declares
// (sevenn/calculator.py:430-483) with the signatures of sevenn/pair_e3gnn/pair_d3_for_ase.cu:2034-2082, as shims over
// this library's snet_d3_* engine (csrc/snet_d3.hip).  A maintainer can point the reference's `_load('pair_d3')` at
// libsnet_hip.so and keep D3Calculator unchanged.
//
// What the reference's library carries as compiled-in tables travels here as one binary blob next to the library
// (`data/d3_params.bin`, written by sevennet_amd/build.py from data/d3_params.npz; SNET_D3_PARAMS overrides the path):
//   "SNETD3P1" | int64 n_c6 | r0ab[94*94] c6ab[n_c6*5] r2r4[94] rcov[94] (float64)
//   | int32 n_sets | per set: int32 damping id (0 zero, 1 bj), int32 n, n x { char name[32]; double p[5] }
// Conventions of the reference kept: positions / forces in the caller's (LAMMPS-rotated) frame, box as boxlo / boxhi /
// xy xz yz, `pair_get_stress` = the six virial sums (xx yy zz xy xz yz; the caller divides by -volume), types 1-based.
#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "snet_common.h"

namespace {

struct D3Params {
  bool ok = false;
  std::string error;
  std::vector<double> r0ab, c6ab, r2r4, rcov;
  int64_t n_c6 = 0;
  std::map<std::string, std::vector<double>> func[2];  // per damping id: functional name -> (s6, rs6, s18, rs18, alp)
};

std::string default_blob_path() {
  if (const char *e = getenv("SNET_D3_PARAMS")) return e;
  Dl_info info;
  if (dladdr(reinterpret_cast<const void *>(&default_blob_path), &info) && info.dli_fname) {
    std::string p = info.dli_fname;
    const size_t s = p.find_last_of('/');
    return (s == std::string::npos ? std::string(".") : p.substr(0, s)) + "/data/d3_params.bin";
  }
  return "data/d3_params.bin";
}

const D3Params &params() {
  static D3Params P;
  static std::once_flag once;
  std::call_once(once, [] {
    const std::string path = default_blob_path();
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) {
      P.error = "pair_d3: cannot open the D3 parameter blob " + path + " (run `python -m sevennet_amd.build`, or set SNET_D3_PARAMS)";
      return;
    }
    auto rd = [&](void *dst, size_t n) { return fread(dst, 1, n, f) == n; };
    char magic[8];
    bool good = rd(magic, 8) && memcmp(magic, "SNETD3P1", 8) == 0 && rd(&P.n_c6, 8) && P.n_c6 > 0 && P.n_c6 < (1 << 24);
    if (good) {
      P.r0ab.resize(94 * 94); P.c6ab.resize((size_t)P.n_c6 * 5); P.r2r4.resize(94); P.rcov.resize(94);
      good = rd(P.r0ab.data(), P.r0ab.size() * 8) && rd(P.c6ab.data(), P.c6ab.size() * 8) && rd(P.r2r4.data(), 94 * 8) &&
             rd(P.rcov.data(), 94 * 8);
    }
    int32_t n_sets = 0;
    good = good && rd(&n_sets, 4) && n_sets >= 0 && n_sets <= 8;
    for (int s = 0; good && s < n_sets; ++s) {
      int32_t damping = -1, n = 0;
      good = rd(&damping, 4) && rd(&n, 4) && n >= 0 && n < 4096;
      for (int i = 0; good && i < n; ++i) {
        char name[32];
        std::vector<double> p(5);
        good = rd(name, 32) && rd(p.data(), 40);
        name[31] = 0;
        if (good && (damping == 0 || damping == 1)) P.func[damping][name] = p;
      }
    }
    fclose(f);
    if (!good) P.error = "pair_d3: malformed D3 parameter blob " + path;
    P.ok = good;
  });
  return P;
}

}  // namespace

struct PairD3 {  // the reference's opaque handle type name
  snet_d3 *h = nullptr;
  int natoms = 0, ntypes = 0;
  std::vector<int> type;
  std::vector<double> x;
  double cell[9] = {0};
  int pbc[3] = {1, 1, 1};
  bool have_domain = false, have_settings = false, failed = false, tables_set = false;
  // last arguments of pair_run_settings: the reference's callers repeat the call every step with the same values
  double set_rthr = 0.0, set_cnthr = 0.0;
  std::string set_damp, set_func;
  double result_E = 0.0;
  std::vector<double> result_F;
  double result_S[6] = {0, 0, 0, 0, 0, 0};
};

namespace {
void fail(PairD3 *p, const std::string &what) {
  // the reference's functions return void and abort on CUDA errors; here a failure is sticky, printed once, readable through
  // snet_last_error() / pair_failed(), and the getters stop handing out results: pair_get_force / pair_get_stress return NULL
  // and pair_get_energy NaN while it is set, so no caller can keep running on zero dispersion
  if (p) p->failed = true;
  snet::set_error(what);
  fprintf(stderr, "%s\n", what.c_str());
}
}  // namespace

extern "C" {

PairD3 *pair_init() {
  auto *p = new PairD3;
  if (snet_d3_create(&p->h)) fail(p, std::string("pair_init: ") + snet_last_error());
  return p;
}

void pair_set_atom(PairD3 *pair, int natoms, int ntypes, int *type, double *x_flat) {
  if (!pair) return;
  if (natoms <= 0 || ntypes <= 0 || !type || !x_flat) return fail(pair, "pair_set_atom: bad argument");
  pair->natoms = natoms;
  pair->ntypes = ntypes;
  pair->type.assign(type, type + natoms);
  pair->x.assign(x_flat, x_flat + (size_t)natoms * 3);
  pair->result_F.assign((size_t)natoms * 3, 0.0);
}

void pair_set_domain(PairD3 *pair, int xperiodic, int yperiodic, int zperiodic, double *boxlo, double *boxhi, double xy,
                     double xz, double yz) {
  if (!pair) return;
  if (!boxlo || !boxhi) return fail(pair, "pair_set_domain: null box");
  // LAMMPS restricted triclinic box -> lattice vectors as rows: a = (lx, 0, 0), b = (xy, ly, 0), c = (xz, yz, lz)
  const double lx = boxhi[0] - boxlo[0], ly = boxhi[1] - boxlo[1], lz = boxhi[2] - boxlo[2];
  const double c9[9] = {lx, 0, 0, xy, ly, 0, xz, yz, lz};
  memcpy(pair->cell, c9, sizeof c9);
  pair->pbc[0] = xperiodic != 0; pair->pbc[1] = yperiodic != 0; pair->pbc[2] = zperiodic != 0;
  pair->have_domain = true;
}

void pair_run_settings(PairD3 *pair, double rthr, double cnthr, const char *damp_name, const char *func_name) {
  if (!pair || pair->failed) return;
  const D3Params &P = params();
  if (!P.ok) return fail(pair, P.error);
  if (!damp_name || !func_name) return fail(pair, "pair_run_settings: null name");
  if (pair->have_settings && rthr == pair->set_rthr && cnthr == pair->set_cnthr && pair->set_damp == damp_name && pair->set_func == func_name)
    return;   // unchanged since the last call
  int damping = -1;
  if (strcmp(damp_name, "damp_zero") == 0) damping = 0;
  else if (strcmp(damp_name, "damp_bj") == 0) damping = 1;
  else return fail(pair, std::string("pair_run_settings: unsupported damping ") + damp_name +
                             " (damp_zero / damp_bj; the reference's damp_zerom / damp_bjm kernels are empty too)");
  auto it = P.func[damping].find(func_name);
  if (it == P.func[damping].end()) return fail(pair, std::string("pair_run_settings: functional name unknown: ") + func_name);
  if (snet_d3_settings(pair->h, rthr, cnthr, damping, it->second.data())) return fail(pair, snet_last_error());
  pair->have_settings = true;
  pair->set_rthr = rthr; pair->set_cnthr = cnthr; pair->set_damp = damp_name; pair->set_func = func_name;
}

void pair_run_coeff(PairD3 *pair, int *atomic_numbers) {
  if (!pair || pair->failed) return;
  const D3Params &P = params();
  if (!P.ok) return fail(pair, P.error);
  if (!atomic_numbers || pair->natoms <= 0 || !pair->have_domain) return fail(pair, "pair_run_coeff: call pair_set_atom and pair_set_domain first");
  std::vector<int32_t> z((size_t)pair->natoms);
  for (int i = 0; i < pair->natoms; ++i) {
    const int t = pair->type[i];
    if (t < 1 || t > pair->ntypes) return fail(pair, "pair_run_coeff: atom type out of range (types are 1-based)");
    z[i] = atomic_numbers[t - 1];
  }
  // the parameter tables go to the device once per handle, not once per step
  if (!pair->tables_set) {
    if (snet_d3_set_tables(pair->h, P.r0ab.data(), P.c6ab.data(), P.n_c6, P.r2r4.data(), P.rcov.data())) return fail(pair, snet_last_error());
    pair->tables_set = true;
  }
  if (snet_d3_set_atoms(pair->h, pair->natoms, z.data(), pair->x.data()) || snet_d3_set_cell(pair->h, pair->cell, pair->pbc))
    return fail(pair, snet_last_error());
}

void pair_run_compute(PairD3 *pair) {
  if (!pair || pair->failed) return;
  if (!pair->have_settings) return fail(pair, "pair_run_compute: call pair_run_settings first");
  if (snet_d3_compute(pair->h, nullptr)) return fail(pair, snet_last_error());
  pair->result_E = snet_d3_energy(pair->h);
  const double *f = snet_d3_forces(pair->h), *s = snet_d3_stress(pair->h);
  if (!f || !s) return fail(pair, "pair_run_compute: no results");
  pair->result_F.assign(f, f + (size_t)pair->natoms * 3);
  // snet_d3_stress = dE/d(strain) / V (ASE sign); the reference returns the virial sums W = -V * stress, order xx yy zz xy xz yz
  const double *c = pair->cell;
  const double vol = c[0] * (c[4] * c[8] - c[5] * c[7]) - c[1] * (c[3] * c[8] - c[5] * c[6]) + c[2] * (c[3] * c[7] - c[4] * c[6]);
  const double v = vol < 0 ? -vol : vol;
  pair->result_S[0] = -v * s[0]; pair->result_S[1] = -v * s[4]; pair->result_S[2] = -v * s[8];
  pair->result_S[3] = -v * s[1]; pair->result_S[4] = -v * s[2]; pair->result_S[5] = -v * s[5];
}

double pair_get_energy(PairD3 *pair) { return (pair && !pair->failed) ? pair->result_E : __builtin_nan(""); }

double *pair_get_force(PairD3 *pair) { return (pair && !pair->failed && !pair->result_F.empty()) ? pair->result_F.data() : nullptr; }

double *pair_get_stress(PairD3 *pair) { return (pair && !pair->failed) ? pair->result_S : nullptr; }

// extension (not in the reference's ABI): 1 once any call on this handle has failed; the message is snet_last_error()
int pair_failed(PairD3 *pair) { return (!pair || pair->failed) ? 1 : 0; }

void pair_fin(PairD3 *pair) {
  if (!pair) return;
  snet_d3_destroy(pair->h);
  delete pair;
}

}  // extern "C"
