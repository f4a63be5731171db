// pair_style d3 over libsnet_hip.so (see pair_d3_hip.h).  Replaces sevenn/pair_e3gnn/pair_d3.cu:261-285 (settings),
// :644-656 (coeff: element symbols -> atomic numbers), :1970-2021 (compute / update: energy, forces, virial).
#include "pair_d3_hip.h"

#include <algorithm>
#include <cctype>
#include <cstring>

#include "atom.h"
#include "domain.h"
#include "error.h"
#include "memory.h"
#include "neighbor.h"
#include "snet_d3_ref.h"
#include "snet_hip.h"

using namespace LAMMPS_NS;

namespace {
const char *const SYMBOLS[] = {
    "h",  "he", "li", "be", "b",  "c",  "n",  "o",  "f",  "ne", "na", "mg", "al", "si", "p",  "s",  "cl", "ar", "k",  "ca",
    "sc", "ti", "v",  "cr", "mn", "fe", "co", "ni", "cu", "zn", "ga", "ge", "as", "se", "br", "kr", "rb", "sr", "y",  "zr",
    "nb", "mo", "tc", "ru", "rh", "pd", "ag", "cd", "in", "sn", "sb", "te", "i",  "xe", "cs", "ba", "la", "ce", "pr", "nd",
    "pm", "sm", "eu", "gd", "tb", "dy", "ho", "er", "tm", "yb", "lu", "hf", "ta", "w",  "re", "os", "ir", "pt", "au", "hg",
    "tl", "pb", "bi", "po", "at", "rn", "fr", "ra", "ac", "th", "pa", "u",  "np", "pu"};   // D3 is parameterised up to Z = 94

int atomic_number_of(std::string key) {
  std::transform(key.begin(), key.end(), key.begin(), [](unsigned char c) { return (char)std::tolower(c); });
  for (int z = 0; z < 94; ++z)
    if (key == SYMBOLS[z]) return z + 1;
  return -1;
}
}  // namespace

PairD3Hip::PairD3Hip(LAMMPS *lmp) : Pair(lmp) {
  single_enable = 0;
  restartinfo = 0;
  one_coeff = 1;
  manybody_flag = 1;
  no_virial_fdotr_compute = 1;   // the virial comes from the engine, not from f . r over ghosts
}

PairD3Hip::~PairD3Hip() {
  if (d3) pair_fin(d3);
  if (allocated) {
    memory->destroy(setflag);
    memory->destroy(cutsq);
  }
}

void PairD3Hip::allocate() {
  allocated = 1;
  const int n = atom->ntypes;
  memory->create(setflag, n + 1, n + 1, "pair:setflag");
  memory->create(cutsq, n + 1, n + 1, "pair:cutsq");
  for (int i = 1; i <= n; ++i)
    for (int j = 1; j <= n; ++j) setflag[i][j] = 0;
}

// pair_style d3 rthr cnthr damping functional          (pair_d3.cu:261-285)
void PairD3Hip::settings(int narg, char **arg) {
  if (narg != 4)
    error->all(FLERR, "Pair_style d3 needs Four arguments:\n"
                      "\t rthr: cutoff radius for dispersion interaction (a.u.^2)\n"
                      "\t cnthr: cutoff raius for coordination number (a.u.^2)\n"
                      "\t damping: name of the damping function (e.g., damp_zero, damp_bj)\n"
                      "\t functional: name of the functional (e.g., pbe, b3-lyp)\n");
  rthr = std::stod(arg[0]);
  cnthr = std::stod(arg[1]);
  damping = arg[2];
  functional = arg[3];
  if (damping != "damp_zero" && damping != "damp_bj")
    error->all(FLERR, "Unknown damping function (this build: damp_zero, damp_bj)");
}

// pair_coeff * * element1 element2 ...                   (pair_d3.cu:644-656)
void PairD3Hip::coeff(int narg, char **arg) {
  if (!allocated) allocate();
  const int ntypes = atom->ntypes;
  if (narg != ntypes + 2) error->all(FLERR, "Pair_coeff needs: * * element1 element2 ...");
  atomic_numbers.assign(ntypes, 0);
  for (int i = 0; i < ntypes; ++i) {
    atomic_numbers[i] = atomic_number_of(arg[i + 2]);
    if (atomic_numbers[i] < 1) error->all(FLERR, std::string("pair_style d3: unknown element ") + arg[i + 2]);
  }
  for (int i = 1; i <= ntypes; ++i)
    for (int j = 1; j <= ntypes; ++j) setflag[i][j] = 1;
  if (snet_abi_version() != SNET_ABI_VERSION) error->all(FLERR, "pair_style d3: libsnet_hip.so was built from a different snet_hip.h");
  if (!d3) d3 = pair_init();
  if (!d3 || pair_failed(d3)) error->all(FLERR, std::string("pair_style d3: the D3 engine of libsnet_hip.so could not be created: ") + snet_last_error());
  // functional / damping names and the parameter blob are resolved HERE, so that a typo or a missing data/d3_params.bin stops
  // the run at setup like the reference's error->all (pair_d3.cu:261-285, :303-640), not as zero dispersion at step one
  pair_run_settings(d3, rthr, cnthr, damping.c_str(), functional.c_str());
  if (pair_failed(d3)) error->all(FLERR, std::string("pair_style d3: ") + snet_last_error());
}

void PairD3Hip::init_style() {
  if (atomic_numbers.empty()) error->all(FLERR, "pair_style d3: pair_coeff must be set");
  neighbor->add_request(this, NeighConst::REQ_FULL);   // as the reference does (its list is not read either)
}

double PairD3Hip::init_one(int i, int j) {
  if (setflag[i][j] == 0) error->all(FLERR, "All pair coeffs are not set");
  return 0.0;   // no neighbor-list cutoff: the engine sums its own lattice translations (pair_d3.cu:2023-2028)
}

// energy, forces, virial of the whole periodic cell held by this process            (pair_d3.cu:1970-2021)
void PairD3Hip::compute(int eflag, int vflag) {
  if (eflag || vflag)   // (the reference's own idiom, pair_e3gnn.cpp:80-83 / pair_d3.cu:2000)
    ev_setup(eflag, vflag);
  else
    evflag = vflag_fdotr = 0;
  const int n = (int)atom->natoms;
  if (n != atom->nlocal) error->all(FLERR, "pair_style d3 is a one-process style (like the reference's CUDA version)");
  xflat.resize((size_t)n * 3);
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < 3; ++k) xflat[3 * (size_t)i + k] = atom->x[i][k];
  pair_set_atom(d3, n, atom->ntypes, atom->type, xflat.data());
  pair_set_domain(d3, domain->xperiodic, domain->yperiodic, domain->zperiodic, domain->boxlo, domain->boxhi, domain->xy,
                  domain->xz, domain->yz);
  // (settings were resolved in coeff(); the tables went to the device with the first pair_run_coeff: per step only the
  // positions, types and the box travel)
  pair_run_coeff(d3, atomic_numbers.data());
  pair_run_compute(d3);
  const double *f = pair_get_force(d3);
  const double *s = pair_get_stress(d3);
  if (pair_failed(d3) || !f || !s) error->all(FLERR, std::string("pair_style d3: ") + snet_last_error());
  if (eflag_global) eng_vdwl += pair_get_energy(d3);
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < 3; ++k) atom->f[i][k] += f[3 * (size_t)i + k];
  if (vflag_global)
    for (int k = 0; k < 6; ++k) virial[k] += s[k];   // already LAMMPS order xx yy zz xy xz yz (pair_d3.cu:1986-1993)
}
