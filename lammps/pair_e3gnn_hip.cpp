/* ----------------------------------------------------------------------
   LAMMPS pair styles e3gnn and e3gnn/parallel on the MI355X force engine.
   See pair_e3gnn_hip.h.  Replaces sevenn/pair_e3gnn/pair_e3gnn.cpp and
   pair_e3gnn_parallel.cpp (+ the comm_brick.cpp patch) of the reference.
------------------------------------------------------------------------- */
#include "pair_e3gnn_hip.h"

#include "atom.h"
#include "comm.h"
#include "error.h"
#include "force.h"
#include "memory.h"
#include "neigh_list.h"
#include "neighbor.h"

#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <string>

#include "snet_hip.h"

using namespace LAMMPS_NS;

PairE3GNNHip::PairE3GNNHip(LAMMPS *lmp) : Pair(lmp) {
  single_enable = 0;
  restartinfo = 0;
  one_coeff = 1;
  manybody_flag = 1;
  no_virial_fdotr_compute = 1;  // the engine returns the virial itself (pair_e3gnn.cpp:232-255)

  // one process per GPU: pick the device by node-local rank (reference: pair_e3gnn_parallel.cpp:75-118)
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0)
    error->all(FLERR, "pair e3gnn: no ROCm device visible (the HIP engine has no CPU path)");
  const char *lr = std::getenv("OMPI_COMM_WORLD_LOCAL_RANK");
  if (!lr) lr = std::getenv("MV2_COMM_WORLD_LOCAL_RANK");
  if (!lr) lr = std::getenv("SLURM_LOCALID");
  const int local_rank = lr ? std::atoi(lr) : comm->me;
  if (hipSetDevice(local_rank % n_dev) != hipSuccess) error->all(FLERR, "pair e3gnn: hipSetDevice failed");
  hipStream_t st;
  if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess)
    error->all(FLERR, "pair e3gnn: cannot create a HIP stream");
  stream = st;
}

PairE3GNNHip::~PairE3GNNHip() {
  if (host) snet_md_destroy(host);
  if (model) snet_model_destroy(model);
  if (stream) (void)hipStreamDestroy(static_cast<hipStream_t>(stream));
  if (allocated) {
    memory->destroy(setflag);
    memory->destroy(cutsq);
    memory->destroy(map);
  }
}

void PairE3GNNHip::allocate() {
  allocated = 1;
  const int n = atom->ntypes;
  memory->create(setflag, n + 1, n + 1, "pair:setflag");
  memory->create(cutsq, n + 1, n + 1, "pair:cutsq");
  memory->create(map, n + 1, "pair:map");
  for (int i = 0; i <= n; i++) {
    map[i] = -1;
    for (int j = 0; j <= n; j++) setflag[i][j] = 0;
  }
}

void PairE3GNNHip::settings(int narg, char ** /*arg*/) {
  if (narg != 0) error->all(FLERR, "Illegal pair_style command");
}

// pair_coeff * * model.snet <element per atom type>
void PairE3GNNHip::coeff(int narg, char **arg) {
  if (allocated) error->all(FLERR, "pair e3gnn: pair_coeff called twice");
  allocate();
  if (narg < 3 || strcmp(arg[0], "*") != 0 || strcmp(arg[1], "*") != 0)
    error->all(FLERR, "e3gnn: first and second input of pair_coeff should be '*'");
  if (snet_model_load(arg[2], &model)) error->all(FLERR, std::string("e3gnn: ") + snet_last_error());
  if (snet_md_create(model, &host)) error->all(FLERR, std::string("e3gnn: ") + snet_last_error());

  char buf[4096];
  if (snet_model_meta(model, "model_type", buf, sizeof(buf)) || strcmp(buf, "E3_equivariant_model") != 0)
    error->all(FLERR, "given model type is not E3_equivariant_model");
  float rc = 0.f;
  int32_t n_species = 0, n_layers = 0, comm_dims[64];
  snet_model_info(model, &rc, &n_species, &n_layers, comm_dims, 64);
  cutoff = rc;
  max_comm_dim = 3;
  for (int t = 1; t < n_layers && t < 64; t++) max_comm_dim = comm_dims[t] > max_comm_dim ? comm_dims[t] : max_comm_dim;

  if (snet_model_meta(model, "chemical_symbols_to_index", buf, sizeof(buf))) error->all(FLERR, snet_last_error());
  std::vector<std::string> symbols;
  for (char *tok = std::strtok(buf, " "); tok; tok = std::strtok(nullptr, " ")) symbols.emplace_back(tok);

  const int ntypes = atom->ntypes;
  if (ntypes > narg - 3)
    error->all(FLERR, "Not enough chemical specie is given. Check pair_coeff and types in your data/script");
  for (int i = 3; i < narg && i - 2 <= ntypes; i++) {
    int found = -1;
    for (size_t j = 0; j < symbols.size(); j++)
      if (symbols[j] == arg[i]) found = (int)j;
    if (found < 0) error->all(FLERR, "Unknown chemical specie is given");
    map[i - 2] = found;
    if (comm->me == 0) utils::logmesg(lmp, "Chemical specie '{}' is assigned to type {}\n", arg[i], i - 2);
  }
  for (int i = 1; i <= ntypes; i++)
    for (int j = 1; j <= ntypes; j++)
      if (map[i] >= 0 && map[j] >= 0) {
        setflag[i][j] = 1;
        cutsq[i][j] = cutoff * cutoff;
      }

  if (ghost_mode == 1) {
    comm_forward = max_comm_dim;  // one feature row per atom and exchange
    comm_reverse = max_comm_dim;
    if (snet_model_set_halo(model, &PairE3GNNHip::halo_forward, &PairE3GNNHip::halo_reverse, this, 0))
      error->all(FLERR, snet_last_error());
  }
}

void PairE3GNNHip::init_style() {
  if (ghost_mode == 1 && force->newton_pair == 0) error->all(FLERR, "Pair style e3gnn/parallel requires newton pair on");
  neighbor->add_request(this, NeighConst::REQ_FULL);  // many-body: full list (pair_e3gnn.cpp:423)
}

double PairE3GNNHip::init_one(int /*i*/, int /*j*/) { return cutoff; }

void PairE3GNNHip::compute(int eflag, int vflag) {
  ev_init(eflag, vflag);
  if (ghost_mode == 1 && vflag_atom) error->all(FLERR, "atomic stress is not supported\n");

  const int nall = atom->nlocal + atom->nghost;
  node_to_atom.resize(nall);
  int64_t n_nodes = 0, n_edges = 0;
  double e = 0.0, v[6] = {0, 0, 0, 0, 0, 0};
  const int rc = snet_md_compute(host, list->inum, list->ilist, list->numneigh, list->firstneigh, nall, &atom->x[0][0],
                                 atom->type, atom->tag, (int)sizeof(tagint), map, atom->ntypes, ghost_mode, eflag_atom,
                                 vflag_either, vflag_atom, &atom->f[0][0], &e, v, eflag_atom ? eatom : nullptr,
                                 vflag_atom ? &vatom[0][0] : nullptr, node_to_atom.data(), &n_nodes, &n_edges, stream);
  if (rc) error->one(FLERR, std::string("e3gnn: ") + snet_last_error());
  if (eflag_global) eng_vdwl += e;
  if (vflag_global)
    for (int k = 0; k < 6; k++) virial[k] += v[k];
}

/* ---- ghost-node feature exchange through LAMMPS' comm ---------------------------------------
   The engine hands a DEVICE matrix x[n_nodes, dim].  Rows are staged to the host by atom index,
   exchanged with comm->forward_comm(this) / reverse_comm(this), and staged back.  Several ghost
   atoms can be images of one identity (one graph node): forward copies any of them (all equal);
   reverse places the node's gradient on node_to_atom[node] only, so it is summed once.        */
int PairE3GNNHip::halo_forward(void *self, float *x_dev, int64_t n_total, int64_t n_local, int32_t dim, void *st) {
  auto *p = static_cast<PairE3GNNHip *>(self);
  const int nall = p->atom->nlocal + p->atom->nghost;
  p->row_dim = dim;
  p->rows.assign((size_t)nall * dim, 0.f);
  p->stage.resize((size_t)n_total * dim);
  if (hipMemcpyAsync(p->stage.data(), x_dev, (size_t)n_local * dim * 4, hipMemcpyDeviceToHost, (hipStream_t)st) != hipSuccess ||
      hipStreamSynchronize((hipStream_t)st) != hipSuccess)
    return 1;
  for (int64_t g = 0; g < n_local; g++)
    memcpy(&p->rows[(size_t)p->node_to_atom[g] * dim], &p->stage[(size_t)g * dim], (size_t)dim * 4);
  p->comm->forward_comm(p);
  for (int64_t g = n_local; g < n_total; g++)
    memcpy(&p->stage[(size_t)g * dim], &p->rows[(size_t)p->node_to_atom[g] * dim], (size_t)dim * 4);
  if (hipMemcpyAsync(x_dev + n_local * dim, p->stage.data() + n_local * dim, (size_t)(n_total - n_local) * dim * 4,
                     hipMemcpyHostToDevice, (hipStream_t)st) != hipSuccess)
    return 1;
  return hipStreamSynchronize((hipStream_t)st) != hipSuccess;  // `stage` is reused by the next exchange
}

int PairE3GNNHip::halo_reverse(void *self, float *x_dev, int64_t n_total, int64_t n_local, int32_t dim, void *st) {
  auto *p = static_cast<PairE3GNNHip *>(self);
  const int nall = p->atom->nlocal + p->atom->nghost;
  p->row_dim = dim;
  p->rows.assign((size_t)nall * dim, 0.f);
  p->stage.resize((size_t)n_total * dim);
  if (hipMemcpyAsync(p->stage.data(), x_dev, (size_t)n_total * dim * 4, hipMemcpyDeviceToHost, (hipStream_t)st) != hipSuccess ||
      hipStreamSynchronize((hipStream_t)st) != hipSuccess)
    return 1;
  for (int64_t g = 0; g < n_total; g++)
    memcpy(&p->rows[(size_t)p->node_to_atom[g] * dim], &p->stage[(size_t)g * dim], (size_t)dim * 4);
  p->comm->reverse_comm(p);  // ghost rows are added into their owners (unpack_reverse_comm)
  for (int64_t g = 0; g < n_local; g++)
    memcpy(&p->stage[(size_t)g * dim], &p->rows[(size_t)p->node_to_atom[g] * dim], (size_t)dim * 4);
  if (hipMemcpyAsync(x_dev, p->stage.data(), (size_t)n_local * dim * 4, hipMemcpyHostToDevice, (hipStream_t)st) != hipSuccess)
    return 1;
  return hipStreamSynchronize((hipStream_t)st) != hipSuccess;
}

int PairE3GNNHip::pack_forward_comm(int n, int *list_, double *buf, int /*pbc_flag*/, int * /*pbc*/) {
  int m = 0;
  for (int i = 0; i < n; i++) {
    const float *r = &rows[(size_t)list_[i] * row_dim];
    for (int k = 0; k < row_dim; k++) buf[m++] = r[k];
  }
  return m;
}

void PairE3GNNHip::unpack_forward_comm(int n, int first, double *buf) {
  int m = 0;
  for (int i = first; i < first + n; i++) {
    float *r = &rows[(size_t)i * row_dim];
    for (int k = 0; k < row_dim; k++) r[k] = (float)buf[m++];
  }
}

int PairE3GNNHip::pack_reverse_comm(int n, int first, double *buf) {
  int m = 0;
  for (int i = first; i < first + n; i++) {
    const float *r = &rows[(size_t)i * row_dim];
    for (int k = 0; k < row_dim; k++) buf[m++] = r[k];
  }
  return m;
}

void PairE3GNNHip::unpack_reverse_comm(int n, int *list_, double *buf) {
  int m = 0;
  for (int i = 0; i < n; i++) {
    float *r = &rows[(size_t)list_[i] * row_dim];
    for (int k = 0; k < row_dim; k++) r[k] += (float)buf[m++];
  }
}
