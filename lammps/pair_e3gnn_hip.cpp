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
#include <mpi.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <numeric>
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
  if (halo) snet_halo_destroy(halo);
  if (host) snet_md_destroy(host);
  if (model) snet_model_destroy(model);
  if (rccl_comm) snet_rccl_comm_destroy(rccl_comm);
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
  if (snet_abi_version() != SNET_ABI_VERSION)
    error->all(FLERR, "e3gnn: libsnet_hip.so was built from a different snet_hip.h (ABI version " + std::to_string(snet_abi_version()) +
                          ", this pair style expects " + std::to_string(SNET_ABI_VERSION) + "); rebuild the library or the pair style");
  if (snet_model_load(arg[2], &model)) error->all(FLERR, std::string("e3gnn: ") + snet_last_error());
  if (snet_md_create(model, &host)) error->all(FLERR, std::string("e3gnn: ") + snet_last_error());

  char buf[4096];
  if (snet_model_meta(model, "model_type", buf, sizeof(buf)) || strcmp(buf, "E3_equivariant_model") != 0)
    error->all(FLERR, "given model type is not E3_equivariant_model");
  float rc = 0.f;
  int32_t n_species = 0, n_layers = 0;
  snet_model_info(model, &rc, &n_species, &n_layers, nullptr, 0);
  cutoff = rc;

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
    comm_forward = 2;  // (owner rank, owner's graph row), sent when the neighbor list was rebuilt
    // one RCCL communicator over the LAMMPS world: rank 0 makes the id, MPI carries its 128 bytes
    char id[128];
    if (comm->me == 0 && snet_rccl_unique_id(id)) error->one(FLERR, std::string("e3gnn/parallel: ") + snet_last_error());
    MPI_Bcast(id, 128, MPI_BYTE, 0, world);
    if (snet_rccl_comm_create(id, comm->nprocs, comm->me, &rccl_comm))
      error->one(FLERR, std::string("e3gnn/parallel: ") + snet_last_error());
  }
}

void PairE3GNNHip::init_style() {
  // the serial style aliases ghosts to the local atom with the same tag: an atom owned by another rank has no
  // such partner and its edges would silently vanish (reference: pair_e3gnn.cpp is serial-only as well)
  if (ghost_mode == 0 && comm->nprocs > 1)
    error->all(FLERR, "Pair style e3gnn runs on one MPI rank only; use e3gnn/parallel for domain decomposition");
  if (ghost_mode == 1 && force->newton_pair == 0) error->all(FLERR, "Pair style e3gnn/parallel requires newton pair on");
  if (ghost_mode == 1 && atom->tag_consecutive() == 0)
    error->all(FLERR, "Pair style e3gnn/parallel requires consecutive atom IDs");
  neighbor->add_request(this, NeighConst::REQ_FULL);  // many-body: full list (pair_e3gnn.cpp:423)
}

double PairE3GNNHip::init_one(int /*i*/, int /*j*/) { return cutoff; }

/* ---- ghost exchange plan (e3gnn/parallel), rebuilt with the neighbor list ----------------------------------
   1. snet_md_nodes: the graph nodes the engine will use -- owned atoms in ilist order, then one node per ghost
      identity owned elsewhere.
   2. every owned atom publishes (my rank, its graph row); one stock forward_comm (2 doubles per atom) carries
      that to all its ghost images, through however many swaps the brick decomposition needs.
   3. ghost nodes sorted by owner rank give the receive layout; the rows each owner has to send are exchanged
      with one MPI_Alltoall + MPI_Alltoallv of int32 indices.
   4. snet_halo_create + snet_model_set_rccl_halo: from here on every layer's exchange is one RCCL group inside
      snet_model_eval, device to device.  (Reference: 6 blocking swaps per exchange through comm_brick.cpp with
      optional host staging, pair_e3gnn_parallel.cpp:747-911.)                                                  */
void PairE3GNNHip::build_halo_plan() {
  const int nlocal = atom->nlocal, nall = atom->nlocal + atom->nghost;
  const int inum = list->inum, me = comm->me, np = comm->nprocs;
  node_to_atom.resize(nall);
  int64_t n_nodes = 0;
  if (snet_md_nodes(inum, list->ilist, nall, atom->tag, (int)sizeof(tagint), 1, node_to_atom.data(), &n_nodes))
    error->one(FLERR, std::string("e3gnn/parallel: ") + snet_last_error());
  n_ghost_nodes = (int)(n_nodes - inum);

  owner_info.assign((size_t)nall * 2, -1.0);
  for (int ii = 0; ii < inum; ii++) {
    const int i = list->ilist[ii];
    if (i < nlocal) {
      owner_info[2 * (size_t)i] = me;
      owner_info[2 * (size_t)i + 1] = ii;
    }
  }
  comm->forward_comm(this);

  // receive side: ghost nodes grouped by owner rank (stable), k-th row of that stream = ghost row perm[k]
  std::vector<int> owner(n_ghost_nodes), orow(n_ghost_nodes), order(n_ghost_nodes);
  for (int k = 0; k < n_ghost_nodes; k++) {
    const int a = node_to_atom[inum + k];
    owner[k] = (int)owner_info[2 * (size_t)a];
    orow[k] = (int)owner_info[2 * (size_t)a + 1];
    if (owner[k] < 0 || owner[k] >= np || owner[k] == me || orow[k] < 0)
      error->one(FLERR, "e3gnn/parallel: a ghost atom has no owner among the other ranks (is the ghost cutoff >= the model cutoff?)");
  }
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return owner[a] < owner[b]; });
  std::vector<int32_t> recv_counts(np, 0), send_counts(np, 0), recv_perm(n_ghost_nodes), want(n_ghost_nodes);
  for (int k = 0; k < n_ghost_nodes; k++) {
    recv_counts[owner[order[k]]]++;
    recv_perm[k] = order[k];
    want[k] = orow[order[k]];  // the owner's graph row, in the order the owner has to send them
  }
  MPI_Alltoall(recv_counts.data(), 1, MPI_INT, send_counts.data(), 1, MPI_INT, world);
  std::vector<int> rdisp(np + 1, 0), sdisp(np + 1, 0);
  for (int p = 0; p < np; p++) {
    rdisp[p + 1] = rdisp[p] + recv_counts[p];
    sdisp[p + 1] = sdisp[p] + send_counts[p];
  }
  std::vector<int32_t> send_idx(std::max(sdisp[np], 1));
  MPI_Alltoallv(want.data(), recv_counts.data(), rdisp.data(), MPI_INT, send_idx.data(), send_counts.data(), sdisp.data(),
                MPI_INT, world);

  if (halo) snet_halo_destroy(halo);
  halo = nullptr;
  if (snet_halo_create(rccl_comm, np, me, send_counts.data(), send_idx.data(), recv_counts.data(), recv_perm.data(), &halo))
    error->one(FLERR, std::string("e3gnn/parallel: ") + snet_last_error());
  // fold_forces 0: ghost forces stay in f[ghost], LAMMPS folds them with its own reverse_comm (newton on)
  if (snet_model_set_rccl_halo(model, halo, 0)) error->one(FLERR, std::string("e3gnn/parallel: ") + snet_last_error());
}

void PairE3GNNHip::compute(int eflag, int vflag) {
  if (eflag || vflag)   // (the reference's own idiom, pair_e3gnn.cpp:80-83 / pair_d3.cu:2000)
    ev_setup(eflag, vflag);
  else
    evflag = vflag_fdotr = 0;
  if (ghost_mode == 1 && vflag_atom) error->all(FLERR, "atomic stress is not supported\n");
  // an empty sub-domain: the reference's parallel style fails inside LibTorch there (docs/source/user_guide/lammps_torch.md:
  // 111-113: "encounters an error when one of the subdomain cells contains no atoms"); this one stops with a message that names
  // the remedy.  error->one, not error->all: only the empty rank gets here, the others are already inside the step's exchange.
  if (list->inum == 0)
    error->one(FLERR, "e3gnn: this MPI rank owns no atoms; every rank must own at least one "
                      "(use the `processors` command or `fix balance` so that no sub-domain is empty)");
  if (ghost_mode == 1 && (neighbor->ago == 0 || halo == nullptr)) build_halo_plan();

  const int nall = atom->nlocal + atom->nghost;
  node_to_atom.resize(nall);
  int64_t n_nodes = 0, n_edges = 0;
  double e = 0.0, v[6] = {0, 0, 0, 0, 0, 0};
  if (neighbor->ago > 0) snet_md_list_unchanged(host);   // no rebuild since the last step: the uploaded list is reused
  const int rc = snet_md_compute(host, list->inum, list->ilist, list->numneigh, list->firstneigh, nall, &atom->x[0][0],
                                 atom->type, atom->tag, (int)sizeof(tagint), map, atom->ntypes, ghost_mode, eflag_atom,
                                 vflag_either, vflag_atom, &atom->f[0][0], &e, v, eflag_atom ? eatom : nullptr,
                                 vflag_atom ? &vatom[0][0] : nullptr, node_to_atom.data(), &n_nodes, &n_edges, stream);
  if (rc) error->one(FLERR, std::string("e3gnn: ") + snet_last_error());
  if (ghost_mode == 1 && n_nodes - list->inum != n_ghost_nodes)
    error->one(FLERR, "e3gnn/parallel: the ghost set changed without a neighbor-list rebuild");
  if (eflag_global) eng_vdwl += e;
  if (vflag_global)
    for (int k = 0; k < 6; k++) virial[k] += v[k];
}

/* the only traffic through LAMMPS' comm: two doubles per atom when the neighbor list was rebuilt */
int PairE3GNNHip::pack_forward_comm(int n, int *list_, double *buf, int /*pbc_flag*/, int * /*pbc*/) {
  int m = 0;
  for (int i = 0; i < n; i++) {
    buf[m++] = owner_info[2 * (size_t)list_[i]];
    buf[m++] = owner_info[2 * (size_t)list_[i] + 1];
  }
  return m;
}

void PairE3GNNHip::unpack_forward_comm(int n, int first, double *buf) {
  int m = 0;
  for (int i = first; i < first + n; i++) {
    owner_info[2 * (size_t)i] = buf[m++];
    owner_info[2 * (size_t)i + 1] = buf[m++];
  }
}
