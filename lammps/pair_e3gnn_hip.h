/* -*- c++ -*- ----------------------------------------------------------
   LAMMPS pair styles backed by libsnet_hip.so (MI355X force engine).

   pair_style e3gnn            (one process holds the periodic cell; ghosts aliased by tag)
   pair_style e3gnn/parallel   (spatial decomposition; ghost atoms are graph nodes, their features
                                travel GPU to GPU over RCCL inside the engine)

   Same style names and pair_coeff grammar as the reference's TorchScript-backed styles
   (sevenn/pair_e3gnn/pair_e3gnn.h:15, pair_e3gnn_parallel.h:15):
       pair_coeff * * <model.snet> <element of type 1> <element of type 2> ...
   The model file is written by `python -m sevennet_amd.deploy` (replaces `sevenn get_model`).
   Unlike the reference, e3gnn/parallel needs NO patched comm_brick.cpp, no LibTorch and no CUDA-aware
   MPI: the per-layer ghost exchange is the engine's own RCCL send/recv group (snet_halo_*,
   csrc/snet_halo.cpp) -- device to device over xGMI.  LAMMPS' comm is used once per neighbor-list
   rebuild, with two doubles per atom, to learn which rank owns each ghost and under which row.

   NOT COMPILED AGAINST LAMMPS in the development image (no LAMMPS tree / MPI there); tests/test_lammps_glue_cpu.py
   only checks that both files parse and type-check against a small mock of the LAMMPS API subset they
   use (tests/lammps_mock/).  Everything below that API -- graph build from the neighbor list,
   evaluation, force / virial accumulation (snet_md_compute) and the ghost exchange (snet_halo_*) -- IS
   tested on the GPU (tests/test_md_host_gpu.py, tests/test_engine_gpu.py).
------------------------------------------------------------------------- */
#ifdef PAIR_CLASS
// clang-format off
PairStyle(e3gnn, PairE3GNNHip)
PairStyle(e3gnn/parallel, PairE3GNNHipParallel)
// clang-format on
#else

#ifndef LMP_PAIR_E3GNN_HIP
#define LMP_PAIR_E3GNN_HIP

#include "pair.h"

#include <vector>

struct snet_model;
struct snet_md_host;
struct snet_halo;

namespace LAMMPS_NS {

class PairE3GNNHip : public Pair {
 public:
  PairE3GNNHip(class LAMMPS *);
  ~PairE3GNNHip() override;
  void compute(int, int) override;
  void settings(int, char **) override;
  void coeff(int, char **) override;
  void init_style() override;
  double init_one(int, int) override;

  // e3gnn/parallel, once per neighbor-list rebuild: (owner rank, owner's graph row) of every atom -> its ghosts
  int pack_forward_comm(int, int *, double *, int, int *) override;
  void unpack_forward_comm(int, int, double *) override;

 protected:
  int ghost_mode = 0;  // 0: e3gnn, 1: e3gnn/parallel
  double cutoff = 0.0;
  snet_model *model = nullptr;
  snet_md_host *host = nullptr;
  snet_halo *halo = nullptr;
  void *rccl_comm = nullptr;  // created by libsnet_hip.so (snet_rccl_comm_create) from an id broadcast over MPI
  void *stream = nullptr;     // hipStream_t

  std::vector<int> node_to_atom;     // graph node -> atom index
  std::vector<double> owner_info;    // [nall][2]: owning rank, graph row on that rank
  int n_ghost_nodes = 0;

  void allocate();
  void build_halo_plan();
};

class PairE3GNNHipParallel : public PairE3GNNHip {
 public:
  PairE3GNNHipParallel(class LAMMPS *lmp) : PairE3GNNHip(lmp) { ghost_mode = 1; }
};

}  // namespace LAMMPS_NS

#endif
#endif
