/* -*- c++ -*- ----------------------------------------------------------
   LAMMPS pair styles backed by libsnet_hip.so (MI355X force engine).

   pair_style e3gnn            (one process holds the periodic cell; ghosts aliased by tag)
   pair_style e3gnn/parallel   (spatial decomposition; ghost atoms are graph nodes, their
                                features travel through LAMMPS' own forward/reverse comm)

   Same style names and pair_coeff grammar as the reference's TorchScript-backed styles
   (sevenn/pair_e3gnn/pair_e3gnn.h:15, pair_e3gnn_parallel.h:15):
       pair_coeff * * <model.snet> <element of type 1> <element of type 2> ...
   The model file is written by `python -m sevennet_amd.deploy` (replaces `sevenn get_model`).
   Unlike the reference, e3gnn/parallel needs NO patched comm_brick.cpp: it uses the stock
   Pair::pack_forward_comm / pack_reverse_comm hooks, and one model file instead of one
   TorchScript segment per layer.

   NOT COMPILED in the development image (no LAMMPS tree / MPI there).  Everything below the
   LAMMPS API surface -- graph build from the neighbor list, evaluation, force/virial
   accumulation, ghost-node exchange hooks -- is snet_md_compute(), which IS tested
   (tests/test_md_host_gpu.py drives it with the same arrays this file passes).
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

  // ghost-node feature exchange (e3gnn/parallel only): stock LAMMPS comm hooks
  int pack_forward_comm(int, int *, double *, int, int *) override;
  void unpack_forward_comm(int, int, double *) override;
  int pack_reverse_comm(int, int, double *) override;
  void unpack_reverse_comm(int, int *, double *) override;

 protected:
  int ghost_mode = 0;  // 0: e3gnn, 1: e3gnn/parallel
  double cutoff = 0.0;
  snet_model *model = nullptr;
  snet_md_host *host = nullptr;
  void *stream = nullptr;  // hipStream_t

  // host staging of one feature exchange, addressed by LAMMPS atom index
  std::vector<float> rows;           // [nall, row_dim]
  std::vector<float> stage;          // pinned-size staging of device rows [n_nodes, row_dim]
  std::vector<int> node_to_atom;     // graph node -> atom index (filled by snet_md_compute)
  int row_dim = 0;
  int max_comm_dim = 0;

  void allocate();
  static int halo_forward(void *self, float *x_dev, int64_t n_total, int64_t n_local, int32_t dim, void *stream);
  static int halo_reverse(void *self, float *x_dev, int64_t n_total, int64_t n_local, int32_t dim, void *stream);
};

class PairE3GNNHipParallel : public PairE3GNNHip {
 public:
  PairE3GNNHipParallel(class LAMMPS *lmp) : PairE3GNNHip(lmp) { ghost_mode = 1; }
};

}  // namespace LAMMPS_NS

#endif
#endif
