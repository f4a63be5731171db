// MOCK (see lmp_mock_core.h): the members of LAMMPS_NS::Pair the glue touches
#pragma once
#include "lmp_mock_core.h"
namespace LAMMPS_NS {
class Pair : protected Pointers {
 public:
  explicit Pair(LAMMPS *);
  ~Pair() override;
  double eng_vdwl = 0.0, eng_coul = 0.0;
  double virial[6] = {0, 0, 0, 0, 0, 0};
  double *eatom = nullptr, **vatom = nullptr;
  int comm_forward = 0, comm_reverse = 0;
  int single_enable = 1, restartinfo = 1, one_coeff = 0, manybody_flag = 0, no_virial_fdotr_compute = 0;
  NeighList *list = nullptr;
  virtual void compute(int, int) = 0;
  virtual void settings(int, char **) = 0;
  virtual void coeff(int, char **) = 0;
  virtual void init_style();
  virtual double init_one(int, int) { return 0.0; }
  virtual int pack_forward_comm(int, int *, double *, int, int *) { return 0; }
  virtual void unpack_forward_comm(int, int, double *) {}
  virtual int pack_reverse_comm(int, int, double *) { return 0; }
  virtual void unpack_reverse_comm(int, int *, double *) {}

 protected:
  int allocated = 0;
  int **setflag = nullptr;
  double **cutsq = nullptr;
  int *map = nullptr;
  int eflag_either = 0, eflag_global = 0, eflag_atom = 0;
  int vflag_either = 0, vflag_global = 0, vflag_atom = 0;
  // LAMMPS' flag encoding (pair.h / integrate.cpp): eflag 1 global, 2 per atom; vflag 1 | 2 global, 4 per atom
  void ev_setup(int eflag, int vflag, int alloc = 1);
  int evflag = 0, vflag_fdotr = 0;
  int maxeatom = 0, maxvatom = 0;
};
}  // namespace LAMMPS_NS
