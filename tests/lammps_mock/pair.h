// MOCK (see lmp_mock_core.h): the members of LAMMPS_NS::Pair the glue touches
#pragma once
#include "lmp_mock_core.h"
namespace LAMMPS_NS {
class Pair : protected Pointers {
 public:
  explicit Pair(LAMMPS *);
  ~Pair() override;
  double eng_vdwl, eng_coul;
  double virial[6];
  double *eatom, **vatom;
  int comm_forward, comm_reverse;
  int single_enable, restartinfo, one_coeff, manybody_flag, no_virial_fdotr_compute;
  NeighList *list;
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
  int allocated;
  int **setflag;
  double **cutsq;
  int *map;
  int eflag_either, eflag_global, eflag_atom;
  int vflag_either, vflag_global, vflag_atom;
  void ev_init(int eflag, int vflag, int alloc = 1);
};
}  // namespace LAMMPS_NS
