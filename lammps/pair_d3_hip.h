/* -*- c++ -*- ----------------------------------------------------------
   LAMMPS pair style `d3` backed by libsnet_hip.so: Grimme D3 dispersion on the MI355X.

   Same style name, arguments and pair_coeff grammar as the reference's CUDA style (sevenn/pair_e3gnn/pair_d3.h:12,
   pair_d3.cu:261-285,644-656):
       pair_style d3 <rthr (bohr^2)> <cnthr (bohr^2)> <damp_zero | damp_bj> <functional>
       pair_coeff * * <element of type 1> <element of type 2> ...
   typically combined with the model style:  pair_style hybrid/overlay e3gnn d3 9000 1600 damp_bj pbe
   Like the reference's, it is a one-process style (it reads atom->natoms positions and the periodic box, no neighbor list).
   The arithmetic lives in the library (csrc/snet_d3.hip) behind the reference's own `pair_*` C binding (include/snet_d3_ref.h).

   NOT COMPILED AGAINST LAMMPS in the development image; type-checked against tests/lammps_mock/ only (see pair_e3gnn_hip.h).
------------------------------------------------------------------------- */
#ifdef PAIR_CLASS
// clang-format off
PairStyle(d3, PairD3Hip)
// clang-format on
#else

#ifndef LMP_PAIR_D3_HIP
#define LMP_PAIR_D3_HIP

#include "pair.h"

#include <string>
#include <vector>

struct PairD3;   // handle of the library's D3 engine (snet_d3_ref.h)

namespace LAMMPS_NS {

class PairD3Hip : public Pair {
 public:
  PairD3Hip(class LAMMPS *);
  ~PairD3Hip() override;
  void compute(int, int) override;
  void settings(int, char **) override;
  void coeff(int, char **) override;
  void init_style() override;
  double init_one(int, int) override;

 protected:
  ::PairD3 *d3 = nullptr;
  double rthr = 9000.0, cnthr = 1600.0;
  std::string damping = "damp_bj", functional = "pbe";
  std::vector<int> atomic_numbers;   // per LAMMPS type
  std::vector<double> xflat;
  void allocate();
};

}  // namespace LAMMPS_NS

#endif
#endif
