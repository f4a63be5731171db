// Minimal MOCK of the LAMMPS API subset used by lammps/pair_e3gnn_hip.{h,cpp}.
// NOT LAMMPS: declarations only, written from the public LAMMPS developer documentation, so that the glue
// can meet a C++ type checker in an image that has no LAMMPS tree (tests/test_lammps_glue_cpu.py runs
// `g++ -fsyntax-only`).  It proves the file parses and the calls are type-consistent with these
// declarations -- nothing about behaviour inside LAMMPS.
#pragma once
#include <cstdint>
#include <string>

#include "mpi.h"

#define FLERR __FILE__, __LINE__

namespace LAMMPS_NS {
typedef int tagint;  // LAMMPS_SMALLBIG default

class LAMMPS;
class Pair;

class Error {
 public:
  [[noreturn]] void all(const std::string &file, int line, const std::string &msg);
  [[noreturn]] void one(const std::string &file, int line, const std::string &msg);
};
class Memory {
 public:
  template <typename T> T **create(T **&array, int n1, int n2, const char *name);
  template <typename T> T *create(T *&array, int n, const char *name);
  template <typename T> void destroy(T **&array);
  template <typename T> void destroy(T *&array);
};
class Atom {
 public:
  long natoms;
  int nlocal, nghost, ntypes;
  double **x, **f;
  int *type;
  tagint *tag;
  int tag_consecutive();
};
class Domain {
 public:
  int xperiodic, yperiodic, zperiodic;
  double boxlo[3], boxhi[3];
  double xy, xz, yz;
};
class Comm {
 public:
  int me, nprocs;
  virtual void forward_comm(Pair *);
  virtual void reverse_comm(Pair *);
};
class Force {
 public:
  int newton_pair;
};
class NeighList {
 public:
  int inum;
  int *ilist, *numneigh;
  int **firstneigh;
};
namespace NeighConst {
enum { REQ_DEFAULT = 0, REQ_FULL = 1 << 0 };
}
class NeighRequest;
class Neighbor {
 public:
  int ago;
  NeighRequest *add_request(Pair *, int flags = 0);
};
namespace utils {
template <typename... Args> void logmesg(LAMMPS *lmp, const std::string &format, Args &&...args);
}

class Pointers {
 public:
  explicit Pointers(LAMMPS *);
  virtual ~Pointers() = default;

 protected:
  LAMMPS *lmp;
  Memory *&memory;
  Error *&error;
  Atom *&atom;
  Comm *&comm;
  Force *&force;
  Neighbor *&neighbor;
  Domain *&domain;
  MPI_Comm &world;
};
}  // namespace LAMMPS_NS
