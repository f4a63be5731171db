// Minimal MOCK of the LAMMPS API subset used by lammps/pair_e3gnn_hip.{h,cpp} and lammps/pair_d3_hip.{h,cpp}.
// NOT LAMMPS: written from the public LAMMPS developer documentation for an image that has no LAMMPS tree and no MPI.
// Two uses: (1) tests/test_lammps_glue_cpu.py type-checks the glue against these declarations (`g++ -fsyntax-only`);
// (2) round 5: lmp_mock_runtime.cpp gives every declaration a single-rank BEHAVIOUR (error->all throws, memory->create
// allocates LAMMPS-style contiguous 2-d arrays, comm->forward_comm copies owner -> periodic ghost through the pair's own
// pack / unpack, MPI_Bcast / Alltoall(v) are self-copies) and run_pair.cpp drives `pair_coeff -> init_style -> compute` on a real
// structure, so PairE3GNNHip::coeff / build_halo_plan / compute EXECUTE against libsnet_hip.so (tests/test_lammps_glue_gpu.py).
// It still proves nothing about LAMMPS' own internals (neighbor build, comm_brick swaps, fix / run loops).
#pragma once
#include <cstdint>
#include <cstdlib>
#include <sstream>
#include <string>
#include <utility>
#include <vector>

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
  // LAMMPS layout: one contiguous data block + a row-pointer vector (memory.h: create(TYPE **&, int, int, const char *))
  template <typename T> T **create(T **&array, int n1, int n2, const char * /*name*/) {
    T *data = static_cast<T *>(std::calloc((size_t)n1 * n2 + 1, sizeof(T)));
    array = static_cast<T **>(std::malloc(sizeof(T *) * (size_t)(n1 > 0 ? n1 : 1)));
    for (int i = 0; i < n1; ++i) array[i] = data + (size_t)i * n2;
    if (n1 == 0) array[0] = data;
    return array;
  }
  template <typename T> T *create(T *&array, int n, const char * /*name*/) {
    array = static_cast<T *>(std::calloc((size_t)n + 1, sizeof(T)));
    return array;
  }
  template <typename T> void destroy(T **&array) {
    if (array) { std::free(array[0]); std::free(array); }
    array = nullptr;
  }
  template <typename T> void destroy(T *&array) {
    std::free(array);
    array = nullptr;
  }
};
class Atom {
 public:
  long natoms = 0;
  int nlocal = 0, nghost = 0, ntypes = 0;
  double **x = nullptr, **f = nullptr;
  int *type = nullptr;
  tagint *tag = nullptr;
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
  int me = 0, nprocs = 1;
  virtual ~Comm() = default;
  virtual void forward_comm(Pair *);   // owner -> ghost through pair->pack_forward_comm / unpack_forward_comm
  virtual void reverse_comm(Pair *);   // ghost -> owner through pair->pack_reverse_comm / unpack_reverse_comm
  // (mock state, filled by the driver: single rank, so every ghost is the periodic image of an owned atom)
  int first_ghost = 0;
  std::vector<int> ghost_owner;        // ghost k = atom first_ghost + k is an image of atom ghost_owner[k]
};
class Force {
 public:
  int newton_pair = 1;
};
class NeighList {
 public:
  int inum = 0;
  int *ilist = nullptr, *numneigh = nullptr;
  int **firstneigh = nullptr;
};
namespace NeighConst {
enum { REQ_DEFAULT = 0, REQ_FULL = 1 << 0 };
}
class NeighRequest;
class Neighbor {
 public:
  int ago = 0;
  NeighRequest *add_request(Pair *, int flags = 0);
  int requested_flags = -1;            // (mock state) what the pair style asked for
};
namespace utils {
void mock_log(const std::string &text);
inline void mock_fmt(std::string &out, const std::string &f, size_t pos) { out += f.substr(pos); }
template <typename T, typename... Rest> void mock_fmt(std::string &out, const std::string &f, size_t pos, T &&v, Rest &&...rest) {
  const size_t k = f.find("{}", pos);   // the subset of {fmt} the glue uses: positional "{}" only
  if (k == std::string::npos) { out += f.substr(pos); return; }
  std::ostringstream o;
  o << v;
  out += f.substr(pos, k - pos) + o.str();
  mock_fmt(out, f, k + 2, std::forward<Rest>(rest)...);
}
template <typename... Args> void logmesg(LAMMPS * /*lmp*/, const std::string &format, Args &&...args) {
  std::string out;
  mock_fmt(out, format, 0, std::forward<Args>(args)...);
  mock_log(out);
}
}

// the owner of the singletons a Pointers-derived class reaches through its reference members (lammps.h)
class LAMMPS {
 public:
  Memory *memory = nullptr;
  Error *error = nullptr;
  Atom *atom = nullptr;
  Comm *comm = nullptr;
  Force *force = nullptr;
  Neighbor *neighbor = nullptr;
  Domain *domain = nullptr;
  MPI_Comm world = 0;
};

class Pointers {
 public:
  explicit Pointers(LAMMPS *ptr)
      : lmp(ptr), memory(ptr->memory), error(ptr->error), atom(ptr->atom), comm(ptr->comm), force(ptr->force),
        neighbor(ptr->neighbor), domain(ptr->domain), world(ptr->world) {}
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
