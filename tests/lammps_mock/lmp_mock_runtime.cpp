// Single-rank BEHAVIOUR for the mock LAMMPS classes of lmp_mock_core.h / pair.h / mpi.h (see the header comment there).
// Test scaffolding: no LAMMPS code, written from the public developer documentation of the calls the glue makes.
#include <cstdio>
#include <cstring>
#include <stdexcept>

#include "pair.h"

namespace LAMMPS_NS {

// error->all / error->one end the run in LAMMPS; here they throw so the driver can report the message and exit non-zero
void Error::all(const std::string &file, int line, const std::string &msg) {
  throw std::runtime_error("ERROR: " + msg + " (" + file + ":" + std::to_string(line) + ")");
}
void Error::one(const std::string &file, int line, const std::string &msg) {
  throw std::runtime_error("ERROR on proc 0: " + msg + " (" + file + ":" + std::to_string(line) + ")");
}

int Atom::tag_consecutive() {
  std::vector<char> seen((size_t)natoms + 1, 0);
  for (int i = 0; i < nlocal; ++i) {
    if (tag[i] < 1 || tag[i] > natoms || seen[tag[i]]) return 0;
    seen[tag[i]] = 1;
  }
  return 1;
}

// comm.cpp Comm::forward_comm(Pair *): per swap  n = pair->pack_forward_comm(sendnum, sendlist, buf, pbc_flag, pbc);
// (send / self-copy)  pair->unpack_forward_comm(recvnum, firstrecv, buf).  One rank: one self-copy swap whose send list is the
// owner of every ghost image.
void Comm::forward_comm(Pair *pair) {
  const int n = (int)ghost_owner.size();
  if (n == 0) return;
  std::vector<double> buf((size_t)n * (pair->comm_forward > 0 ? pair->comm_forward : 1));
  int pbc[6] = {0, 0, 0, 0, 0, 0};
  pair->pack_forward_comm(n, ghost_owner.data(), buf.data(), 1, pbc);
  pair->unpack_forward_comm(n, first_ghost, buf.data());
}
void Comm::reverse_comm(Pair *pair) {
  const int n = (int)ghost_owner.size();
  if (n == 0) return;
  std::vector<double> buf((size_t)n * (pair->comm_reverse > 0 ? pair->comm_reverse : 1));
  pair->pack_reverse_comm(n, first_ghost, buf.data());
  pair->unpack_reverse_comm(n, ghost_owner.data(), buf.data());
}

NeighRequest *Neighbor::add_request(Pair *, int flags) {
  requested_flags = flags;
  return nullptr;
}

namespace utils {
void mock_log(const std::string &text) { std::fputs(text.c_str(), stderr); }
}  // namespace utils

Pair::Pair(LAMMPS *ptr) : Pointers(ptr) {}
Pair::~Pair() {
  std::free(eatom);
  if (vatom) { std::free(vatom[0]); std::free(vatom); }
}
void Pair::init_style() {}

// pair.cpp Pair::ev_setup restated for the flags the glue reads: zero the accumulators, size the per-atom arrays
void Pair::ev_setup(int eflag, int vflag, int /*alloc*/) {
  evflag = 1;
  eflag_either = eflag != 0;
  eflag_global = eflag & 1;
  eflag_atom = (eflag & 2) != 0;
  vflag_either = vflag != 0;
  vflag_global = (vflag & 3) != 0;
  vflag_atom = (vflag & 4) != 0;
  eng_vdwl = eng_coul = 0.0;
  for (double &v : virial) v = 0.0;
  const int nall = atom->nlocal + atom->nghost;
  if (eflag_atom) {
    if (maxeatom < nall) { std::free(eatom); eatom = static_cast<double *>(std::calloc((size_t)nall + 1, sizeof(double))); maxeatom = nall; }
    std::memset(eatom, 0, sizeof(double) * (size_t)nall);
  }
  if (vflag_atom) {
    if (maxvatom < nall) {
      if (vatom) { std::free(vatom[0]); std::free(vatom); }
      memory->create(vatom, nall, 6, "pair:vatom");
      maxvatom = nall;
    }
    std::memset(vatom[0], 0, sizeof(double) * 6 * (size_t)nall);
  }
}

}  // namespace LAMMPS_NS

// ---- MPI over one rank: every collective is a self-copy ---------------------------------------------------------------
static size_t mock_type_size(MPI_Datatype t) { return t == MPI_INT ? sizeof(int) : 1; }
int MPI_Bcast(void *, int, MPI_Datatype, int, MPI_Comm) { return 0; }
int MPI_Alltoall(const void *sendbuf, int sendcount, MPI_Datatype sendtype, void *recvbuf, int, MPI_Datatype, MPI_Comm) {
  std::memcpy(recvbuf, sendbuf, (size_t)sendcount * mock_type_size(sendtype));
  return 0;
}
int MPI_Alltoallv(const void *sendbuf, const int *sendcounts, const int *sdispls, MPI_Datatype sendtype, void *recvbuf,
                  const int *, const int *rdispls, MPI_Datatype, MPI_Comm) {
  const size_t sz = mock_type_size(sendtype);
  std::memcpy(static_cast<char *>(recvbuf) + (size_t)rdispls[0] * sz, static_cast<const char *>(sendbuf) + (size_t)sdispls[0] * sz,
              (size_t)sendcounts[0] * sz);
  return 0;
}
