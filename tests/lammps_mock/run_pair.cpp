// Driver of the runnable LAMMPS mock (see lmp_mock_core.h): what LAMMPS does around a pair style for ONE `run 0` on one rank --
//   read a structure; pair_style <args>; pair_coeff <args>; init_style; init_one -> neighbor cutoff; periodic ghost images and a
//   FULL neighbor list (cutoff + skin, rotated ilist); compute(eflag, vflag); fold the ghost forces onto their owners (newton on)
// -- with the real pair style classes of lammps/pair_e3gnn_hip.cpp and lammps/pair_d3_hip.cpp linked against libsnet_hip.so.
//   run_pair <e3gnn|e3gnn/parallel|d3> <structure.txt> <out.json> [--steps N] [--empty] [style args ...] -- <pair_coeff args ...>
// structure.txt:  natoms ntypes / 3 cell rows / skin / natoms x (type x y z)     (cell rows a, b, c; LAMMPS' restricted form for d3)
// Output JSON: energy, forces[natoms][3] (tag order), virial[6] (LAMMPS order xx yy zz xy xz yz), eatom, counts.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "pair.h"
#include "../../lammps/pair_d3_hip.h"
#include "../../lammps/pair_e3gnn_hip.h"

using namespace LAMMPS_NS;

namespace {
struct Vec3 { double v[3]; };
}  // namespace

int main(int argc, char **argv) {
  if (argc < 5) {
    std::fprintf(stderr, "usage: run_pair <style> <structure.txt> <out.json> [--steps N] [--empty] [style args] -- <pair_coeff args>\n");
    return 2;
  }
  const std::string style = argv[1];
  int steps = 1;
  bool empty = false;
  std::vector<char *> sargs, cargs;
  bool after = false;
  for (int i = 4; i < argc; ++i) {
    if (!after && std::strcmp(argv[i], "--") == 0) { after = true; continue; }
    if (!after && std::strcmp(argv[i], "--steps") == 0 && i + 1 < argc) { steps = std::atoi(argv[++i]); continue; }
    if (!after && std::strcmp(argv[i], "--empty") == 0) { empty = true; continue; }
    (after ? cargs : sargs).push_back(argv[i]);
  }
  try {
    std::ifstream in(argv[2]);
    if (!in) throw std::runtime_error(std::string("cannot open ") + argv[2]);
    int natoms = 0, ntypes = 0;
    double cell[3][3], skin = 0.0;
    in >> natoms >> ntypes;
    for (auto &row : cell) in >> row[0] >> row[1] >> row[2];
    in >> skin;
    std::vector<int> type0(natoms);
    std::vector<Vec3> pos(natoms);
    for (int i = 0; i < natoms; ++i) in >> type0[i] >> pos[i].v[0] >> pos[i].v[1] >> pos[i].v[2];
    if (!in) throw std::runtime_error("malformed structure file");

    LAMMPS lmp;
    Memory memory; Error error; Atom atom; Comm comm; Force force; Neighbor neighbor; Domain domain;
    lmp.memory = &memory; lmp.error = &error; lmp.atom = &atom; lmp.comm = &comm; lmp.force = &force;
    lmp.neighbor = &neighbor; lmp.domain = &domain;
    atom.natoms = natoms; atom.nlocal = natoms; atom.ntypes = ntypes;
    domain.xperiodic = domain.yperiodic = domain.zperiodic = 1;
    for (int k = 0; k < 3; ++k) { domain.boxlo[k] = 0.0; domain.boxhi[k] = cell[k][k]; }
    domain.xy = cell[1][0]; domain.xz = cell[2][0]; domain.yz = cell[2][1];

    // the owned atoms exist before any pair command runs (read_data); ghosts are added once the neighbor cutoff is known
    memory.create(atom.x, natoms, 3, "atom:x");
    memory.create(atom.f, natoms, 3, "atom:f");
    memory.create(atom.type, natoms, "atom:type");
    memory.create(atom.tag, natoms, "atom:tag");
    for (int i = 0; i < natoms; ++i) {
      for (int d = 0; d < 3; ++d) atom.x[i][d] = pos[i].v[d];
      atom.type[i] = type0[i];
      atom.tag[i] = i + 1;
    }
    std::unique_ptr<Pair> pair;
    if (style == "e3gnn") pair.reset(new PairE3GNNHip(&lmp));
    else if (style == "e3gnn/parallel") pair.reset(new PairE3GNNHipParallel(&lmp));
    else if (style == "d3") pair.reset(new PairD3Hip(&lmp));
    else throw std::runtime_error("unknown pair style " + style);
    pair->settings((int)sargs.size(), sargs.data());
    pair->coeff((int)cargs.size(), cargs.data());
    pair->init_style();
    const double cut = pair->init_one(1, 1) + skin;   // neighbor.cpp: cutneigh = cutforce + skin

    // periodic ghost images within `cut` of any owned atom (what comm->borders() leaves behind), tags = the owners'
    double vol = cell[0][0] * (cell[1][1] * cell[2][2] - cell[1][2] * cell[2][1]) - cell[0][1] * (cell[1][0] * cell[2][2] - cell[1][2] * cell[2][0]) +
                 cell[0][2] * (cell[1][0] * cell[2][1] - cell[1][1] * cell[2][0]);
    vol = std::fabs(vol);
    int reach[3];
    for (int k = 0; k < 3; ++k) {
      const double *a = cell[(k + 1) % 3], *b = cell[(k + 2) % 3];
      const double cx = a[1] * b[2] - a[2] * b[1], cy = a[2] * b[0] - a[0] * b[2], cz = a[0] * b[1] - a[1] * b[0];
      const double height = vol / std::sqrt(cx * cx + cy * cy + cz * cz);
      reach[k] = cut > 0 ? (int)std::ceil(cut / height) : 0;
    }
    std::vector<Vec3> xs(pos);
    std::vector<int> types(type0), tags(natoms), owner;
    for (int i = 0; i < natoms; ++i) tags[i] = i + 1;
    for (int sx = -reach[0]; sx <= reach[0]; ++sx)
      for (int sy = -reach[1]; sy <= reach[1]; ++sy)
        for (int sz = -reach[2]; sz <= reach[2]; ++sz) {
          if (!sx && !sy && !sz) continue;
          for (int j = 0; j < natoms; ++j) {
            Vec3 p;
            for (int d = 0; d < 3; ++d) p.v[d] = pos[j].v[d] + sx * cell[0][d] + sy * cell[1][d] + sz * cell[2][d];
            bool near = false;
            for (int i = 0; i < natoms && !near; ++i) {
              const double dx = p.v[0] - pos[i].v[0], dy = p.v[1] - pos[i].v[1], dz = p.v[2] - pos[i].v[2];
              near = dx * dx + dy * dy + dz * dz < cut * cut;
            }
            if (near) { xs.push_back(p); types.push_back(type0[j]); tags.push_back(j + 1); owner.push_back(j); }
          }
        }
    const int nall = (int)xs.size();
    atom.nghost = nall - natoms;
    memory.destroy(atom.x); memory.destroy(atom.f); memory.destroy(atom.type); memory.destroy(atom.tag);
    memory.create(atom.x, nall, 3, "atom:x");
    memory.create(atom.f, nall, 3, "atom:f");
    memory.create(atom.type, nall, "atom:type");
    memory.create(atom.tag, nall, "atom:tag");
    for (int i = 0; i < nall; ++i) {
      for (int d = 0; d < 3; ++d) atom.x[i][d] = xs[i].v[d];
      atom.type[i] = types[i];
      atom.tag[i] = tags[i];
    }
    comm.first_ghost = natoms;
    comm.ghost_owner = owner;

    // FULL neighbor list of the owned atoms, ilist rotated (a pair style must not assume ilist[ii] == ii)
    NeighList list;
    std::vector<int> ilist(natoms), numneigh(nall, 0);
    std::vector<std::vector<int>> rows(nall);
    std::vector<int *> first(nall, nullptr);
    for (int ii = 0; ii < natoms; ++ii) ilist[ii] = (ii + 7) % natoms;
    for (int i = 0; i < natoms; ++i) {
      for (int j = 0; j < nall; ++j) {
        if (j == i) continue;
        const double dx = xs[j].v[0] - xs[i].v[0], dy = xs[j].v[1] - xs[i].v[1], dz = xs[j].v[2] - xs[i].v[2];
        if (dx * dx + dy * dy + dz * dz < cut * cut) rows[i].push_back(j);
      }
      numneigh[i] = (int)rows[i].size();
      first[i] = rows[i].data();
    }
    list.inum = empty ? 0 : natoms;
    list.ilist = ilist.data(); list.numneigh = numneigh.data(); list.firstneigh = first.data();
    pair->list = &list;   // (neighbor->init_pair hands the style its list)

    double energy = 0.0, vir[6] = {0, 0, 0, 0, 0, 0};
    std::vector<double> eatom_out(natoms, 0.0);
    for (int s = 0; s < steps; ++s) {
      neighbor.ago = s;   // step 0: the list was just rebuilt; later steps reuse it
      for (int i = 0; i < nall; ++i) atom.f[i][0] = atom.f[i][1] = atom.f[i][2] = 0.0;
      const int eflag = style == "e3gnn" ? 3 : 1, vflag = 2;   // global energy + virial; per-atom energy where the style has it
      pair->compute(eflag, vflag);
      for (int k = 0; k < atom.nghost; ++k)   // newton on: comm->reverse_comm() adds every ghost's force to its owner
        for (int d = 0; d < 3; ++d) atom.f[owner[k]][d] += atom.f[natoms + k][d];
      energy = pair->eng_vdwl;
      std::memcpy(vir, pair->virial, sizeof vir);
      if ((eflag & 2) && pair->eatom) for (int i = 0; i < natoms; ++i) eatom_out[i] = pair->eatom[i];
    }
    FILE *o = std::fopen(argv[3], "w");
    if (!o) throw std::runtime_error(std::string("cannot write ") + argv[3]);
    std::fprintf(o, "{\"energy\": %.17g, \"volume\": %.17g, \"nlocal\": %d, \"nghost\": %d, \"neigh_flags\": %d, \"cutneigh\": %.17g,\n \"virial\": [",
                 energy, vol, natoms, atom.nghost, neighbor.requested_flags, cut);
    for (int k = 0; k < 6; ++k) std::fprintf(o, "%s%.17g", k ? ", " : "", vir[k]);
    std::fprintf(o, "],\n \"eatom\": [");
    for (int i = 0; i < natoms; ++i) std::fprintf(o, "%s%.17g", i ? ", " : "", eatom_out[i]);
    std::fprintf(o, "],\n \"forces\": [");
    for (int i = 0; i < natoms; ++i) std::fprintf(o, "%s[%.17g, %.17g, %.17g]", i ? ", " : "", atom.f[i][0], atom.f[i][1], atom.f[i][2]);
    std::fprintf(o, "]}\n");
    std::fclose(o);
    pair.reset();
  } catch (const std::exception &e) {
    std::fprintf(stderr, "%s\n", e.what());
    return 3;
  }
  return 0;
}
