/* The reference's D3 C-ABI, exported by libsnet_hip.so under the reference's own names and signatures
 * (sevenn/pair_e3gnn/pair_d3_for_ase.cu:2034-2082; bound by ctypes in sevenn/calculator.py:430-483), so the reference's
 * D3Calculator can load this library in place of its CUDA-only libpair_d3.so.
 * Shims over snet_d3_* (snet_hip.h); implementation csrc/snet_d3_ref.cpp.  Units: eV, Angstrom, cut-offs in bohr^2. */
#ifndef SNET_D3_REF_H
#define SNET_D3_REF_H
#ifdef __cplusplus
extern "C" {
#endif
typedef struct PairD3 PairD3;
PairD3 *pair_init(void);                                                           /* pair_d3_for_ase.cu:2035 */
void pair_set_atom(PairD3 *pair, int natoms, int ntypes, int *type, double *x_flat); /* :2039  types 1-based, x[natoms*3] */
void pair_set_domain(PairD3 *pair, int xperiodic, int yperiodic, int zperiodic, double *boxlo, double *boxhi, double xy,
                     double xz, double yz);                                        /* :2048  LAMMPS restricted triclinic box */
void pair_run_settings(PairD3 *pair, double rthr, double cnthr, const char *damp_name, const char *func_name); /* :2052 */
void pair_run_coeff(PairD3 *pair, int *atomic_numbers);                            /* :2056  atomic number of every type */
void pair_run_compute(PairD3 *pair);                                               /* :2060 */
double pair_get_energy(PairD3 *pair);                                              /* :2064  eV */
double *pair_get_force(PairD3 *pair);                                              /* :2068  [natoms*3] eV/A */
double *pair_get_stress(PairD3 *pair);                                             /* :2072  [6] virial sums xx yy zz xy xz yz (eV) */
void pair_fin(PairD3 *pair);                                                       /* :2076 */
/* Failure behaviour.  The reference's functions abort the process on a CUDA error; these never abort: the first failure on a
 * handle (missing data/d3_params.bin, unknown functional / damping name, HIP error) is sticky, printed once to stderr, and from
 * then on pair_get_force / pair_get_stress return NULL and pair_get_energy NaN.  pair_failed (an extension) reads the flag;
 * the message is snet_last_error(). */
int pair_failed(PairD3 *pair);
#ifdef __cplusplus
}
#endif
#endif /* SNET_D3_REF_H */
