/*
 * snet_hip.h -- C ABI of libsnet_hip.so, the MI355X (gfx950) force-engine kernels
 * for SevenNet's per-MD-step energy/force path.
 *
 * Boundary rules
 *   - extern "C", plain pointers and sizes only (no torch / ATen types);
 *   - every pointer argument is a DEVICE pointer unless the name ends in _host;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*;
 *     NULL = the legacy default stream); the caller owns all buffers;
 *   - return value: 0 = ok, nonzero = error, message via snet_last_error()
 *     (thread-local).  Style follows the reference's only existing C ABI,
 *     sevenn/pair_e3gnn/pair_d3_for_ase.cu:2034-2082 (opaque handle + int rc).
 *   - all floating point is fp32 (the reference is fp32-only:
 *     sevenn/main/sevenn.py:138-139), indices are int32.
 *
 * Feature layout: node features are `ir_mul` (for each irrep block, a
 * [2l+1][mul] slab; blocks concatenated) inside the engine.  The reference's
 * boundary layout `mul_ir` ([mul][2l+1]) is converted with snet_permute_cols().
 *
 * Each entry point cites the reference code it replaces.
 */
#ifndef SNET_HIP_H
#define SNET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever an exported signature or a file format changes incompatibly; every host checks snet_abi_version() against
 * the header it was built from (sevennet_amd/_lib.py, lammps/pair_*_hip.cpp).  History: 1 = rounds 1-3; 2 = round 4 (snet_nl_grid /
 * _bin / _count / _fill gained the open-axis arguments, .snet files moved to "SNETMDL4") and round 5 (pair_failed); 3 = round 6 (the
 * unwired two-fp16-term grouped GEMM snet_gemm_f16_size / _pack / snet_gemm_grouped_f16 left the library). */
#define SNET_ABI_VERSION 3

/* ---- housekeeping ------------------------------------------------------- */
int snet_abi_version(void);
const char *snet_last_error(void);
/* number of compiled tensor-product shapes; tag i (12 hex chars) */
int snet_conv_num_shapes(void);
const char *snet_conv_shape_tag(int i);

/* ---- a1: edge embedding -------------------------------------------------
 * replaces EdgeEmbedding.forward, sevenn/nn/edge_embedding.py:207-217
 * (BesselBasis :101-103, PolynomialCutoff :125-132, XPLORCutoff :150-160,
 *  SphericalEncoding :164-185 = e3nn SphericalHarmonics 'component').       */
typedef struct snet_edge_params {
  float cutoff;        /* rc */
  int32_t n_basis;     /* <= 16 */
  int32_t cutoff_kind; /* 0 poly_cut, 1 XPLOR */
  int32_t poly_p;      /* poly_cut p value */
  float cutoff_on;     /* XPLOR r_on */
  int32_t lmax;        /* <= 3 */
  int32_t normalize;   /* SH on unit vector (1) or raw vector (0, <0.10 checkpoints) */
} snet_edge_params;

/* Per-step input of an MD host: positions, not edge vectors.  edge_vec[e] = pos[src[e]] - pos[center[e]] + shift[e]
 * (fp64 positions [n_total,3] and per-edge periodic-image offsets shift[E,3] = S.cell, nullable = no images; the
 * subtraction runs in fp64, as in the reference's hosts: pair_e3gnn.cpp:160-175, train/dataload.py:62-70).  With the
 * topology resident, a step uploads 24 bytes per atom instead of 12 bytes per edge. */
int snet_edge_vectors(const double *pos, const int32_t *center, const int32_t *src, const double *shift, int64_t n_edges,
                      float *edge_vec, void *stream);
/* edge_vec[E,3] -> emb[E,n_basis], sh[E,nsh], nsh = (lmax+1)^2; coeffs_host[n_basis] = Bessel c_n
 * (HOST).  dsh (nullable) receives the Jacobian d sh[e,i] / d edge_vec[e,a] as [E,nsh,3]; the
 * tensor-product reverse kernel contracts with it so only 3 values per edge are reduced. */
int snet_edge_embed_fwd(const snet_edge_params *p_host, const float *coeffs_host, const float *edge_vec,
                        int64_t n_edges, float *emb, float *sh, float *dsh, void *stream);
/* g_vec[E,3] (+)= d/d edge_vec of <g_emb,emb> + <g_sh,sh>  (replaces autograd through a1).
 * g_sh may be NULL (spherical part already accumulated by snet_conv_bwd_edge_vec);
 * accumulate != 0 adds to g_vec instead of overwriting it. */
int snet_edge_embed_bwd(const snet_edge_params *p_host, const float *coeffs_host, const float *edge_vec,
                        int64_t n_edges, const float *g_emb, const float *g_sh, float *g_vec, int32_t accumulate,
                        void *stream);

/* ---- a3/a4/a2.1: dense channel mixing on MFMA (fp32 in / fp32 acc) --------
 * One call = one per-irrep block of e3nn o3.Linear (sevenn/nn/linear.py:94-100),
 * one species slice of the FCTP self-connection (self_connection.py:11-67), or
 * one layer of the radial FullyConnectedNet (convolution.py:93-95,121).
 *   C[node(n), m, :] (+)= A[node(n), m, :] @ B          n < n_nodes, m < d
 *   A row address = A + node*a_node_stride + a_off + m*K      (K contiguous)
 *   C row address = C + node*c_node_stride + c_off + m*N
 *   node(n) = row_idx ? row_idx[n] : n        (species-grouped rows for FCTP)
 * B is [K,N] row-major with the e3nn path normalisation already folded in.   */
int snet_gemm(const float *A, const float *B, float *C, int64_t n_nodes, int32_t d, int32_t K, int32_t N,
              int64_t a_node_stride, int64_t a_off, int64_t c_node_stride, int64_t c_off,
              const int32_t *row_idx, int32_t accumulate, void *stream);

/* All per-irrep GEMMs of ONE equivariant linear in a single launch (same A, C, node strides, row
 * list; blocks writing the same output block must not be grouped unless the later ones accumulate
 * into distinct rows -- the host keeps accumulating blocks in separate launches).             */
typedef struct snet_gemm_desc {
  const float *B;      /* device [K,N] row-major (exact fp32 MFMA), or NULL when B_split is given */
  const void *B_split; /* device copy of a snet_gemm_split_pack buffer: bf16 x 6 split-precision MFMA
                          (fp32-rounding-class error, ~2.5x the fp32 matrix-pipe rate); or NULL */
  int64_t a_off, c_off;
  int32_t d, K, N, accumulate;
} snet_gemm_desc;
#define SNET_MAX_GEMM_GROUP 8
int snet_gemm_grouped(const snet_gemm_desc *descs_host, int32_t n_desc, const float *A, float *C, int64_t n_nodes,
                      int64_t a_node_stride, int64_t c_node_stride, const int32_t *row_idx, void *stream);
/* weights [K,N] (HOST, fp32, normalisation folded) -> matrix-core B fragments, each value written as
 * a sum of three bf16 terms (HOST buffer of snet_gemm_split_size bytes; upload it and pass the device
 * copy as snet_gemm_desc.B_split).  All problems of one grouped launch must use the same kind. */
int64_t snet_gemm_split_size(int32_t K, int32_t N);
int snet_gemm_split_pack(const float *B_host, int32_t K, int32_t N, void *packed_host);
/* Fused radial MLP, e3nn FullyConnectedNet([nb,h1,h2,wn], act) (convolution.py:93-95,121):
 *   fwd  w[E,wn] = (act(act(emb W0) cst W1) cst) W2          W0[nb,h1] W1[h1,h2] W2[h2,wn]
 *   bwd  g_emb[E,nb] += d<g_w,w>/d emb
 * A plan holds device copies of the weights (HOST pointers in, row-major, 1/sqrt(fan_in) already
 * folded).  Hidden activations stay in registers (fwd) or are recomputed (bwd).
 *   mode 0: exact fp32 MFMA (v_mfma_f32_32x32x2_f32, bitwise an fmaf chain)
 *   mode 1: bf16 x 6 split products on v_mfma_f32_32x32x16_bf16 (operands written as 3-term bf16
 *           sums, the six products of order <= 2 accumulated in fp32: fp32-rounding-class error at
 *           6/16 of the fp32 matrix-pipe time)
 * Only h1 = h2 = 64 and nb <= 32 are fused (rc 2 otherwise: use snet_gemm + snet_act_*). */
typedef struct snet_mlp_plan snet_mlp_plan;
int snet_radial_mlp_plan_create(int32_t nb, int32_t h1, int32_t h2, int32_t wn, const float *W0_host,
                                const float *W1_host, const float *W2_host, int32_t act, float cst, int32_t mode,
                                snet_mlp_plan **plan);
void snet_radial_mlp_plan_destroy(snet_mlp_plan *plan);
int snet_radial_mlp_fwd(const snet_mlp_plan *plan, const float *emb, int64_t n_edges, float *w_out, void *stream);
int snet_radial_mlp_bwd(const snet_mlp_plan *plan, const float *emb, const float *g_w, int64_t n_edges,
                        float *g_emb, void *stream);

/* a = act(z)*cst  /  g_z = g_a * cst * act'(z)   (act: 0 silu, 1 tanh);
 * radial-MLP hidden activations, normalize2mom constant `cst` */
int snet_act_fwd(const float *z, float *a, int64_t n, int32_t act, float cst, void *stream);
int snet_act_bwd(const float *z, const float *g_a, float *g_z, int64_t n, int32_t act, float cst, void *stream);

/* ---- a2.2-a2.5: fused gather -> uvu tensor product -> segmented reduce ----
 * replaces IrrepsConvolution.forward lines 131-135 / the `convolution_cls`
 * plug-in call of IrrepsScatterGatterFusedConvolution (convolution.py:270-278):
 *   out[i] = scale * sum_{e: dst(e)=i} TP_uvu(x[src(e)], sh[e]; w[e])
 * Edges must be sorted by destination (CSR row_ptr over the n_dst owned
 * nodes); x has n_src >= n_dst rows (ghost rows are gather sources only,
 * convolution.py:125-138).  A plan is a compiled specialisation looked up by
 * its shape tag (sevennet_amd.model_spec.ConvSpec.tag).                      */
typedef struct snet_conv_plan snet_conv_plan;
int snet_conv_plan_create(const char *tag, snet_conv_plan **plan);
void snet_conv_plan_destroy(snet_conv_plan *plan);
/* Shapes are compiled ahead of time (sevennet_amd/shapes.py) or ON DEMAND: a shape library is a small .so holding
 * the generated kernels of one or more shapes (python -m sevennet_amd.jit <irreps...>, sevennet_amd.jit.ensure_conv_shape,
 * which the Python plug-in and engine call themselves for an unknown shape -- the reference's accelerator back ends JIT
 * theirs the same way, convolution.py:237-247).  Loading it registers its shapes with this library. */
int snet_conv_register_library(const char *path);
int snet_conv_plan_dims(const snet_conv_plan *plan, int32_t *dx, int32_t *dout, int32_t *nsh, int32_t *wn);

/* w_row (nullable, int32[E]): row of w that edge e reads.  NULL = edge e reads row e (the reference's
 * layout, convolution.py:121).  The radial weights depend on |r| only, so the two directed edges of
 * one undirected pair carry the same w row; snet_edge_pairs builds the map that lets the radial MLP
 * run once per pair and w hold one row per pair.                                                   */
int snet_conv_fwd(const snet_conv_plan *plan, const float *x, const float *sh, const float *w, const int32_t *w_row,
                  const int32_t *row_ptr, const int32_t *src, int64_t n_dst, float scale, float *out,
                  void *stream);
/* ---- fused radial weights + tensor product: w[E,wn] and g_w[E,wn] never exist in memory ----------
 * The contract of the reference's own accelerator plug-ins (flash_helper.py:43, convolution.py:118-141:
 * no `weight` / `message` tensor survives) taken one step further: the radial MLP's LAST layer
 * w_e = h2_e @ W2 (64 -> weight_numel, the largest FLOP term of the model) runs on the matrix cores
 * inside the tensor-product kernels (v_mfma_f32_16x16x32_bf16 tiles whose accumulator layout is the
 * tensor product's operand layout), and the reverse kernel contracts the weight gradient with W2^T on
 * the fly.  What crosses HBM per edge is h2[64] in and g_h2[64] out instead of w[wn] and g_w[wn].
 *   h2[R,64]   hidden activations a2 of the MLP (snet_radial_mlp_hidden_fwd), R = edges or undirected
 *              pairs; edge e reads row w_row[e] (NULL: row e)
 *   terms      precision mode of the in-kernel products (both operands split into low-precision terms, fp32 accumulate):
 *              4 = f16x3 (default of both hosts): two fp16 terms per operand, hi*hi + hi*lo + lo*hi on
 *              v_mfma_f32_16x16x32_f16, operands scaled by powers of two (W2 per matrix at plan creation, h2 / g_w per
 *              16-edge tile in the kernel) -- 22 significand bits per operand: fp32-rounding class;
 *              3 = bf16x6 (three bf16 terms, six products: fp32-rounding class at twice the matrix-core work);
 *              2 = bf16x3 (two bf16 terms: ~2^-17 per product; whole-model force error at max|F| = 8 eV/A up to
 *              3.5e-4 eV/A, i.e. outside the 1e-4 bar -- kept for throughput comparisons);
 *              1 = plain bf16 (1e-2 relative)
 *   tile_ptr   int32[n_dst+1], exclusive scan of ceil(degree/16), and tile_node int32[n_tiles] (tile -> node), both
 *              from snet_edge_tiles: the reverse kernel gives each 16-edge tile of a node's CSR segment to one
 *              wavefront (tile_node capacity: n_dst + n_edges / 16 entries always suffice)
 * reverse outputs: g_xe[E,dx] (nullable; chunk order of snet_fused_plan_gxe_chunks; sum per source with
 * snet_segment_sum_rows_chunked), g_vec[E,3] ACCUMULATED (as
 * snet_conv_bwd_edge_vec), and exactly ONE of
 *   g_h2[E,64]  overwritten; feed to snet_radial_mlp_hidden_bwd, or
 *   g_emb[E,nb] ACCUMULATED: the kernel also reverses the MLP's two hidden layers per 16-edge tile (it reads
 *               emb[E,nb], the radial basis values per DIRECTED edge), so g_h2 never reaches memory either.
 *               Needs snet_fused_plan_has_mlp_tail(plan) != 0 (n_basis <= 16 and a multiple of 4).
 * x_rowmax[n_rows of x] / g_rowmax[n_dst]: upper bounds of max|x[row]| and max|g_out[node]| (snet_row_absmax, or
 *   snet_row_norm2 of the linear map's input that produced g_out); required for terms = 4, ignored (NULL) otherwise.
 * snet_conv_fused_available() != 0 iff the shape has these kernels (channel multiplicities % 16 == 0). */
#define SNET_FUSED_TERMS_DEFAULT 4
typedef struct snet_fused_plan snet_fused_plan;
int snet_radial_mlp_hidden_fwd(const snet_mlp_plan *plan, const float *emb, int64_t n_edges, float *h2, void *stream);
/* the hidden activations of n_layers (1 .. 8) interaction layers in ONE launch: the layers read the same edge embedding
 * (nn/edge_embedding.py:176-181 feeds every convolution's weight_nn, nn/convolution.py:124) and differ in weights only;
 * h2[l] <- what snet_radial_mlp_hidden_fwd(plans[l], ...) writes, bit for bit (it is the same kernel with one layer). */
int snet_radial_mlp_hidden_fwd_layers(const snet_mlp_plan *const *plans, int32_t n_layers, const float *emb, int64_t n_edges,
                                      float *const *h2, void *stream);
int snet_radial_mlp_hidden_bwd(const snet_mlp_plan *plan, const float *emb, const float *g_h2, int64_t n_edges,
                               float *g_emb, void *stream);
int snet_conv_fused_available(const snet_conv_plan *plan);
int snet_fused_plan_create(const snet_conv_plan *plan, const snet_mlp_plan *mlp, int32_t terms, snet_fused_plan **out);
void snet_fused_plan_destroy(snet_fused_plan *plan);
int snet_edge_tiles(const int32_t *row_ptr, int64_t n_dst, int32_t *tile_ptr, int32_t *tile_node, int64_t tile_capacity,
                    int64_t *n_tiles, void *stream);
/* Packed work list (reverse kernels whose snet_fused_plan_tile_mode() is 1 -- every shape whose registers hold a second
 * row's g_out entries; the lmax-3 shapes keep the per-row tiles of snet_edge_tiles, mode 0): a tile is a window of <= 16 CONSECUTIVE CSR edges of
 * the destination rows [node_begin, node_end) that touches at most two rows.  tile_e0[n_tiles + 1] (first edge of every
 * tile, then the end of the last) goes where the reverse kernel takes tile_ptr, tile_nodes[2 n_tiles] (row of the tile's
 * first / last edge) where it takes tile_node.  (node_end - node_begin) + n_edges / 16 tiles always suffice. */
int snet_edge_tiles_packed(const int32_t *row_ptr, int64_t node_begin, int64_t node_end, int32_t *tile_e0,
                           int32_t *tile_nodes, int64_t tile_capacity, int64_t *n_tiles, void *stream);
int snet_fused_plan_tile_mode(const snet_fused_plan *plan);
/* Scalar-output shapes (every path (l, l -> 0): the last interaction layer): the source-row gradient
 *   g_x[j] = sum over the edges e that have j as their source of  w_e * T * Y(e) * g_out[center(e)]
 * is itself a uvu convolution -- shape `tag` (x = the g_out row, out = g_x), radial weights W2 with column c scaled by
 * col_scale[c], run with snet_conv_fwd_fused over the edges grouped by SOURCE (row_ptr = col_ptr, src = center_t, w_row =
 * w_row_t, sh rows gathered by eperm; n_dst = n_total).  It gathers dout floats per edge where the per-edge path
 * (g_xe of snet_conv_bwd_fused + snet_segment_sum_rows_chunked) writes and re-reads dx floats.  Replaces nothing in the
 * reference (e3nn's autograd scatters g_x rows with index_add, sevenn/nn/convolution.py:130-136); exists because the
 * reverse kernel's g_xe stores are half of the last layer's time.
 * snet_conv_plan_transposed: tag[13] ("" when the shape has a non-scalar output), col_scale[wn] (nullable), dead[2 k] =
 * (offset, length) ranges of g_x the transposed product leaves unwritten (zero them).
 * snet_edges_by_source: center_t[e'] / w_row_t[e'] of the edge eperm[e'] (w_row NULL: w_row_t = eperm). */
int snet_conv_plan_transposed(const snet_conv_plan *plan, char *tag, float *col_scale, int32_t *dead, int32_t dead_capacity,
                              int32_t *n_dead);
int snet_edges_by_source(const int32_t *row_ptr, int64_t n_dst, const int32_t *eperm, const int32_t *w_row, int64_t n_edges,
                         int32_t *center_t, int32_t *w_row_t, void *stream);
int snet_conv_fwd_fused(const snet_fused_plan *plan, const float *x, const float *sh, const float *h2,
                        const int32_t *w_row, const int32_t *row_ptr, const int32_t *src, int64_t n_dst, float scale,
                        float *out, void *stream);
int snet_conv_bwd_fused(const snet_fused_plan *plan, const float *x, const float *sh, const float *dsh,
                        const float *h2, const int32_t *w_row, const int32_t *row_ptr, const int32_t *src,
                        const int32_t *tile_ptr, const int32_t *tile_node, int64_t n_tiles, float scale,
                        const float *g_out, float *g_xe, float *g_h2, const float *emb, float *g_emb, float *g_vec,
                        const float *x_rowmax, const float *g_rowmax, void *stream);
/* The same reverse pass for hosts that differentiate with respect to the harmonics THEMSELVES -- the reference's plug-in point
 * hands `edge_attr` to autograd (sevenn/nn/convolution.py:118-141, flash_helper.py:33-48): g_sh[E, nsh] is OVERWRITTEN with
 * dE/dY of every edge; no dsh / g_vec (the caller chains through its own spherical-harmonics module). */
int snet_conv_bwd_fused_sh(const snet_fused_plan *plan, const float *x, const float *sh, const float *h2, const int32_t *w_row,
                           const int32_t *row_ptr, const int32_t *src, const int32_t *tile_ptr, const int32_t *tile_node,
                           int64_t n_tiles, float scale, const float *g_out, float *g_xe, float *g_h2, const float *emb,
                           float *g_emb, float *g_sh, const float *x_rowmax, const float *g_rowmax, void *stream);
int snet_fused_plan_has_mlp_tail(const snet_fused_plan *plan);
/* g_xe[E,dx] of snet_conv_bwd_fused is an intermediate with its own row layout: the 16-channel chunks of a row are stored in
 * the order the kernel produces them ([x block][channel tile][component]: each (block, tile) writes one contiguous run per
 * edge instead of half cache lines).  chunk_pos[s] (dx / 16 entries, host) = position of standard chunk s in the row;
 * snet_segment_sum_rows_chunked (chunk_pos on the DEVICE) sums such rows per source atom and returns standard order. */
int snet_fused_plan_gxe_chunks(const snet_fused_plan *plan, int32_t *chunk_pos, int32_t capacity);
int snet_segment_sum_rows_chunked(const float *x, const int32_t *seg_ptr, const int32_t *perm, int64_t n_seg, int32_t dim,
                                  const int32_t *chunk_pos, float *out, void *stream);
/* out[r] = max_k |x[r,k]| for r < n_rows.  The f16x3 reverse kernel (terms = 4) scales its fp16 operand g_w per edge by a
 * power of two derived from a BOUND of |g_w| -- (sum |C|) max|g_out[node]| max|x[src]| max|Y_e| -- so that no entry can
 * overflow fp16 whatever the model's feature magnitudes are; the two row maxima come from this kernel. */
int snet_row_absmax(const float *x, int64_t n_rows, int32_t dim, float *out, void *stream);
/* the same for n (1 .. 8) matrices x[j][n_rows[j], dims[j]] -> out[j] in one launch (both hosts: the source-row bounds of all
 * interaction layers at the start of the reverse pass); matrices with n_rows[j] <= 0 are skipped */
int snet_row_absmax_multi(const float *const *x, const int64_t *n_rows, const int32_t *dims, float *const *out, int32_t n,
                          void *stream);
/* out[r] = mult * ||x[r,:]||_2.  With mult = the largest row norm of a linear map's matrix this bounds every entry of
 * the map's output row (Cauchy-Schwarz): both hosts bound g_out = SI2^T g_y this way from the 5x narrower g_y instead
 * of reading g_out[N, dmid] once more (g_rowmax of snet_conv_bwd_fused may be ANY upper bound of max|g_out[node]|). */
int snet_row_norm2(const float *x, int64_t n_rows, int32_t dim, float mult, float *out, void *stream);
/* per-edge gradients given g_out[n_dst,dout]: g_w[E,wn] (overwritten), g_sh[E,nsh] (ACCUMULATED,
 * so one buffer collects all layers) and, if g_xe != NULL, this edge's contribution to the gradient
 * of its source row, g_xe[E,dx] (overwritten; sum it per source node with snet_segment_sum_rows --
 * 1.9 KB/edge written once instead of the 12.5 KB/edge g_out gathers of snet_conv_bwd_node).    */
int snet_conv_bwd_edge(const snet_conv_plan *plan, const float *x, const float *sh, const float *w,
                       const int32_t *w_row, const int32_t *row_ptr, const int32_t *src, int64_t n_dst, float scale,
                       const float *g_out, float *g_w, float *g_xe, float *g_sh, void *stream);
/* same, but the spherical-harmonic gradient is contracted with dsh[E,nsh,3] (snet_edge_embed_fwd)
 * inside the kernel and ACCUMULATED into g_vec[E,3]: 3 instead of nsh values per edge cross the
 * wavefront reduction.  This is the variant the whole-model engine uses. */
int snet_conv_bwd_edge_vec(const snet_conv_plan *plan, const float *x, const float *sh, const float *dsh,
                           const float *w, const int32_t *w_row, const int32_t *row_ptr, const int32_t *src,
                           int64_t n_dst, float scale, const float *g_out, float *g_w, float *g_xe, float *g_vec,
                           void *stream);
/* source-node gradient g_x[n_src,dx] (overwritten) via the source-sorted edge
 * permutation: col_ptr[n_src+1], eperm[E] (edge ids grouped by source), dst[E].
 * Deterministic replacement of the scatter-add autograd performs for x[src]. */
int snet_conv_bwd_node(const snet_conv_plan *plan, const float *sh, const float *w, const int32_t *w_row,
                       const int32_t *col_ptr,
                       const int32_t *eperm, const int32_t *dst, int64_t n_src, float scale,
                       const float *g_out, float *g_x, void *stream);

/* out[s,:] = sum_{k in [seg_ptr[s], seg_ptr[s+1])} x[perm[k],:]  (deterministic segmented row sum;
 * with col_ptr/eperm it turns g_xe[E,dx] into g_x[n_src,dx], the reverse of the x[src] gather) */
int snet_segment_sum_rows(const float *x, const int32_t *seg_ptr, const int32_t *perm, int64_t n_seg, int32_t dim,
                          float *out, void *stream);

/* ---- a5: equivariant gate ------------------------------------------------
 * replaces EquivariantGate.forward, sevenn/nn/equivariant_gate.py:57-59     */
typedef struct snet_gate_seg {
  int32_t kind;     /* 0: scalars -> act(s)*cst ; 1: gated[m,u] * (act(gate[u])*cst) */
  int32_t in_off;   /* offset in the gate input row */
  int32_t out_off;  /* offset in the output row */
  int32_t mul;
  int32_t l;
  int32_t gate_off; /* kind 1: offset of this segment's gate scalars in the input row */
  int32_t act;      /* 0 silu, 1 tanh */
  float cst;        /* normalize2mom constant of act */
} snet_gate_seg;
#define SNET_MAX_GATE_SEGS 16
/* addend (nullable, [n_nodes, dim_in]): the self-connection term of SelfConnectionOutro
 * (self_connection.py:118-138); y += addend is applied in place before the gate, so y holds the gate
 * input that snet_gate_bwd needs. */
int snet_gate_fwd(float *y, const float *addend, float *out, int64_t n_nodes, int32_t dim_in, int32_t dim_out,
                  const snet_gate_seg *segs_host, int32_t n_segs, void *stream);
int snet_gate_bwd(const float *y, const float *g_out, float *g_y, int64_t n_nodes, int32_t dim_in,
                  int32_t dim_out, const snet_gate_seg *segs_host, int32_t n_segs, void *stream);
/* snet_gate_bwd that also returns row_norm[n] = norm_mult * ||g_y[n]||_2 (nullable), the bound snet_row_norm2 gives: the rows
 * are in registers anyway, one pass over g_y less per layer. */
int snet_gate_bwd_norm(const float *y, const float *g_out, float *g_y, int64_t n_nodes, int32_t dim_in, int32_t dim_out,
                       const snet_gate_seg *segs, int32_t n_segs, float norm_mult, float *row_norm, void *stream);

/* ---- a6: species embedding (one-hot @ W == table lookup) ------------------
 * replaces OnehotEmbedding + first IrrepsLinear, node_embedding.py:44-53,
 * model_build.py:505-521.  out[n,:] = table[types[n],:]                     */
int snet_embed_rows(const float *table, const int32_t *types, float *out, int64_t n_nodes, int32_t dim,
                    void *stream);

/* ---- a4.1 and friends: y += x ; column permutation (layout conversion) --- */
int snet_add_inplace(float *y, const float *x, int64_t n, void *stream);
/* y[r, :] += bias[:] for r < n_rows: the constant a multi-modal linear's one-hot inputs contribute for a
 * fixed fidelity channel (IrrepsLinear._patch_modal_to_data, sevenn/nn/linear.py:72-92) */
int snet_add_row_bias(float *y, const float *bias, int64_t n_rows, int32_t dim, void *stream);
int snet_permute_cols(const float *x, const int32_t *col_idx, float *out, int64_t n_rows, int32_t dim,
                      void *stream); /* out[r,c] = x[r,col_idx[c]] */

/* ---- a7/a8: per-species rescale + total energy ---------------------------
 * replaces (SpeciesWise)Rescale.forward scale.py:53-56,155-162 and AtomReduce
 * linear.py:127-141.  e_atom[n] = e[n]*scale[t]+shift[t] (n_scale==1: global);
 * *energy (device double) = sum over the first n_nodes rows (deterministic).  */
int snet_rescale_reduce(const float *e_scaled, const int32_t *types, const float *scale, const float *shift,
                        int32_t n_scale, int64_t n_nodes, float *e_atom, double *energy, void *stream);

/* out[i] = in[i] + delta over n int32 entries (tile pointers of a sub-list of the reverse kernels' tile list) */
int snet_i32_shift(const int32_t *in, int32_t delta, int32_t *out, int64_t n, void *stream);

/* Folded readout (a3 + a7 + a8 in one pass): the reference's two readout linears (reduce_input_to_hidden,
 * reduce_hidden_to_energy; sevenn/model_build.py, nn/linear.py:94-100) have no nonlinearity between them, so
 * a host may fold them to one vector v[dim] and constant c (fp64, at load time).  For the first n rows:
 * e_atom[i] = (x[i] . v + c) * scale[t] + shift[t] with the dot product and the rescale in fp64,
 * *energy (device double) = their deterministic fp64 sum.  v is a DEVICE pointer to doubles.  */
int snet_readout_energy(const float *x, int64_t n_nodes, int32_t dim, const double *v, double c, const int32_t *types,
                        const float *scale, const float *shift, int32_t n_scale, float *e_atom, double *energy,
                        void *stream);
/* its reverse: g_x[i, k] = scale[type_i] * v[k]  (dE/dx of the folded readout)  */
int snet_readout_grad(const double *v, int32_t dim, const int32_t *types, const float *scale, int32_t n_scale,
                      int64_t n_nodes, float *g_x, void *stream);

/* ---- a9/a11: forces and virial from dE/d edge_vec --------------------------
 * replaces ForceStressOutputFromEdge.forward force_output.py:189-228 and the
 * host loops of pair_e3gnn.cpp:210-270.  Over n_nodes rows (locals + ghosts):
 *   F[i]   = sum_{e: center(e)=i} g[e] - sum_{e: neighbor(e)=i} g[e]
 *   vir[i] = -sum_{e: neighbor(e)=i} (rx gx, ry gy, rz gz, rx gy, ry gz, rz gx)
 * in-edges via row_ptr (edges sorted by center), out-edges via col_ptr/eperm.
 * virial_atom may be NULL; virial_total[6] (device double) may be NULL.       */
int snet_edge_force(const float *g_vec, const float *edge_vec, const int32_t *row_ptr, const int32_t *col_ptr,
                    const int32_t *eperm, int64_t n_nodes, int64_t n_edges, float *forces, float *virial_atom,
                    double *virial_total, void *stream);

/* ---- a12: halo pack / unpack ---------------------------------------------
 * replaces PairE3GNNParallel::pack_forward_comm_gnn / unpack_reverse_comm_gnn,
 * pair_e3gnn_parallel.cpp:747-911 (index_select / scatter / scatter-add).
 *   gather:      out[i,:]      = x[idx[i],:]
 *   scatter_add: y[idx[i],:]  += x[i,:]       (idx unique within one call)   */
int snet_gather_rows(const float *x, const int32_t *idx, float *out, int64_t n_idx, int32_t dim, void *stream);
int snet_scatter_add_rows(const float *x, const int32_t *idx, float *y, int64_t n_idx, int32_t dim,
                          void *stream);

/* ---- f1: neighbor list on the GPU ---------------------------------------------
 * replaces the host graph build unlabeled_atoms_to_graph / _graph_build_{ase,matscipy}
 * (sevenn/train/dataload.py:32-129) and yields the CSR-by-center edge layout directly.
 * cell_host[9] row-major lattice vectors (HOST), positions / wrapped positions fp64 (device).
 * pbc_host[3] (HOST, nullable = fully periodic): open axes are neither wrapped nor imaged; frac_range_host[6] (HOST,
 * required with an open axis) = lower (3) and upper (3) bound of the atoms' fractional coordinates, read for open axes.
 * A periodic cell may be thinner than the cutoff (it then meets itself through ceil(cutoff / height) images).
 * Sequence: snet_nl_grid (bins per axis) -> snet_nl_bin (wrapped positions, image index, bin id)
 * -> [caller sorts atoms by bin id: order[], bin_start[]] -> snet_nl_count -> [exclusive scan:
 * row_ptr] -> snet_nl_fill (src, center, edge_vec fp32, optional image shifts).              */
int snet_nl_grid(const double *cell_host, double cutoff, const int32_t *pbc_host, const double *frac_range_host,
                 int32_t *nbins_host);
int snet_nl_bin(const double *cell_host, double cutoff, const int32_t *pbc_host, const double *frac_range_host,
                const double *pos, int64_t n_atoms, double *wpos, int32_t *wrap, int32_t *cell_id, void *stream);
int snet_nl_count(const double *cell_host, double cutoff, const int32_t *pbc_host, const double *frac_range_host,
                  const double *wpos, const int32_t *cell_id, const int32_t *order, const int32_t *bin_start,
                  int64_t n_atoms, int32_t *count, void *stream);
int snet_nl_fill(const double *cell_host, double cutoff, const int32_t *pbc_host, const double *frac_range_host,
                 const double *wpos, const int32_t *wrap, const int32_t *cell_id, const int32_t *order,
                 const int32_t *bin_start, int64_t n_atoms, const int32_t *row_ptr, int32_t *src, int32_t *center,
                 float *edge_vec, int32_t *shifts, void *stream);

/* ---- whole-model sequencer ------------------------------------------------------------------
 * replaces, for a native (C++) host, `model.forward(input_dict)` + `torch::autograd::grad(...)` of
 * the LAMMPS pair styles (sevenn/pair_e3gnn/pair_e3gnn.cpp:200-207, pair_e3gnn_parallel.cpp:424-503)
 * and the TorchScript archive they load (`torch::jit::load`, pair_e3gnn.cpp:356).  The model comes
 * from a `.snet` file written by sevennet_amd.model_file.write_model_file (the analogue of
 * `sevenn get_model`, sevenn/scripts/deploy.py:16-76).  One evaluation runs the op sequence above on
 * the caller's stream out of a grow-only device arena owned by the model: no per-step allocation
 * once the largest system has been seen.                                                        */
typedef struct snet_model snet_model;
int snet_model_load(const char *path, snet_model **model);
int snet_model_load_memory(const void *blob, int64_t n_bytes, snet_model **model);
void snet_model_destroy(snet_model *model);
/* cutoff, species count, number of interaction layers and (comm_dims[t], t < max_layers) the row
 * width of the ghost exchange before layer t -- what pair_e3gnn_parallel.cpp:571-660 reads from the
 * archive's extra files ("cutoff", "comm_size", ...).  Any output pointer may be NULL. */
int snet_model_info(const snet_model *model, float *cutoff, int32_t *n_species, int32_t *n_layers,
                    int32_t *comm_dims, int32_t max_layers);
/* value of one metadata key of the model file into value[capacity] (NUL-terminated); keys as in the
 * reference's deployed model (pair_e3gnn.cpp:321-330): chemical_symbols_to_index, cutoff, num_species,
 * model_type, version, dtype.  rc 3: no such key. */
int snet_model_meta(const snet_model *model, const char *key, char *value, int32_t capacity);
/* Ghost exchange hooks (pair_e3gnn_parallel.cpp:369,435; comm_brick.cpp:1057-1123):
 *   forward(user, x[n_total,dim], ...)  fill rows n_local.. with their owners' rows
 *   reverse(user, gx[n_total,dim], ...) add rows n_local.. into their owners' rows
 * x is a DEVICE pointer, work must be ordered on `stream`; return non-zero to abort the evaluation.
 * Needed iff an evaluation has n_total > n_local.  fold_forces != 0: the engine also calls `reverse`
 * on forces[n_total,3] (and virial_atom) so ghost rows end up in their owners; 0: ghost rows are
 * left to the host (LAMMPS folds them itself with reverse_comm when newton_pair is on). */
typedef int (*snet_halo_fn)(void *user, float *x, int64_t n_total, int64_t n_local, int32_t dim, void *stream);
/* Topology cache: with it on, snet_model_eval keeps what depends on the edge list only (tile list of the reverse kernels --
 * its construction reads a count back, i.e. synchronises the stream --, the edges grouped by source, per-species row lists)
 * across evaluations for as long as the caller passes the SAME device index arrays (row_ptr, src, eperm, w_row; same counts)
 * and has not called snet_model_topology_changed.  A caller that rewrites those arrays in place must call it.  Off by default.
 * snet_model_eval_syncs: stream synchronisations issued inside snet_model_eval since the model was loaded (diagnostic). */
/* Bricks of a spatial decomposition that number their local atoms INTERIOR FIRST (rows [0, n_interior) have no ghost source;
 * sevennet_amd.parallel.BrickGraph.n_interior) let the sequencer run the interior rows of the fused convolutions while the
 * ghost exchange of the layer is in flight (library halo installed with snet_model_set_rccl_halo; second stream inside the
 * model).  0 (default) = no split.  Applies to the following evaluations until set again.                              */
int snet_model_set_interior(snet_model *model, int64_t n_interior);
int snet_model_set_topology_cache(snet_model *model, int32_t enable);
int snet_model_topology_changed(snet_model *model);
int64_t snet_model_eval_syncs(const snet_model *model);
int snet_model_set_halo(snet_model *model, snet_halo_fn forward, snet_halo_fn reverse, void *user,
                        int32_t fold_forces);
/* ---- a12, native: the ghost exchange itself, on RCCL (xGMI point-to-point), no host staging --------------
 * What a C++ host (LAMMPS pair style, any MD driver) installs instead of writing its own hooks; replaces the
 * reference's PairE3GNNParallel::{pack,unpack}_{forward,reverse}_comm_gnn (pair_e3gnn_parallel.cpp:747-911) and the
 * float overloads of CommBrick::forward_comm / reverse_comm (comm_brick.cpp:1057-1123): six sequential blocking
 * MPI swaps (with ghost-of-ghost forwarding and optional host staging, :806-809) become ONE ncclGroup of
 * send / recv pairs per call -- every peer concurrently, each over its own xGMI link, on the caller's stream.
 *   snet_rccl_unique_id / snet_rccl_comm_create   one communicator per process group: rank 0 makes the 128-byte id,
 *       the host broadcasts it (MPI_Bcast, torch.distributed ...), every rank creates its communicator
 *   snet_halo_create   the exchange plan of one decomposition: send_counts[world] rows go to each peer, taken from
 *       local rows send_idx (HOST int32, concatenated in peer order); recv_counts[world] ghost rows arrive from
 *       each peer and land contiguously, in peer order, behind the local rows -- or, with recv_perm (HOST
 *       int32[n_ghost], nullable), the k-th row of that peer-ordered stream is the host's ghost row recv_perm[k]
 *       (hosts that number their ghost nodes in their own order, e.g. a LAMMPS pair style)
 *   snet_halo_forward / snet_halo_reverse   have the snet_halo_fn signature (user = the snet_halo*): forward fills
 *       ghost rows, reverse adds ghost rows into their owners (received rows are summed per target row in fixed
 *       peer order: deterministic).  snet_model_set_rccl_halo installs both on a model.
 * RCCL is bound with dlopen at first use: libsnet_hip.so has no link-time dependency on it.                        */
typedef struct snet_halo snet_halo;
int snet_rccl_unique_id(void *id128);
int snet_rccl_comm_create(const void *id128, int32_t world, int32_t rank, void **comm);
void snet_rccl_comm_destroy(void *comm);
/* snet_rccl_available: 1 if librccl.so can be bound in this process -- local and non-collective, so that ranks can agree on the
 * transport BEFORE entering the collective snet_rccl_comm_create.  snet_rccl_comm_info: the size and this process's rank as RCCL
 * itself reports them (ncclCommCount / ncclCommUserRank): a host's start-up check that the communicator spans the ranks it
 * thinks it does. */
int snet_rccl_available(void);
int snet_rccl_comm_info(void *comm, int32_t *world_out, int32_t *rank_out);
int snet_rccl_allreduce_sum_f64(void *comm, double *dev_values, int64_t n, void *stream);
int snet_halo_create(void *comm, int32_t world, int32_t rank, const int32_t *send_counts, const int32_t *send_idx_host,
                     const int32_t *recv_counts, const int32_t *recv_perm_host, snet_halo **out);
void snet_halo_destroy(snet_halo *halo);
int64_t snet_halo_ghost_rows(const snet_halo *halo);
int64_t snet_halo_send_rows(const snet_halo *halo);
int snet_halo_forward(void *halo, float *x, int64_t n_total, int64_t n_local, int32_t dim, void *stream);
int snet_halo_reverse(void *halo, float *gx, int64_t n_total, int64_t n_local, int32_t dim, void *stream);
/* snet_halo_reverse in two halves, for hosts that overlap the exchange with work that still WRITES the local rows
 * (interior / boundary split of the convolution): _exchange reads the ghost rows gx[n_local ..] only and stages what the peers
 * return inside the halo; _accumulate adds the staged rows into gx[.. n_local].  One exchange may be staged at a time.   */
int snet_halo_reverse_exchange(void *halo, const float *gx, int64_t n_total, int64_t n_local, int32_t dim, void *stream);
int snet_halo_reverse_accumulate(void *halo, float *gx, int32_t dim, void *stream);
int snet_model_set_rccl_halo(snet_model *model, snet_halo *halo, int32_t fold_forces);
/* In-process stand-in for the RCCL communicator -- TEST infrastructure for one GPU: W host threads play W ranks; a
 * send posts {device pointer, ready event}, the matching receive copies device to device on the receiver's stream
 * and acknowledges with an event the sender's stream waits on (ncclSend / ncclRecv inside one ncclGroup: FIFO per
 * ordered pair, the group end blocks until the peers arrived).  A communicator made by snet_loopback_comm_create is
 * accepted wherever one from snet_rccl_comm_create is (snet_halo_create ...; not the all-reduce) and is destroyed with
 * snet_rccl_comm_destroy; everything above the transport is the code the RCCL path runs.                          */
int snet_loopback_hub_create(int32_t world, void **hub);
void snet_loopback_hub_abort(void *hub);
void snet_loopback_hub_destroy(void *hub);
int snet_loopback_comm_create(void *hub, int32_t rank, void **comm);

/* One energy/force evaluation.  Device inputs: types[n_total] species index, row_ptr[n_local+1] /
 * src[E] edges sorted by center (CSR), col_ptr[n_total+1] / eperm[E] the same edges grouped by source,
 * edge_vec[E,3] = r_src - r_center.  types_host[n_local] (HOST, may be NULL for models without a
 * per-species self-connection).  w_row[E] / pair_edge[n_pairs] (snet_edge_pairs; both NULL and
 * n_pairs 0 = one radial-weight row per directed edge).  Device outputs, each nullable: energy (double), e_atom[n_local],
 * dE_dr[E,3], forces[n_total,3] (ghost rows folded into owners when halo hooks are set),
 * virial[6] (double, xx yy zz xy yz zx = -sum r (x) dE/dr), virial_atom[n_total,6].             */
int snet_model_eval(snet_model *model, int64_t n_total, int64_t n_local, int64_t n_edges, const int32_t *types,
                    const int32_t *types_host, const int32_t *row_ptr, const int32_t *src, const int32_t *col_ptr,
                    const int32_t *eperm, const float *edge_vec, const int32_t *w_row, const int32_t *pair_edge,
                    int64_t n_pairs, double *energy, float *e_atom, float *dE_dr, float *forces, double *virial,
                    float *virial_atom, void *stream);

/* ---- undirected pairs: one radial-weight row per pair --------------------------------------------
 * The radial MLP's input (Bessel x cutoff of |r|, edge_embedding.py:101-160) is the same for the two
 * directed edges i->j and j->i, so w[e] == w[rev(e)].  Given the center-sorted edge list of
 * n_local owned atoms, this finds each edge's reverse (same atom pair, opposite vector within
 * 2e-5) and numbers the undirected pairs: w_row[e] in [0, n_pairs) is the row edge e reads,
 * pair_edge[p] is one edge of pair p (whose embedding row feeds the MLP).  Edges whose source is a
 * ghost (row on another rank) form singleton pairs.  Evaluate the radial MLP on the n_pairs gathered
 * embedding rows and pass w_row to the snet_conv_* calls: half the MLP work and half the w bytes for
 * a bulk cell, identical results.  Device in/out; *n_pairs on the HOST (the call synchronises). */
int snet_edge_pairs(const int32_t *row_ptr, const int32_t *src, const float *edge_vec, int64_t n_local,
                    int64_t n_edges, int32_t *w_row, int32_t *pair_edge, int64_t *n_pairs, void *stream);

/* ---- MD host: LAMMPS-style neighbor list in, forces accumulated out --------------------------
 * replaces the body of PairE3GNN::compute (sevenn/pair_e3gnn/pair_e3gnn.cpp:74-289) and the graph
 * build + force epilogue of PairE3GNNParallel::compute (pair_e3gnn_parallel.cpp:194-306,459-506).
 * All arrays are HOST pointers exactly as a pair style holds them:
 *   inum, ilist[inum], numneigh[nall] (indexed by atom), firstneigh[nall] (rows of a FULL list,
 *   entries may carry special-bond bits), x[nall,3] = &atom->x[0][0], type[nall] (1-based),
 *   tag[nall] (tag_bytes 4 or 8), type_map[ntypes+1] = the pair style's `map` (type -> species).
 * ghost_mode 0 (pair e3gnn): a neighbor is aliased to the local atom with the same tag, others are
 *   dropped (tag_map, :102,147-149) -- one process holding the whole periodic cell.
 * ghost_mode 1 (pair e3gnn/parallel): every ghost identity is a graph node after the inum locals
 *   (tag_to_graph_idx, pair_e3gnn_parallel.cpp:262-287); set the model's halo hooks first (with
 *   fold_forces 0: ghost forces are added to f[ghost] for LAMMPS' reverse_comm).  node_to_atom_out
 *   (nullable, capacity nall) receives node -> atom index BEFORE the model runs, for those hooks.
 * The neighbor filter (r^2 < cutoff^2 in fp64, edge_vec = x[j]-x[i] rounded to fp32), the CSR /
 * source grouping, the model and the force / virial reduction all run on `stream`; results are
 * ADDED to f[nall,3], *eng, virial[6] (LAMMPS order xx yy zz xy xz yz), eatom[nall], vatom[nall,6]
 * as ev_tally-free pair styles do.  Blocking: returns after the results are on the host.        */
typedef struct snet_md_host snet_md_host;
int snet_md_create(snet_model *model, snet_md_host **host);
/* node -> atom index exactly as snet_md_compute will number the graph nodes for these arrays (capacity nall);
 * lets a pair style lay out its ghost exchange (snet_halo_create) when the neighbor list was rebuilt */
int snet_md_nodes(int32_t inum, const int32_t *ilist, int32_t nall, const void *tag, int32_t tag_bytes,
                  int32_t ghost_mode, int32_t *node_to_atom_out, int64_t *n_nodes_out);
void snet_md_destroy(snet_md_host *host);
/* One-shot hint: the NEXT snet_md_compute call gets the same neighbor list as the previous one (LAMMPS `neighbor->ago > 0`:
 * same inum / ilist / numneigh / firstneigh / nall / tags / types) -- the flattened list, node maps and species uploaded then
 * are reused, only the positions travel (the edge set inside the cutoff is still re-derived from them every step).       */
int snet_md_list_unchanged(snet_md_host *host);
int snet_md_compute(snet_md_host *host, int32_t inum, const int32_t *ilist, const int32_t *numneigh,
                    const int32_t *const *firstneigh, int32_t nall, const double *x, const int32_t *type,
                    const void *tag, int32_t tag_bytes, const int32_t *type_map, int32_t ntypes, int32_t ghost_mode,
                    int32_t eflag_atom, int32_t vflag, int32_t vflag_atom, double *f, double *eng, double *virial,
                    double *eatom, double *vatom, int32_t *node_to_atom_out, int64_t *n_nodes_out,
                    int64_t *n_edges_out, void *stream);

/* ---- DFT-D3 dispersion -----------------------------------------------------------------------------
 * What the reference's CUDA library behind sevenn.calculator.D3Calculator computes (pair_d3_for_ase.cu), as own HIP
 * kernels (csrc/snet_d3.hip: one workgroup per atom over its (neighbour, lattice translation) row, fp64, deterministic
 * reductions), behind the same call sequence as the reference's C-ABI (pair_d3_for_ase.cu:2034-2082):
 *   pair_init          -> snet_d3_create            pair_run_coeff     -> snet_d3_set_tables (published tables, from the
 *   pair_set_atom      -> snet_d3_set_atoms                               blob sevennet_amd/data/d3_params.npz) + set_atoms
 *   pair_set_domain    -> snet_d3_set_cell          pair_run_compute   -> snet_d3_compute
 *   pair_run_settings  -> snet_d3_settings          pair_get_energy / _force / _stress -> snet_d3_energy / _forces / _stress
 *   pair_fin           -> snet_d3_destroy
 * Units at the boundary: A, eV, eV/A, eV/A^3; cutoffs in bohr^2 like the reference (vdw 9000, cn 1600).
 *   snet_d3_set_tables  r0ab[94*94] (A), c6ab[n_c6*5] (C6, Z_i + 100 ref_i, Z_j + 100 ref_j, CN_i, CN_j), r2r4[94], rcov[94]
 *   snet_d3_settings    damping 0 = damp_zero, 1 = damp_bj; func5 = (s6, rs6, s18, rs18, alp) of the functional
 *   snet_d3_set_cell    cell[9]: lattice vectors as rows (any orientation: no LAMMPS-style rotation needed), pbc[3]
 *   snet_d3_stress      [9] = dE/d(strain) / volume, row-major symmetric (ASE sign convention)                          */
typedef struct snet_d3 snet_d3;
int snet_d3_create(snet_d3 **out);
void snet_d3_destroy(snet_d3 *d3);
int snet_d3_set_tables(snet_d3 *d3, const double *r0ab, const double *c6ab, int64_t n_c6, const double *r2r4, const double *rcov);
int snet_d3_settings(snet_d3 *d3, double vdw_cutoff_au2, double cn_cutoff_au2, int32_t damping, const double *func5);
int snet_d3_set_atoms(snet_d3 *d3, int32_t n, const int32_t *atomic_numbers, const double *positions);
int snet_d3_set_cell(snet_d3 *d3, const double *cell9, const int32_t *pbc3);
int snet_d3_compute(snet_d3 *d3, void *stream);
double snet_d3_energy(const snet_d3 *d3);
const double *snet_d3_forces(const snet_d3 *d3);
const double *snet_d3_stress(const snet_d3 *d3);
const double *snet_d3_coordination_numbers(const snet_d3 *d3);

#ifdef __cplusplus
}
#endif
#endif /* SNET_HIP_H */
