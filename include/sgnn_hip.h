/*
 * sgnn_hip.h — C ABI of libsgnn_hip.so, the MI355X (gfx950) sparse generative 3D
 * convolution hot path that sits behind SG-NN's `sparseconvnet` operator surface.
 *
 * Boundary being replaced.  The reference (angeladai/sgnn) reaches its sparse-op
 * arithmetic only through `import sparseconvnet as scn` (torch/model.py:7; call
 * sites torch/model.py:31-47, 178-188, 253-257, 296, 380).  Upstream's own native
 * boundary is a pybind11 module (`sparseconvnet.SCN`: class Metadata_3 plus
 * <Op>_updateOutput / <Op>_backward free functions taking at::Tensor&), i.e. not
 * a C ABI.  This header is the C ABI that takes its place; each entry point names
 * the scn operation / reference line it serves.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless the comment says "host";
 *  - every function is asynchronous on `stream` (a hipStream_t passed as void*),
 *    performs no allocation and no device synchronisation;
 *  - the caller owns all memory (features, tables, hash storage, workspaces);
 *    workspace sizes come from the *_ws_bytes() queries;
 *  - return value: 0 = SGNN_OK, negative = error; sgnn_last_error() returns a
 *    thread-local message.  Nothing throws across the ABI;
 *  - coordinates are int32[4] = {z, y, x, batch} per site (16-byte rows), each
 *    spatial coordinate in [0, 65535], batch in [0, 32767]
 *    (layout contract: torch/scene_dataloader.py:13-36);
 *  - feature matrices are row-major float32 (N, C), contiguous.
 *  - neighbour / children tables are int32 [K][ld] (offset-major), -1 = no rule.
 *    A "rulebook" in upstream's sense is { (k, table[k][j], j) : table[k][j] >= 0 }.
 *  - capacity mode (device-side row counts): wherever a function takes a trailing `const int64_t *n_dev`
 *    (or nf_dev / m_dev ...), a non-NULL pointer makes the kernels read the row count from device memory at
 *    run time, clamped to the host value n, which then is only the CAPACITY the launch and the buffers are
 *    sized for.  The generative path (torch/model.py:229-247, 315-336) decides every level's row count from
 *    predicted logits; with the counts left on the device a whole training step has no host read-back and a
 *    fixed launch sequence, i.e. it can be captured in a HIP graph and replayed (sgnn_amd/train.py GraphStep).
 *    NULL = the host value is exact (the classic path).  Counts that exceed a capacity are clamped and
 *    SGNN_STATUS_OVERFLOW is raised in the status word; sgnn_adam_flat then leaves the parameters untouched.
 */
#ifndef SGNN_HIP_H
#define SGNN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGNN_OK 0
#define SGNN_EINVAL (-1)    /* bad argument */
#define SGNN_EHIP (-2)      /* HIP runtime / launch failure */
#define SGNN_EOVERFLOW (-3) /* size exceeds an internal 31-bit index */
#define SGNN_ENOWS (-4)     /* workspace too small */

/* status bits written (atomically OR-ed) into the device `status` word */
#define SGNN_STATUS_COORD_RANGE 1 /* a coordinate was outside the supported range ([0, 65535], batch [0, 32767]) — or, from
                                   * sgnn_rulebook_subm3_volume, outside the volume bounds the caller declared */
#define SGNN_STATUS_DUPLICATE 2   /* InputLayer(mode=0) saw the same site twice */
#define SGNN_STATUS_OVERFLOW 4    /* capacity mode: a level produced more rows than its buffers hold (step discarded) */

typedef void *sgnn_stream_t; /* hipStream_t */

const char *sgnn_last_error(void);
int sgnn_version(void);
/* name of the gfx target the kernels were compiled for ("gfx950") */
const char *sgnn_arch(void);

/* ---------------------------------------------------------------------------
 * Voxel hash grid — scn.InputLayer(3, size, mode=0) (torch/model.py:31,178,185,253)
 * ------------------------------------------------------------------------- */

/* capacity (power of two, >= 2n, >= 1024) of the open-addressing table for n sites */
int64_t sgnn_hash_capacity(int64_t n);

/* int64 [z,y,x,b] rows (the reference's LongTensor locs) -> int32 rows; sets
 * SGNN_STATUS_COORD_RANGE if a value is out of range */
int sgnn_coords_from_i64(const int64_t *locs, int64_t n, int32_t *coords, int32_t *status,
                         const int64_t *n_dev, sgnn_stream_t stream);
/* and back (metadata.getSpatialLocations, torch/model.py:380) */
int sgnn_coords_to_i64(const int32_t *coords, int64_t n, int64_t *locs, const int64_t *n_dev,
                       sgnn_stream_t stream);

/* build key->row table: keys[cap] (uint64), vals[cap] (int32).  The function
 * clears the table itself.  Sets SGNN_STATUS_DUPLICATE on repeated sites. */
int sgnn_hash_build(const int32_t *coords, int64_t n, uint64_t *keys, int32_t *vals, int64_t cap,
                    int32_t *status, const int64_t *n_dev, sgnn_stream_t stream);

/* row of each query site, or -1 (concat_skip join, torch/model.py:338-355) */
int sgnn_hash_lookup(const uint64_t *keys, const int32_t *vals, int64_t cap, const int32_t *query,
                     int64_t m, int32_t *rows, const int64_t *m_dev, sgnn_stream_t stream);

/* ---------------------------------------------------------------------------
 * Measurement switches: ONE table.  Every kernel variant / launch fusion that was ever A/B-measured is selected by a field of
 * this struct; defaults are the measured winners.  No field changes results beyond fp32 / fp64 summation order (each entry
 * says what stays bit-identical).  No reference counterpart.  Set by name: sgnn_tune_set("conv_wide_epi", 0) returns the
 * previous value, SGNN_TUNE_UNKNOWN (with sgnn_last_error) for an unknown name or a value out of range; sgnn_tune_get reads;
 * sgnn_tune_names = the comma-separated field names; sgnn_tune_current = the live table (read-only).  Process-global, not
 * thread-safe: set them before the work they steer (workspace sizes follow conv_bwd_fused / conv_bwd_fused_rows).
 * Host layer: sgnn_amd._lib.tune(name, value); environment SGNN_TUNE="name=value,name=value" at load.
 * ------------------------------------------------------------------------- */
#define SGNN_TUNE_UNKNOWN INT64_MIN
typedef struct sgnn_tune {
  /* levels below conv_small_rows run a latency-oriented kernel (16 rows per workgroup, the four waves split the offsets);
   * 0 = the 64-row variant of the big kernel.  The two sum the offsets in different orders (fp32 round-off).  Default 1. */
  int64_t conv_small;
  /* row count below which the small-level kernel is used.  Default 40 960. */
  int64_t conv_small_rows;
  /* levels above that run the wide-row 27-offset layers and every 8-offset walk as straight-line code (conv_unrolled.hip:
   * exact s_waitcnt counts, rule entries loaded up front); 0 = the looped kernel everywhere.  Bit-identical.  Default 1. */
  int64_t conv_unrolled;
  /* large levels: every workgroup of the rulebook walk takes as many consecutive 256-row tiles as it needs for ALL live
   * workgroups to be resident at once (no partial second round); 0 = one tile per workgroup.  Rows bit-identical; BatchNorm
   * statistics partials are summed per workgroup, so their fp64 grouping differs.  The tile count follows the workgroups of
   * the kernel the DEVICE holds at once (occupancy x compute units, queried per device on first use): statistics are
   * bit-reproducible for a given device model, driver and compiler, not across them.  Default 1. */
  int64_t conv_one_round;
  /* epilogue of the 256-row walk for output rows of 8 / 12 / 16 channels (torch/model.py:38-42, 180, 255, forward and data
   * gradient): 1 = the tile leaves the MFMA layout through a quad transpose, so the residual addend, the BatchNorm input of
   * the backward statistics and the output rows move as row-contiguous 16-byte accesses; 0 = one 4-byte access per
   * accumulator element.  Used when every row stride is a multiple of 4 floats and the bases are 16-byte aligned.  Stored
   * rows AND statistics partials bit-identical.  Default 1. */
  int64_t conv_wide_epi;
  /* row blocks a weight-gradient launch aims for (1 .. 4096; the launch is row blocks x offset groups workgroups).  Same
   * per-block sums in the same order for a given setting; the blocks' partial sums are added in block order.  Default 256. */
  int64_t conv_dw_blocks;
  /* weight gradient of a one-channel input (1 -> 8, the network's first convolution): 1 = the register-accumulating VALU
   * kernel, 0 = the MFMA kernel of the other shapes.  Different summation order (fp32 round-off).  Default 1. */
  int64_t conv_dw_c1;
  /* 1 = sgnn_prog_backward runs 16-channel 3x3x3 layers on levels of >= conv_bwd_fused_rows rows through the fused backward
   * kernel (sgnn_conv_bwd_fused: dX and dW from one gather of dy).  dX rows bit-identical, dW another fixed summation
   * order.  Default 0: 0.95x the two kernels stand-alone, +0.04 .. +0.14 ms per step (profiles/r06_fused_backward.txt). */
  int64_t conv_bwd_fused;
  int64_t conv_bwd_fused_rows;   /* >= 256.  Default 40 960. */
  /* 1 = the 3x3x3 rulebook builder hashes the voxel index of a 768-row window around each 256-row tile into LDS (global
   * table only for neighbours not found there); 0 = the global-probe kernel, faster on MI355X (96.7 vs 123.9 us at N =
   * 366 k).  Identical tables.  Default 0. */
  int64_t rulebook_lds;
  /* sgnn_rulebook_subm3_multi: 1 = all levels in one pre-fill launch and one builder launch, 0 = one sgnn_rulebook_subm3
   * per level.  Identical tables.  Default 1. */
  int64_t rulebook_multi;
  /* compactions / stride-2 levels: 1 = the write kernels sum the (<= 4096) raw block counts themselves, 0 = a scan launch
   * between the count and the write kernel.  Identical results.  Default 1. */
  int64_t scan_inline;
  /* sgnn_down2_chain_tables: 1 = the tables pass of level l and the hash insertion of level l + 1 in one launch.
   * Identical results.  Default 1. */
  int64_t chain_merged;
  /* sgnn_prog_forward / _backward: 0 switches the epilogue fusions off (conv -> AddTable, conv -> BatchNorm statistics,
   * in-place JoinTable): the per-layer launch sequence, identical arithmetic.  Default 1. */
  int64_t prog_fusion;
  /* a per-site linear head that is the only reader of a BatchNormReLU (the surface head): 1 = its data gradient dy w is
   * formed inside the two BatchNorm backward passes instead of being written and read back (bit-identical values); 0 =
   * k_linear_bwd writes it.  Applies to programs planned after the change.  Default 1. */
  int64_t prog_lin_bn;
  /* a per-site linear head whose input rows already carry a gradient when its backward pass runs (a Refinement's two heads,
   * torch/model.py:230-243): 1 = the head's kernel writes dy w + that gradient in one pass; 0 = an add launch over the level
   * follows.  Same sums (one fp32 addition per element either way).  Default 1. */
  int64_t prog_lin_add;
} sgnn_tune;
int64_t sgnn_tune_set(const char *name, int64_t value);
int64_t sgnn_tune_get(const char *name);
const char *sgnn_tune_names(void);
const sgnn_tune *sgnn_tune_current(void);

/* ---------------------------------------------------------------------------
 * Rulebooks
 * ------------------------------------------------------------------------- */

/* 3x3x3 submanifold rulebook (scn.SubmanifoldConvolution, torch/model.py:32,38,40,179,186,254):
 * nbr[k*ld + j] = row of the site at p_j + d_k, k = (dz+1)*9+(dy+1)*3+(dx+1), else -1.
 * ld >= n; entries j in [n, ld) are written as -1. */
int sgnn_rulebook_subm3(const uint64_t *keys, const int32_t *vals, int64_t cap,
                        const int32_t *coords, int64_t n, int32_t *nbr, int64_t ld,
                        const int64_t *n_dev, sgnn_stream_t stream);

/* `count` <= SGNN_RULEBOOK_MULTI_MAX such rulebooks in one pair of launches (capacity mode: every n_devs[i] != NULL):
 * the coarse levels of one hierarchy (scn.InputLayer + the Convolution(2,2) chain, torch/model.py:228-236) have
 * 0.4 k - 40 k rows each and their single launches are mostly ramp.  Each array holds one entry per level with the
 * meaning of the matching sgnn_rulebook_subm3 argument; table i equals sgnn_rulebook_subm3's for level i entry by
 * entry.  Levels with ns[i] == 0 are skipped. */
#define SGNN_RULEBOOK_MULTI_MAX 4
int sgnn_rulebook_subm3_multi(int count, const uint64_t *const *keys, const int32_t *const *vals, const int64_t *caps,
                              const int32_t *const *coords, const int64_t *ns, int32_t *const *nbrs, const int64_t *lds,
                              const int64_t *const *n_devs, sgnn_stream_t stream);

/* The same table through a dense index volume (volume[((b*Z + z)*Y + y)*X + x] = row, -1 elsewhere): one coalesced
 * 4-byte read per neighbour instead of a hash probe.  `volume` is a persistent workspace of volume_entries int32 that
 * the caller fills with -1 ONCE; the call marks this level's rows, builds the table and clears the rows again, so the
 * volume is all -1 on return.  Blocks with batch index >= volume_entries / (Z*Y*X) and positions outside [0, dims) go
 * through the hash grid (keys / vals / cap): the table equals sgnn_rulebook_subm3's for every input.
 * One volume serves one stream at a time (calls are stream-ordered by the caller).  Measured at N = 366 k (64^3 x 32
 * blocks): 18 us against 97 us for sgnn_rulebook_subm3 (profiles/r02j_rulebook.txt). */
int sgnn_rulebook_subm3_dense(const uint64_t *keys, const int32_t *vals, int64_t cap, const int32_t *coords, int64_t n,
                              int dim_z, int dim_y, int dim_x, int32_t *volume, int64_t volume_entries, int32_t *nbr,
                              int64_t ld, const int64_t *n_dev, sgnn_stream_t stream);
/* The same table for a level whose sites are known to lie inside the volume (the generated levels: children of a dense
 * coarse volume bounded by the model's own sizes, torch/model.py:192-207): volume only, no hash grid of the level is read —
 * or built.  A site the volume does not cover raises SGNN_STATUS_COORD_RANGE in *status. */
int sgnn_rulebook_subm3_volume(const int32_t *coords, int64_t n, int dim_z, int dim_y, int dim_x, int32_t *volume,
                               int64_t volume_entries, int32_t *nbr, int64_t ld, const int64_t *n_dev, int32_t *status,
                               sgnn_stream_t stream);

/* stride-2 / size-2 rulebook, phase 1 (scn.Convolution(...,2,2), torch/model.py:44):
 * finds the coarse active set unique(floor(p/2)) in FIRST-TOUCH order of the fine
 * rows (wave ballot + prefix-sum compaction), writes
 *   parent[i]              coarse row of fine site i
 *   coarse_coords[c*4..]   coordinates of coarse row c   (room for nf rows)
 *   ckeys/cvals (ccap)     hash grid of the coarse level (key -> coarse row)
 *   *n_coarse (device)     number of coarse sites
 * ccap = sgnn_hash_capacity(nf). */
int64_t sgnn_down2_ws_bytes(int64_t nf);
int sgnn_rulebook_down2(const int32_t *fine_coords, int64_t nf, uint64_t *ckeys, int32_t *cvals,
                        int64_t ccap, int32_t *parent, int32_t *coarse_coords, int64_t *n_coarse,
                        void *ws, int64_t ws_bytes, sgnn_stream_t stream);
/* phase 2, once the host knows n_coarse: offset-major tables
 *   children[k*ldc + c] = fine row whose parent is c and whose offset
 *                         (z&1)*4+(y&1)*2+(x&1) is k, else -1        (8 x ldc)
 *   ptable[k*ldf + i]   = parent[i] if offset(i)==k else -1          (8 x ldf)
 * (padding entries up to ldc / ldf are written as -1)
 * children drives the forward conv / unpool-backward, ptable the data-gradient. */
int sgnn_down2_tables(const int32_t *fine_coords, const int32_t *parent, int64_t nf,
                      int32_t *children, int64_t ldc, int64_t nc, int32_t *ptable, int64_t ldf,
                      const int64_t *nf_dev, const int64_t *nc_dev, sgnn_stream_t stream);

/* Phase 1 for `depth` successive levels in one submission (a U-Net's whole stride-2 pyramid), row counts staying on
 * the device: level 0 = fine_coords with *n0_dev rows (n0_dev NULL: n0 rows), at most `cap` rows; for l < depth it
 * writes parent[l] (cap ints: coarse row of every site of level l), coarse_coords[l] (cap x 4), the hash of level l+1
 * (ckeys[l], cvals[l]; ccap = sgnn_hash_capacity(cap) each) and counts_dev[l] = rows of level l+1.  ckeys .. coarse_coords
 * are HOST arrays of device pointers.  Results are identical to `depth` calls of sgnn_rulebook_down2 (first-touch
 * order); the host reads all counts with one copy and then calls sgnn_down2_tables per level.
 * Capacity mode: level_caps (HOST array, depth entries, or NULL) clamps counts_dev[l] to the capacity of the buffers
 * that will hold level l+1's features and raises SGNN_STATUS_OVERFLOW in *status when it had to. */
int64_t sgnn_down2_chain_ws_bytes(int64_t cap);
int sgnn_down2_chain(const int32_t *fine_coords, int64_t n0, const int64_t *n0_dev, int64_t cap, int depth,
                     void *const *ckeys, void *const *cvals, int64_t ccap, void *const *parent,
                     void *const *coarse_coords, int64_t *counts_dev, const int64_t *level_caps, int32_t *status,
                     void *ws, int64_t ws_bytes, sgnn_stream_t stream);

/* Capacity mode: the pyramid AND its tables in one submission — sgnn_down2_chain (row counts from *n0_dev, clamped to
 * level_caps, SGNN_STATUS_OVERFLOW) followed by sgnn_down2_tables of every level, as 3 launches per level + 2 (round 5: insert,
 * count, write kernel; the tables pass of a level shares a launch with the next level's insertion; 5 per level + 1 with
 * sgnn_tune.scan_inline = 0 / chain_merged = 0) instead of 8 per level.  children[l] is (8 x ldc_l), ldc_l = roundup256(min(level_caps[l], cap)); ptable[l] is (8 x ldf_l),
 * ldf_0 = roundup256(cap), ldf_l = ldc_{l-1}.  Only rows below roundup256(live count) of a table are written (and read). */
int64_t sgnn_down2_chain_tables_ws_bytes(int64_t cap, int depth);
int sgnn_down2_chain_tables(const int32_t *fine_coords, const int64_t *n0_dev, int64_t cap, int depth,
                            void *const *ckeys, void *const *cvals, int64_t ccap, void *const *parent,
                            void *const *coarse_coords, int64_t *counts_dev, const int64_t *level_caps,
                            void *const *children, void *const *ptable, int32_t *status, void *ws, int64_t ws_bytes,
                            sgnn_stream_t stream);

/* ---------------------------------------------------------------------------
 * Sparse convolution: out[j] = sum_k W[k]^T x[table[k][j]]   (fp32 MFMA 16x16x4)
 * serves SubmanifoldConvolution fwd (table = nbr, K = 27), Convolution(2,2) fwd
 * (table = children, K = 8) and both data-gradients:
 *   subm  dX: x := dY, table = nbr,    flags = TRANSPOSE_W | FLIP_K
 *   down2 dX: x := dY, table = ptable, flags = TRANSPOSE_W
 * w is always the layer's weight (K, c_in_layer, c_out_layer); with TRANSPOSE_W the
 * call's cin/cout are (c_out_layer, c_in_layer).
 * in_shift: feature row = table value >> in_shift (3 = features live on the
 * parents of an 8-child expansion, torch/model.py:192-207; 0 otherwise).
 * n_in = rows of x.  Table layout contract for the conv entry points: ld is a multiple of 256
 * and the padding entries table[k][n_out .. ld) are -1 (every table this library builds is so);
 * K <= 64; each slab (x, y, table) must be smaller than 4 GiB (raw-buffer addressing).
 * ------------------------------------------------------------------------- */
#define SGNN_CONV_TRANSPOSE_W 1
#define SGNN_CONV_FLIP_K 2
int sgnn_conv_fwd(const float *x, int64_t n_in, int cin, const float *w, int K, const int32_t *table,
                  int64_t ld, int64_t n_out, int cout, float *y, int flags, int in_shift,
                  sgnn_stream_t stream);

/* Generalised rulebook walk (used for the generative up-sampling convolution, see below):
 *   offset k of group g reads table row kmap[g*K + k]       (kmap NULL: row k)
 *   and gathers feature row  (entry >> in_shift)*in_mul + kadd[g*K + k]   (kadd NULL: + 0)
 *   group g uses the weight block w + g*K*cin*cout and owns output rows  row*groups + g
 * so y has n_out*groups rows.  kmap and kadd (groups*K ints each) are device arrays; table_rows =
 * number of offset rows the table really has (27 for a 3x3x3 rulebook).
 * Generative up-sampling (Refinement.n0/n1, torch/model.py:185-186,220-223): all 8 children of a site
 * carry its features (model.py:203), so SubmanifoldConvolution on the 8N children collapses, per child
 * parity g, into 8 parent-level offsets with pre-summed weights: forward = one launch with groups = 8,
 * K = 8 on the PARENT table (3.4x fewer gathers and flops than 27 offsets on 8N rows; the children
 * grid and its rulebook are never built); data gradient = K = 64 offsets with in_mul = 8, kadd = parity, run as 4 groups of 16 offsets whose
 * partial rows are added by sgnn_sum_groups (more workgroups, shorter offset walk per workgroup). */
int sgnn_conv_fwd_ex(const float *x, int64_t n_in, int cin, const float *w, int K, const int32_t *table,
                     int64_t ld, int64_t n_out, int cout, float *y, int flags, int in_shift,
                     const int32_t *kmap, const int32_t *kadd, int in_mul, int groups, int table_rows,
                     sgnn_stream_t stream);
int sgnn_conv_bwd_weight_ex(const float *x, int64_t n_in, int cin, const float *dy, int cout,
                            const int32_t *table, int64_t ld, int K, int64_t n_out, float *dw, int in_shift,
                            const int32_t *kmap, const int32_t *kadd, int in_mul, int groups, int table_rows,
                            void *ws, int64_t ws_bytes, sgnn_stream_t stream);

/* Plain rulebook walk with strided rows and a fused epilogue (what the program executor uses to remove the
 * AddTable / BatchNorm-statistics passes around a convolution, torch/model.py:33-42 and the FullyConvolutionalNet
 * blocks):
 *   x rows have stride ldx floats, y rows ldy, addend rows ld_add (0 = contiguous);
 *   y = conv + addend when addend != NULL (addend may be y itself: accumulate in place);
 *   stats = 1: partial[blk][0][c] = sum over the workgroup's rows of y, [1][c] = sum of y*y;
 *   stats = 2: y is the gradient w.r.t. a BatchNormReLU output whose INPUT rows are bn_x (stride ld_bnx) with
 *              mean / invstd / gamma / beta (gamma, beta may be NULL) and leak: partial = sum dz, sum dz*xhat with
 *              dz = y * (bn_out > 0 ? 1 : leak) — the reduction BatchNorm backward starts with.
 * partial holds sgnn_conv_stats_blocks(n_out) * 2 * cout doubles; sgnn_bn_fwd_ex / sgnn_bn_bwd_ex accept it as
 * pre_partial.  Only for the compiled (cin, cout) shapes; SGNN_EINVAL otherwise. */
int64_t sgnn_conv_stats_blocks(int64_t n_out);
int sgnn_conv_fwd_epi(const float *x, int64_t n_in, int cin, int64_t ldx, const float *w, int K,
                      const int32_t *table, int64_t ld, int64_t n_out, int cout, float *y, int64_t ldy, int flags,
                      const float *addend, int64_t ld_add, int stats, double *partial, const float *bn_x,
                      int64_t ld_bnx, const float *mean, const float *invstd, const float *gamma, const float *beta,
                      float leak, sgnn_stream_t stream);

/* weight gradient dW[k][ci][co] = sum_j x[table[k][j]][ci] * dy[j][co]; deterministic
 * two-stage reduction through the workspace. */
int64_t sgnn_conv_bwd_weight_ws_bytes(int64_t n_out, int K, int cin, int cout);
int sgnn_conv_bwd_weight(const float *x, int64_t n_in, int cin, const float *dy, int cout,
                         const int32_t *table, int64_t ld, int K, int64_t n_out, float *dw, int in_shift, void *ws,
                         int64_t ws_bytes, sgnn_stream_t stream);

/* Backward of a 3x3x3 SubmanifoldConvolution as ONE launch (round 6): data gradient and weight gradient from a single
 * gather of dy (torch/model.py:38,40,180,255 under train.py:262).  dy: (n, cout) rows with stride ld_dy; x: the layer's input
 * rows (n, cin), stride ldx; w: (27, cin, cout); table / ld: the level's neighbour table (sgnn_rulebook_subm3*);
 * dx: (n, cin) rows with stride ld_dx = sum_k dy[table[k][i]] W[26 - k]^T, with the epilogue options of sgnn_conv_fwd_epi
 * (addend, which may alias dx; stats = 2: BatchNorm-backward statistics partials, sgnn_conv_stats_blocks(n) x 2 x cin doubles);
 * dw: (27, cin, cout) = what sgnn_conv_bwd_weight returns (another fixed summation order), through per-workgroup partials in
 * ws (sgnn_conv_bwd_fused_ws_bytes) and the library's fixed-order reduce.  dx rows are bit-identical to
 * sgnn_conv_fwd_epi(flags = TRANSPOSE_W | FLIP_K).  Served shapes: sgnn_conv_bwd_fused_supported (cin = cout = 16, K = 27,
 * levels of at least sgnn_tune.conv_bwd_fused_rows rows, default 40 960); others return SGNN_EINVAL.  Row strides multiples of
 * 4 floats, bases 16-byte aligned.  n_dev: capacity mode (NULL = n is exact).  sgnn_tune.conv_bwd_fused = 1 makes
 * sgnn_prog_backward use it (default 0: stand-alone it is 0.95x the two kernels it replaces, in the step it costs
 * +0.04 .. +0.14 ms — profiles/r06_fused_backward.txt); set it before the first step (workspace sizes follow it). */
int64_t sgnn_conv_bwd_fused_ws_bytes(int64_t n, int cin, int cout);
int sgnn_conv_bwd_fused_supported(int64_t n, int cin, int cout, int K);
int sgnn_conv_bwd_fused(const float *dy, int64_t n, int cout, int64_t ld_dy, const float *x, int cin, int64_t ldx,
                        const float *w, const int32_t *table, int64_t ld, float *dx, int64_t ld_dx, const float *addend,
                        int64_t ld_add, int stats, double *partial, const float *bn_x, int64_t ld_bnx, const float *mean,
                        const float *invstd, const float *gamma, const float *beta, float leak, float *dw, void *ws,
                        int64_t ws_bytes, const int64_t *n_dev, sgnn_stream_t stream);

/* pre-summed weights of the generative up-sampling convolution and their gradient: Wc (64, cin, cout) from the layer's
 * W (27, cin, cout); dW from dWc (see sgnn_conv_fwd_ex) */
int sgnn_expand_weights(const float *w, int cin, int cout, float *wc, sgnn_stream_t stream);
int sgnn_expand_weights_bwd(const float *dwc, int cin, int cout, float *dw, sgnn_stream_t stream);

/* ---------------------------------------------------------------------------
 * scn.BatchNormReLU / BatchNormalization (torch/model.py:37,39,42,45,181,187,256)
 * leak: 0 = ReLU, 1 = plain batch norm.  momentum = fraction of OLD running value kept.
 * training: batch statistics (biased var to normalise, unbiased into running_var),
 * save_mean/save_invstd (C floats each) kept for backward.
 * ------------------------------------------------------------------------- */
int64_t sgnn_bn_ws_bytes(int64_t n, int c);
int sgnn_bn_fwd(const float *x, int64_t n, int c, const float *gamma, const float *beta,
                float *running_mean, float *running_var, float eps, float momentum, int training,
                float leak, float *save_mean, float *save_invstd, float *y, void *ws,
                int64_t ws_bytes, sgnn_stream_t stream);
int sgnn_bn_bwd(const float *x, const float *dy, int64_t n, int c, const float *gamma,
                const float *beta, const float *save_mean, const float *save_invstd, int training,
                float leak, float *dx, float *dgamma, float *dbeta, void *ws, int64_t ws_bytes,
                sgnn_stream_t stream);
/* same, with dx = (BatchNorm gradient) + addend[...]; addend may be dx itself (in-place accumulation) or NULL */
int sgnn_bn_bwd_add(const float *x, const float *dy, int64_t n, int c, const float *gamma, const float *beta,
                    const float *save_mean, const float *save_invstd, int training, float leak,
                    const float *addend, float *dx, float *dgamma, float *dbeta, void *ws, int64_t ws_bytes,
                    sgnn_stream_t stream);

/* strided rows (ldx, ldy, ... in floats; 0 = c) and optional statistics partials from a convolution epilogue
 * (pre_partial / pre_nblk, see sgnn_conv_fwd_epi): the statistics pass is skipped */
int sgnn_bn_fwd_ex(const float *x, int64_t ldx, int64_t n, int c, const float *gamma, const float *beta,
                   float *running_mean, float *running_var, float eps, float momentum, int training, float leak,
                   float *save_mean, float *save_invstd, float *y, int64_t ldy, const double *pre_partial,
                   int64_t pre_nblk, void *ws, int64_t ws_bytes, sgnn_stream_t stream);
int sgnn_bn_bwd_ex(const float *x, int64_t ldx, const float *dy, int64_t ld_dy, int64_t n, int c, const float *gamma,
                   const float *beta, const float *save_mean, const float *save_invstd, int training, float leak,
                   const float *addend, int64_t ld_add, float *dx, int64_t ld_dx, float *dgamma, float *dbeta,
                   const double *pre_partial, int64_t pre_nblk, void *ws, int64_t ws_bytes, sgnn_stream_t stream);

/* ---------------------------------------------------------------------------
 * Row movement (all pure copies / sums, fp32 rows of c floats)
 * ------------------------------------------------------------------------- */
/* Up to 8 device-to-device copies as one launch: dst[k] <- src[k], bytes[k] bytes each (host arrays of device pointers /
 * sizes; regions must not overlap each other).  train.GraphStep loads a batch into the replayed graph's static input
 * buffers with it (torch/train.py:256-262 uploads the batch there): one kernel instead of seven copies per step. */
int sgnn_copy_multi(void *const *dst, const void *const *src, const int64_t *bytes, int nregions, sgnn_stream_t stream);

/* dst[r] = src[idx[r]]  (UnPooling fwd: idx = parent; mask compaction: idx = sel) */
int sgnn_gather_rows(const float *src, int c, const int32_t *idx, int64_t m, float *dst,
                     sgnn_stream_t stream);
/* the same with the row count read from device memory (*m_dev <= m_cap): no host round trip between a mask
 * compaction and the consumers of the compacted coordinates */
int sgnn_gather_rows_dn(const float *src, int c, const int32_t *idx, const int64_t *m_dev, int64_t m_cap,
                        float *dst, sgnn_stream_t stream);
/* dst[r] = [ a[ia?ia[r]:r] | b[ib?ib[r]:r] | c[ic?ic[r]:r] ]  (negative index -> zeros, zero-channel sources skipped) and
 * its gradient (indices unique; destinations reached through an index array are zero-filled first, NULL ones skipped) */
int sgnn_concat3_rows(const float *a, int ca, const int32_t *ia, const float *b, int cb, const int32_t *ib,
                      const float *c, int cc, const int32_t *ic, int64_t m, float *dst, sgnn_stream_t stream);
int sgnn_concat3_rows_bwd(const float *ddst, int ca, const int32_t *ia, int cb, const int32_t *ib, int cc,
                          const int32_t *ic, int64_t m, float *da, int64_t na, float *db, int64_t nb, float *dc,
                          int64_t nc, sgnn_stream_t stream);
/* dst (n_dst rows, zero-filled here) ; dst[idx[r]] = src[r]   (idx unique) */
int sgnn_scatter_rows(const float *src, int c, const int32_t *idx, int64_t m, float *dst,
                      int64_t n_dst, const int64_t *m_dev, sgnn_stream_t stream);
/* dst[j] = sum_k src[table[k*ld+j]] over valid entries (UnPooling bwd: table = children) */
int sgnn_gather_sum(const float *src, int c, const int32_t *table, int64_t ld, int K,
                    int64_t n_out, float *dst, sgnn_stream_t stream);
/* dst[r*rep + t] = src[r]  (to_next_level_locs feature replication, torch/model.py:203) */
int sgnn_repeat_rows(const float *src, int c, int64_t n, int rep, float *dst, sgnn_stream_t stream);
/* dst[r] = sum_t src[r*rep + t]  (its gradient) */
int sgnn_sum_groups(const float *src, int c, int64_t n, int rep, float *dst, sgnn_stream_t stream);
/* dst[r] = [ a[ia ? ia[r] : r] | (ib ? (ib[r] >= 0 ? b[ib[r]] : 0) : b[r]) ]
 * serves JoinTable (ia = ib = NULL), concat_skip (ib = hash_lookup rows, torch/model.py:354)
 * and the fused mask-compaction concat (ia = ib = sel, torch/model.py:242,330) */
int sgnn_concat_rows(const float *a, int ca, const int32_t *ia, const float *b, int cb,
                     const int32_t *ib, int64_t m, float *dst, sgnn_stream_t stream);
/* gradient of sgnn_concat_rows: da (na rows) / db (nb rows) are zero-filled here when the
 * matching index is given; indices are unique so no atomics are needed.  da or db may be NULL. */
int sgnn_concat_rows_bwd(const float *ddst, int ca, const int32_t *ia, int cb, const int32_t *ib,
                         int64_t m, float *da, int64_t na, float *db, int64_t nb,
                         sgnn_stream_t stream);
/* y = a + b  (scn.AddTable) */
int sgnn_add(const float *a, const float *b, int64_t count, float *y, sgnn_stream_t stream);

/* ---------------------------------------------------------------------------
 * Generative glue
 * ------------------------------------------------------------------------- */
/* children coords: out[(8i+j)] = {2z+dz, 2y+dy, 2x+dx, b}, j = 4dz+2dy+dx
 * (Refinement.to_next_level_locs, torch/model.py:192-207) */
int sgnn_expand8_coords(const int32_t *coords, int64_t n, int32_t *out, const int64_t *n_dev, sgnn_stream_t stream);
/* the same, also writing the children as the int64 (z, y, x, b) rows the model returns per level (torch/model.py:207,243;
 * = sgnn_coords_to_i64 of `out`) in the same pass */
int sgnn_expand8_coords_i64(const int32_t *coords, int64_t n, int32_t *out, int64_t *locs, const int64_t *n_dev,
                            sgnn_stream_t stream);
/* all voxel coordinates of a dense (B, d0, d1, d2) volume, batch-major raster order
 * (GenModel.dense_coarse_to_sparse, torch/model.py:319-321) */
int sgnn_dense_coords(int batch, int d0, int d1, int d2, int32_t *out, sgnn_stream_t stream);
/* stable compaction: sel[0..count) = ascending rows i with sigmoid(logits[i*stride]) > 0.5
 * (torch/model.py:233,322); wave ballot + prefix sum.  *count is a device int64. */
int64_t sgnn_compact_ws_bytes(int64_t n);
int sgnn_compact_sigmoid(const float *logits, int64_t stride, int64_t n, int32_t *sel,
                         int64_t *count, void *ws, int64_t ws_bytes, sgnn_stream_t stream);
/* same for an explicit uint8 mask */
/* teacher forcing: keep site i iff the dense (B,1,d0,d1,d2) float volume is > 0.5 at coords[i] = {z,y,x,b} */
int sgnn_compact_dense(const int32_t *coords, int64_t n, const float *vol, int batch, int d0, int d1, int d2,
                       int32_t *sel, int64_t *count, void *ws, int64_t ws_bytes, sgnn_stream_t stream);
int sgnn_compact_mask(const uint8_t *mask, int64_t n, int32_t *sel, int64_t *count, void *ws,
                      int64_t ws_bytes, sgnn_stream_t stream);
/* capacity mode of the two generative mask compactions: the number of candidates is *n_dev (NULL: n), the kept count
 * count2[0] is clamped to keep_cap (SGNN_STATUS_OVERFLOW in *status otherwise) and count2[1] = 8 * count2[0], the row
 * count of the kept sites' 8-child expansion (torch/model.py:192-207), is published with it */
int sgnn_compact_sigmoid_cap(const float *logits, int64_t stride, int64_t n, const int64_t *n_dev, int32_t *sel,
                             int64_t *count2, int64_t keep_cap, int32_t *status, void *ws, int64_t ws_bytes,
                             sgnn_stream_t stream);
int sgnn_compact_dense_cap(const int32_t *coords, int64_t n, const int64_t *n_dev, const float *vol, int batch, int d0,
                           int d1, int d2, int32_t *sel, int64_t *count2, int64_t keep_cap, int32_t *status, void *ws,
                           int64_t ws_bytes, sgnn_stream_t stream);
/* the same two, also writing locs[r] = coords[sel[r]] (16-byte {z,y,x,b} rows) for the kept rows r < keep_cap: the next
 * level's site list (torch/model.py:236-243) without a gather launch of its own */
int sgnn_compact_sigmoid_cap_locs(const float *logits, int64_t stride, int64_t n, const int64_t *n_dev,
                                  const int32_t *coords, int32_t *sel, int32_t *locs, int64_t *count2, int64_t keep_cap,
                                  int32_t *status, void *ws, int64_t ws_bytes, sgnn_stream_t stream);
int sgnn_compact_dense_cap_locs(const int32_t *coords, int64_t n, const int64_t *n_dev, const float *vol, int batch, int d0,
                                int d1, int d2, int32_t *sel, int32_t *locs, int64_t *count2, int64_t keep_cap,
                                int32_t *status, void *ws, int64_t ws_bytes, sgnn_stream_t stream);

/* scn.SparseToDense (torch/model.py:47): dense (B, C, d0, d1, d2) zero-filled here */
int sgnn_sparse_to_dense(const float *feats, const int32_t *coords, int64_t n, int c, float *dense,
                         int batch, int d0, int d1, int d2, sgnn_stream_t stream);
/* feats[r][ch] = dense[b][ch][z][y][x] at coords[r]  (its gradient, and the NCDHW -> rows
 * permute of dense_coarse_to_sparse, torch/model.py:324-327) */
int sgnn_dense_to_sparse(const float *dense, const int32_t *coords, int64_t n, int c, float *feats,
                         int batch, int d0, int d1, int d2, sgnn_stream_t stream);

/* ---------------------------------------------------------------------------
 * Per-site linear heads y[r] = W x[r] + b, W (cout, cin) row-major, cout <= 2
 * (nn.Linear(nf,1) x2 fused, torch/model.py:190-191,230-231; SurfacePrediction.linear :258,271).
 * bwd: dx (may be NULL), dw (cout, cin), dbias (may be NULL); deterministic reduction via ws.
 * ------------------------------------------------------------------------- */
int64_t sgnn_linear_ws_bytes(int64_t n, int cin, int cout);
int sgnn_linear_fwd(const float *x, int64_t n, int cin, const float *w, const float *bias, int cout,
                    float *y, sgnn_stream_t stream);
int sgnn_linear_bwd(const float *x, const float *dy, int64_t n, int cin, const float *w, int cout,
                    float *dx, float *dw, float *dbias, void *ws, int64_t ws_bytes,
                    sgnn_stream_t stream);

/* ---------------------------------------------------------------------------
 * Fused loss of one hierarchy level: sparse predictions vals (m, vstride) at
 * locs (m,4) int64 [z,y,x,b] against dense (B,1,d0,d1,d2) targets —
 *   out2[0] = mean over kept sites of w * BCE_with_logits(vals[:,occ_col], tgt_occ)   (torch/loss.py:58-82)
 *   out2[1] = mean over kept sites of w * |logt(vals[:,sdf_col]) - logt(tgt_sdf)|      (torch/loss.py:122-157)
 * occ_col / sdf_col = -1 skips that term; weights / known may be NULL.
 * mask_mode 0: keep every site (UNK_ID occupancy targets count as 0); 1: keep tgt_occ != -1;
 * 2: keep known < 2.  sums (3 doubles, device) carries {sum bce, sum l1, kept} to the backward call,
 * which writes dvals (m, vstride) given gout2 = d loss / d out2 (device, 2 floats).
 * ------------------------------------------------------------------------- */
int64_t sgnn_loss_ws_bytes(void);
int sgnn_loss_level_fwd(const int64_t *locs, const float *vals, int vstride, int occ_col, int sdf_col,
                        const float *tgt_occ, const float *tgt_sdf, const float *weights,
                        const uint8_t *known, int d0, int d1, int d2, int64_t m, int use_log,
                        int mask_mode, const int64_t *m_dev, double *sums, float *out2, void *ws, int64_t ws_bytes,
                        sgnn_stream_t stream);
int sgnn_loss_level_bwd(const int64_t *locs, const float *vals, int vstride, int occ_col, int sdf_col,
                        const float *tgt_occ, const float *tgt_sdf, const float *weights,
                        const uint8_t *known, int d0, int d1, int d2, int64_t m, int use_log,
                        int mask_mode, const int64_t *m_dev, const double *sums, const float *gout2, float *dvals,
                        sgnn_stream_t stream);

/* Targets of the hierarchical loss in three launches — compute_targets + compute_weights_missing_geo
 * (torch/loss.py:15-32, 35-48): tsdf = clamp(sdf, +-trunc); hier_last = tsdf; occ_last = |tsdf| < trunc with UNK_ID (-1)
 * where known >= 2 (masking); w_last = weight_missing_geo off the input sites, 1 on them; and for the ncoarse (<= 3)
 * coarser levels, listed from the second finest downwards in the host pointer arrays hier_in / occ / w / hier:
 * occ = 2x2x2 max-pool of the level above, w = that level's w[::2,::2,::2], hier = clamp(hier_in).  w_last NULL: no
 * weights.  input_locs = the (n_locs, 4) int64 [z,y,x,b] input sites.  Dimensions must be divisible by 2^ncoarse. */
int sgnn_loss_targets(const float *sdf, const uint8_t *known, const int64_t *input_locs, int64_t n_locs,
                      const int64_t *n_locs_dev, int batch,
                      int d0, int d1, int d2, float trunc, int masking, float weight_missing_geo, int ncoarse,
                      void *const *hier_in, float *tsdf, float *hier_last, float *occ_last, float *w_last,
                      void *const *occ, void *const *w, void *const *hier, sgnn_stream_t stream);
/* total = sum_i coef[i] * out2s[i] over the (bce, l1) pairs sgnn_loss_level_fwd produced for all levels (n <= 10
 * floats; coef is a HOST array, entries of unused slots 0), cur[l] = out2s[2l] + out2s[2l+1] restricted to used slots;
 * and the gradient fan-out g2[i] = g[0] * coef[i] that sgnn_loss_level_bwd consumes (torch/loss.py:160-199). */
int sgnn_loss_combine(const float *out2s, const float *coef_host, int n, float *total, float *cur,
                      sgnn_stream_t stream);
int sgnn_loss_combine_bwd(const float *g, const float *coef_host, int n, float *g2, sgnn_stream_t stream);
/* The same for ALL levels at once (n <= 5): two launches forward (sgnn_loss_level_fwd x n + sgnn_loss_combine), one
 * backward (sgnn_loss_combine_bwd + sgnn_loss_level_bwd x n); per level bit-identical to the entry points above.
 * levels: HOST array of n x 16 int64 = {locs, vals, vstride, occ_col, sdf_col, tgt_occ, tgt_sdf, weights, known, d0, d1,
 * d2, m, use_log, mask_mode, m_dev} (device pointers as integers); coef_host: 2n floats = (bce, l1) weight per level;
 * sums (3n doubles), out2s (2n floats), total, cur (n floats): device; g: device float = d loss / d total; dvals: HOST
 * array of n device pointers. */
int64_t sgnn_loss_multi_ws_bytes(void);
int sgnn_loss_levels_fwd(const int64_t *levels, int n, const float *coef_host, double *sums, float *out2s, float *total,
                         float *cur, void *ws, int64_t ws_bytes, sgnn_stream_t stream);
int sgnn_loss_levels_bwd(const int64_t *levels, int n, const float *coef_host, const double *sums, const float *g,
                         void *const *dvals, sgnn_stream_t stream);

/* ---------------------------------------------------------------------------
 * Sparse-network programs: a static sub-network (what the reference composes from scn.Sequential /
 * ConcatTable / AddTable / JoinTable containers, torch/model.py:31-47, 178-188, 253-257) compiled into a
 * flat op list and run forward / backward from ONE call — same kernels, same order, bit-identical to the
 * per-layer entry points, no host round trip per layer.  All descriptor arrays are HOST memory:
 *   ops   int32[nops][8] = {type, in0, in1, out, param, level, cin, cout}
 *         type 0 subm conv, 1 stride-2 conv (level -> level+1), 2 unpool (out on `level`, in0 on level+1),
 *         3 batch-norm (param .. param+3 = gamma, beta, running_mean, running_var), 4 add, 5 join
 *         (cin / cout = channels of in0 / in1)
 *   opf   float[nops][4] = {eps, momentum, leak, 0}
 *   bufs  int32[nbuf][2] = {level, channels}; buffer 0 is the input
 *   ops are 12 ints each: type, in0, in1, out, param slot, level, cin, cout, in2, ia, ib, ic.  Types 0..5 are the scn
 *         containers' leaves (SUBM conv, stride-2 conv, UnPooling, BatchNorm(ReLU), AddTable, JoinTable); 6 =
 *         CONCAT_IN out = [in0[ia] | in1[ib] | in2[ic]] (the skip join + feature hand-over between generative
 *         stages, torch/model.py:242, 330, 338-355; ia/ib/ic index the idx[] array of int32 device arrays, -1 = rows
 *         as they are); 7 = the 8-child up-sampling convolution on the parent rulebook (out has 8x the rows); 8 =
 *         linear heads (weight row o = slot par+2o, bias par+2o+1).  Buffers [0, n_ext) are caller-owned inputs
 *         (ext[] pointers; their gradients go to gext[]), the rest live in the arena.
 *   lev_* per level: rows, table ld, nbr table, and for the transition level -> level+1 the children /
 *         ptable tables and the parent array (device pointers, NULL where unused); a JoinTable's inputs are column ranges of its output (no concat pass) unless `keep`
 *         asks for one of them; lev_cnt (the array may be NULL): per rows class a device int64 with the live row
 *         count (capacity mode, see the conventions above; lev_n then holds the capacities) or NULL
 *   params / pgrads: host arrays of device pointers.
 * Feature buffers live in a caller-owned arena (sgnn_prog_arena_floats floats; layout via
 * sgnn_prog_buffer_offset); `input` optionally points buffer 0 outside the arena.  Backward: garena has the
 * same layout, the caller fills the output gradients and flags them in ginit[nbuf].
 * keep[nbuf] (host, may be NULL) flags the buffers the caller reads after the forward call (outputs / taps): the
 * executor fuses conv -> AddTable and conv -> BatchNorm statistics into the convolution epilogue and then never
 * materialises the convolution's own output buffer unless it is flagged; a JoinTable whose inputs can be produced in
 * place gets no copy (its inputs live in column ranges of the join buffer and are read / written through a row
 * stride).  The same `keep` must be passed to the arena / offset queries and to the backward call (the layout depends
 * on it).  sgnn_tune.prog_fusion = 0 switches the fusions off (A/B measurements, parity tests).
 * ------------------------------------------------------------------------- */
/* mode 0: floats of the gradient arena sgnn_prog_backward needs (buffers + per-op areas + backward scratch);
 * mode 1: floats of the arena sgnn_prog_forward needs (buffers + per-op areas);
 * mode 2: the same for an INFERENCE call (sgnn_prog_forward with training = 2): no backward pass may follow, so a
 *         buffer's storage is reused once its last reader has run — the arena is the high-water mark of the live set,
 *         3-4x smaller for a U-Net stage (whole-scene inference, BASELINE configs[3]). */
int64_t sgnn_prog_arena_floats(const int32_t *ops, int nops, const int32_t *bufs, int nbuf, int n_ext,
                               const int64_t *lev_n, int nlev, const int32_t *keep, int mode);
int64_t sgnn_prog_ws_bytes(const int32_t *ops, int nops, const int64_t *lev_n, int nlev);
int64_t sgnn_prog_buffer_offset(const int32_t *ops, int nops, const int32_t *bufs, int nbuf, int n_ext,
                                const int64_t *lev_n, int nlev, const int32_t *keep, int infer, int b);
/* training: 0 = eval, 1 = training (batch statistics), | 2 = inference layout (see sgnn_prog_arena_floats mode 2).
 * wait_event (hipEvent_t or NULL): the stream waits for it right before the program's first Convolution(2,2) — the
 * first operation that touches the stride-2 tables and the coarser levels' hash / rulebook / row counts, which the caller
 * may have built on another stream (they then overlap the level-0 operations in front of it). */
int sgnn_prog_forward(const int32_t *ops, const float *opf, int nops, const int32_t *bufs, int nbuf, int n_ext,
                      const int64_t *lev_n, const int64_t *lev_ld, void *const *lev_nbr,
                      void *const *lev_children, void *const *lev_ptable, void *const *lev_parent,
                      void *const *lev_cnt, int nlev, void *const *params, int nparams,
                      void *const *ext, void *const *idx, int nidx,
                      float *arena, int64_t arena_floats, const int32_t *keep, int training, void *wait_event,
                      void *ws, int64_t ws_bytes, sgnn_stream_t stream);
int sgnn_prog_backward(const int32_t *ops, const float *opf, int nops, const int32_t *bufs, int nbuf, int n_ext,
                       const int64_t *lev_n, const int64_t *lev_ld, void *const *lev_nbr,
                       void *const *lev_children, void *const *lev_ptable, void *const *lev_parent,
                       void *const *lev_cnt, int nlev, void *const *params, void *const *pgrads,
                       int nparams, void *const *ext,
                       void *const *gext, void *const *idx, int nidx, const float *arena, float *garena,
                       int64_t arena_floats, void *const *gout, const int32_t *keep, int training, void *ws,
                       int64_t ws_bytes, sgnn_stream_t stream);

/* Optional second lane for sgnn_prog_backward: every weight-gradient launch (dW + its reduce) runs on `stream2`
 * with workspace `ws2`, concurrently with the dX / BatchNorm chain on the caller's stream (both only read dy); the
 * call joins the lane before it returns control of the parameter gradients.  stream2 == NULL turns it off.
 * Process-wide setting; ws2 must be private to the lane and >= the largest sgnn_conv_bwd_weight_ws_bytes of the
 * program (smaller: the lane is silently not used). */
int sgnn_prog_set_side_stream(sgnn_stream_t stream2, void *ws2, int64_t ws2_bytes);
/* 1: sgnn_prog_backward no longer makes its stream wait for the weight-gradient lane at its end; the CALLER joins the
 * lane's stream before anything reads parameter gradients (train.GraphStep: once, before Adam).  A program's last weight
 * gradient — its first, widest convolution — otherwise holds up the next program's dependent chain for as long as it runs.
 * Returns the previous setting. */
int sgnn_prog_defer_join(int on);
/* ---------------------------------------------------------------------------
 * Optimizer step (torch/train.py:81 optim.Adam, :264 optimizer.step()) as one launch over flat buffers of all n
 * parameters.  seg: HOST array of nseg (<= 8) x 5 int64 = {begin, end, cnt, flag, step}: elements [begin, end) are
 * updated iff the segment was reached this step — *flag > 0 (device float, if flag != 0), else *cnt > 0 (device
 * int64 row count, if cnt != 0), else always — which is torch.optim.Adam skipping parameters whose grad is None (a
 * generative stage without input sites, torch/model.py:211,260) decided on the device; step: device float counter
 * of the segment's updates (bias correction), incremented here.  lr_dev: device float.  grads are multiplied by
 * grad_scale (1 / world size after a sum all-reduce).  Nothing is updated while *status has SGNN_STATUS_OVERFLOW.
 * sgnn_seg_flags writes flags[t] = (*cnt_ptrs[t] > 0) for a data-parallel step's all-reduce (cnt_ptrs: HOST array of
 * device addresses, 0 = always 1) and, with status != NULL, flags[7] = this rank's overflow bit.
 * ------------------------------------------------------------------------- */
int sgnn_adam_flat(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t n, const int64_t *seg,
                   int nseg, const float *lr_dev, float beta1, float beta2, float eps, float weight_decay,
                   float grad_scale, const int32_t *status, sgnn_stream_t stream);
int sgnn_seg_flags(const int64_t *cnt_ptrs, int nseg, float *flags, const int32_t *status, sgnn_stream_t stream);
/* data parallel: flags[7] (written by sgnn_seg_flags from *status, summed by the all-reduce) > 0 -> raise
 * SGNN_STATUS_OVERFLOW locally, so that every replica discards the same step */
int sgnn_status_merge(const float *overflow_flag, int32_t *status, sgnn_stream_t stream);

/* ---------------------------------------------------------------------------
 * On-disk formats feeding the path: .sdfs training chunks, .sdf scenes, .knw masks.
 * Replaces torch/data_util.py:63-117 (load_train_file), :121-139 (load_scene), :142-155
 * (load_scene_known) and the mask + collate of torch/scene_dataloader.py:101-105, :13-36.
 *
 * sgnn_io_layout is host-only (no GPU needed): it validates one file image and returns the byte offset
 * and entry count of every section; nothing is copied.  kind: 0 .sdfs, 1 .sdf, 2 .knw.
 * out[24] (int64, -1 = section absent):
 *   [0..2] dimx, dimy, dimz   [3] voxelsize (f32 bit pattern)   [4] offset of world2grid (16 x f32)
 *   [5] n_input  [6] offset of its (x,y,z) u32 triples  [7] offset of its f32 values      (.sdf: the scene)
 *   [8] n_target [9] [10] likewise                                                         (.sdfs only)
 *   [11] offset of the u8 known volume (dimz*dimy*dimx bytes, z-major)                     (.sdfs, .knw)
 *   [12+3h .. 14+3h] count / triple offset / value offset of hierarchy level h, factor 2^(h+1) (.sdfs only)
 *   [21] bytes consumed
 * The device entry points work on a PACKED batch: the sections of all samples concatenated, with
 * seg[s] .. seg[s+1] delimiting sample s (device int64[nb+1]) and voxelsize[s] its voxel size.
 * ------------------------------------------------------------------------- */
int sgnn_io_layout(const void *bytes, int64_t nbytes, int kind, int64_t *out);
/* mask[i] = |value_i / voxelsize| < truncation && z_i < max_z   (scene_dataloader.py:83-86, :101) */
int sgnn_io_flag_entries(const uint32_t *locs_xyz, const float *vals, const float *voxelsize,
                         const int64_t *seg, int nb, int64_t n, float truncation, int64_t max_z,
                         uint8_t *mask, sgnn_stream_t stream);
/* rows j < *count: out_locs[j] = {z,y,x,sample} (int64), out_feats[j] = value/voxelsize of entry sel[j]
 * (sel from sgnn_compact_mask: ascending, so file order is kept as collate keeps it) */
int sgnn_io_emit_entries(const uint32_t *locs_xyz, const float *vals, const float *voxelsize,
                         const int64_t *seg, int nb, const int32_t *sel, const int64_t *count,
                         int64_t n_max, int64_t *out_locs, float *out_feats, sgnn_stream_t stream);
/* dense[s][z][y][x] = value/voxelsize (data_util.py:44-56 sparse_to_dense_np); `dense` is (nb,d0,d1,d2)
 * pre-filled by the caller (-inf); entries outside the volume or with z >= max_z are skipped */
int sgnn_io_scatter_dense(const uint32_t *locs_xyz, const float *vals, const float *voxelsize,
                          const int64_t *seg, int nb, int64_t n, int d0, int d1, int d2, int64_t max_z,
                          float *dense, sgnn_stream_t stream);

/* ---------------------------------------------------------------------------
 * Evaluation metrics on the device.
 * ------------------------------------------------------------------------- */
/* IoU ingredients of one hierarchy level (torch/loss.py:84-120 compute_iou_sparse_dense, fed as in
 * torch/train.py:271-290).  Rows r of locs (m,4 int64 z,y,x,b) count when keep[r] != 0, or — keep == NULL —
 * when sigmoid(logits[r*lstride]) > 0.5, or always when both are NULL.  tgt: dense (nb,1,d0,d1,d2) occupancy,
 * float {0,1,-1 = unknown} or uint8 {0,1,255} (the reference's `.byte()` of -1).  counters (device int64
 * [nb][3]) receive per sample {P = counted rows (minus rows on unknown voxels if use_mask), C = those whose
 * target is 1, T = voxels with target 1}: intersection = C, union = P + T - C. */
int sgnn_iou_counts(const int64_t *locs, const uint8_t *keep, const float *logits, int64_t lstride, int64_t m,
                    const void *tgt, int tgt_is_u8, int nb, int d0, int d1, int d2, int use_mask,
                    int64_t *counters, sgnn_stream_t stream);
/* mean |pred - target| over target-surface voxels (torch/loss.py:201-231 compute_l1_tgtsurf_sparse_dense):
 * surface = |t| < truncation (thresh < 0) or |t| <= thresh; voxels without a prediction count as -truncation;
 * known != NULL drops voxels with known >= 2.  out3 (device double[3]) = {sum, count, mean}. */
int64_t sgnn_l1_tgtsurf_ws_bytes(void);
int sgnn_l1_tgtsurf(const int64_t *locs, const float *vals, int64_t m, const float *tgt_sdf, const uint8_t *known,
                    int nb, int d0, int d1, int d2, float truncation, float thresh, double *out3, void *ws,
                    int64_t ws_bytes, sgnn_stream_t stream);

/* ---------------------------------------------------------------------------
 * Marching cubes + mesh clean-up on the device.  Replaces
 * torch/marching_cubes/marching_cubes.cpp: run_marching_cubes_internal (:458-476), merge_close_vertices with
 * approx = true (:359-456), remove_degenerate_faces (:298-321), remove_duplicate_faces (:266-297) — same vertex
 * order, same indices, same bits.  The caller chains the stages and reads the counts back in between:
 *   sgnn_mc_count  -> *ntri (device)          triangles the volume produces; keeps per-voxel counts in ws
 *   sgnn_mc_emit   -> verts (3*ntri,3) f32 x,y,z and vcols (3*ntri,3) u8, in the reference's z,y,x voxel order
 *   sgnn_weld_build / sgnn_weld_sweep (repeat until *undecided == 0) / sgnn_weld_lookup
 *                  -> creator_of[i] = the soup vertex that vertex i is merged into, is_creator[i]
 *   sgnn_compact_mask(is_creator) -> sel;  sgnn_weld_number(sel) -> newid;  sgnn_take_rows3 -> welded verts / colours
 *   sgnn_mesh_faces -> faces (ntri,3) remapped + keep[t] (not degenerate, first of its vertex triple);
 *   sgnn_compact_mask(keep) + sgnn_take_rows3 -> final faces.
 * tsdf: dense (d0,d1,d2) = (z,y,x) f32, -inf or |d| >= truncation = no data; colors: (d0,d1,d2,3) u8 or NULL (220).
 * ------------------------------------------------------------------------- */
int64_t sgnn_mc_ws_bytes(int d0, int d1, int d2);
int sgnn_mc_count(const float *tsdf, int d0, int d1, int d2, float isovalue, float truncation, float thresh,
                  void *ws, int64_t ws_bytes, int64_t *ntri, sgnn_stream_t stream);
int sgnn_mc_emit(const float *tsdf, const uint8_t *colors, int d0, int d1, int d2, float isovalue,
                 float truncation, float thresh, void *ws, int64_t ws_bytes, float *verts, uint8_t *vcols,
                 sgnn_stream_t stream);
/* hash slots needed for n keys (vertices for the weld, triangles for the duplicate-face set) */
int64_t sgnn_weld_slots(int64_t n);
/* cells (nv,3) i32 = grid cell of every vertex at pitch `thresh` (the reference passes 1e-5);
 * rep/first (cap) i32, state (cap) u8: the cell table, initialised here */
int sgnn_weld_build(const float *verts, int64_t nv, float thresh, int32_t *cells, int32_t *rep, int32_t *first,
                    uint8_t *state, int64_t cap, sgnn_stream_t stream);
/* one sweep of the creation fixed point; *undecided (device) = cells still open after it */
int sgnn_weld_sweep(const int32_t *cells, const int32_t *rep, const int32_t *first, uint8_t *state, int64_t cap,
                    int64_t *undecided, sgnn_stream_t stream);
int sgnn_weld_lookup(const int32_t *cells, int64_t nv, const int32_t *rep, const int32_t *first,
                     const uint8_t *state, int64_t cap, int32_t *creator_of, uint8_t *is_creator,
                     sgnn_stream_t stream);
/* newid[sel[p]] = p */
int sgnn_weld_number(const int32_t *sel, int64_t n, int32_t *newid, sgnn_stream_t stream);
int sgnn_mesh_faces(const int32_t *creator_of, const int32_t *newid, int64_t ntri, int32_t *faces, int32_t *frep,
                    int32_t *ffirst, int64_t cap, uint8_t *keep, sgnn_stream_t stream);
/* dst row p = src row sel[p] for (.,3) arrays of 1- or 4-byte elements */
int sgnn_take_rows3(const void *src, int elem_bytes, const int32_t *sel, int64_t n, void *dst, sgnn_stream_t stream);

/* ---------------------------------------------------------------------------
 * Optional live timing of the convolution launches (bench.py's roofline leg): HIP events are
 * recorded on the caller's stream around every sgnn_conv_fwd (kind 0) / sgnn_conv_bwd_weight
 * main kernel (kind 1).  Off by default.  sgnn_prof_get must follow a stream synchronise.
 * ------------------------------------------------------------------------- */
int sgnn_prof_enable(int max_records);
int sgnn_prof_disable(void);
int sgnn_prof_resume(void); /* keep the records gathered so far */
int sgnn_prof_count(void);
int sgnn_prof_dropped(void);
/* kernel launches issued by this library so far in this process (difference around a step = its launches) */
int64_t sgnn_launch_count(void);
int sgnn_prof_get(int i, int *kind, int64_t *n_out, int *cin, int *cout, int *K, int *flags, float *ms);

/* Device time stamps inside a captured graph (measurement only; scripts/lane_stamps.py).  sgnn_stamp launches a one-thread
 * kernel that writes the device's constant 100 MHz clock into the next slot of `buf` and remembers `label` for it; while no
 * buffer is enabled it launches nothing and returns -1.  Slots are handed out in call order since the last
 * sgnn_stamp_reset / _enable, so a captured step re-writes the same slots on every replay.  buf: max_stamps int64 on
 * the device (NULL, 0 switches stamps off).  SGNN_STAMP_ONLY=label,label,... in the environment keeps only those labels
 * (a stamp is a graph node, and WHERE a node sits decides which stream the HIP graph executor gives its successors:
 * profiles/r06y_graph_executor.txt). */
int sgnn_stamp_enable(int64_t *buf, int max_stamps);
int sgnn_stamp_reset(void);
int sgnn_stamp(const char *label, sgnn_stream_t stream);
int sgnn_stamp_count(void);
const char *sgnn_stamp_label(int i);

#ifdef __cplusplus
}
#endif
#endif /* SGNN_HIP_H */
