/*
 * chromegcn.h -- C ABI of libchromegcn_hip.so: the MI355X (gfx950) implementation of
 * ChromeGCN's per-chromosome gated graph-convolution hot path.
 *
 * The reference (QData/ChromeGCN) has no FFI of its own: its boundary for this path is
 * the Python nn.Module surface.  Each entry point below names the
 * reference code it replaces (file:line relative to the reference repository root); the
 * Python host side in chromegcn_amd/ binds them with ctypes (INTEGRATION.md shows the
 * stub a reference maintainer would add).
 *
 * Conventions
 *   - Every pointer is a DEVICE pointer unless marked "host".  No torch types.
 *   - All work is enqueued on `stream` (a hipStream_t passed as void*; cgcn_layer_bwd may also use
 *     a caller-provided auxiliary stream); nothing here synchronises the host, allocates or frees
 *     device memory, or keeps a caller pointer after return.
 *     All entry points are therefore legal inside HIP-graph stream capture.
 *   - Return value: CGCN_OK (0) or a negative CGCN_ERR_* code; cgcn_strerror() names it.
 *   - Dense matrices are row-major fp32.  Node features are laid out [S, n, d]:
 *     S strands (S = 1: one call of ChromeGCN.forward; S = 2: the forward and the
 *     reverse-complement strand of finetune.py:41-42 sharing one pass over the graph),
 *     n nodes (windows of one chromosome), d features.  d must be 128 or 256.
 *   - A graph is a CSR triple: rowptr[n+1], col[nnz] (int32, columns sorted or not),
 *     val[nnz] fp32 or NULL meaning "all ones", plus an optional per-row scale
 *     row_scale[n] (NULL = 1).  The row-normalised adjacency of process_graph
 *     (utils/util_methods.py:146-180) is A = diag(row_scale) * Ahat with
 *     row_scale = 1/deg; 'hic', 'constant' and 'none' graphs have val == NULL,
 *     'both' graphs carry val in {1,2,3}.
 *   - The backward needs Ahat^T; (rowptr_t, col_t, val_t) is its CSR.  Hi-C graphs are
 *     symmetric (data/7create_graph_new.py:115-116) so callers pass the same arrays.
 *   - aux / aux_t (may be NULL): optional facts about the graph (cgcn_graph_aux below) that change speed, never
 *     results beyond fp32 re-association: a uint16 copy of the column indices and the length of the longest row.
 *   - Dropout (F.dropout of models/ChromeModels.py:42,50) is counter based: a mask bit is a pure
 *     function of rng_state = {seed, step counter} (uint64[2] in DEVICE memory), a stream id and
 *     the element index, so the backward regenerates the forward's mask.  Every kernel of one
 *     train step must see the same counter; cgcn_sgd_step (the last kernel of a step) advances it.
 */
#ifndef CHROMEGCN_H
#define CHROMEGCN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CGCN_OK 0
#define CGCN_ERR_BAD_ARG (-1)     /* null pointer, negative size, misaligned buffer      */
#define CGCN_ERR_UNSUPPORTED (-2) /* d not in {128,256}, S not in {1,2}, size overflow   */
#define CGCN_ERR_LAUNCH (-3)      /* hipGetLastError() != hipSuccess after a launch      */
#define CGCN_ERR_WORKSPACE (-4)   /* workspace too small (see cgcn_*_workspace_bytes)    */

#define CGCN_ABI_VERSION 25

typedef void *cgcn_stream_t; /* hipStream_t */

/*
 * Optional per-graph facts (HOST struct; the arrays it points to are device memory).  Pass NULL for none.
 *   col16       : the same column indices as uint16, for graphs with at most 65 536 columns (every chromosome at 1 kb
 *                 windows with peaks), or NULL.  The feature-sliced aggregation kernels re-read the index list once per
 *                 128-byte column slice; the 16-bit copy halves those bytes.  Used only with implicit values
 *                 (val == NULL); results are bit-identical.
 *   max_row_len : number of stored entries of the longest row (0 = unknown).  The reference keeps the top-K contacts of a
 *                 chromosome (data/7create_graph_new.py:93-104), so a few windows can have thousands of neighbours.  A
 *                 row is aggregated inside one workgroup by the fused forward (one CU's L1: ~65 us for 10 000
 *                 neighbours of 1 KiB); graphs whose longest row exceeds 2 048 entries therefore take the feature-sliced
 *                 route (8 ... 16 workgroups per row) at every table size.
 *   row_order   : a permutation of 0 .. n_rows-1 (int32, device), or NULL = natural order: the order in which the
 *                 feature-sliced kernels deal the rows to their 64-row tiles (position p -> tile p / 64, wave (p % 64) / 8;
 *                 a wave walks its 8 rows side by side until the longest is done).  The engine passes the rows of every
 *                 64-row group sorted by length (a wave's 8 rows are then about equally long, a tile still holds
 *                 neighbouring rows) and the groups heaviest first (tiles with hub rows start the launch).  Results are
 *                 independent of the order up to fp32 re-association of a row's sum (a wave chooses how to walk its rows
 *                 by their lengths).  Not checked: an array that is not a permutation leaves rows unwritten.
 *   band_halfwidth : w > 0 says the CSR is EXACTLY a band: row i holds the columns max(0, i - w) .. min(n - 1, i + w), each
 *                 once (process_graph's 'constant' branch, utils/util_methods.py:137-150: w = 7; the diagonal is the
 *                 added identity), with implicit unit values (val == NULL).  0 = not a band / unknown.  For w = 7 the
 *                 aggregation then runs as a sliding-window stream over the table (k_band_aggregate / k_bwd_band: every
 *                 row read once, no index list) instead of the CSR walk; same bits (same summation order).  Not checked:
 *                 a wrong hint gives the band's sums, not the CSR's.  ABI 21.
 *   bp_rowptr, bp_col, bp_col16, bp_row_order : the "band plus" decomposition of an explicit-value graph whose values are 1
 *                 or 2 with every 2 inside the +-7 band and every position of the band + I present (process_graph's 'both'
 *                 branch on a {0,1} Hi-C matrix, utils/util_methods.py:168-171), or all NULL.  (bp_rowptr, bp_col) is the
 *                 CSR (int32, device) of the UNIT entries that are not the band's own -- entries outside the band, and one
 *                 unit of each value-2 entry inside it -- so that  sum_j w_ij x_j = sum_{bp entries} x_j + sum_{|j-i|<=7} x_j.
 *                 bp_col16 / bp_row_order are what col16 / row_order are for the main CSR.  The feature-sliced kernels then
 *                 walk the bp CSR with implicit values and 16-bit indices and add the band + I half from a window of the
 *                 table staged once per workgroup in LDS ('both' at 'hic' speed); graphs that carry it take the
 *                 feature-sliced route at every table size.  Used only when val != NULL; results equal the merged CSR's
 *                 up to fp32 re-association of a row's sum.  Not checked.  ABI 21.
 * For the backward (aux_t) all of them describe the CSR of Ahat^T.
 */
typedef struct cgcn_graph_aux {
  const uint16_t *col16;
  const int32_t *row_order;
  int32_t max_row_len;
  int32_t band_halfwidth;
  const int32_t *bp_rowptr;
  const int32_t *bp_col;
  const uint16_t *bp_col16;
  const int32_t *bp_row_order;
} cgcn_graph_aux;

/* ABI version of the loaded library (compare with CGCN_ABI_VERSION). */
int cgcn_abi_version(void);

/* Static string for an error code.  Never NULL. */
const char *cgcn_strerror(int code);

/*
 * Y[s,i,:] = row_scale[i] * sum_{k in row i} val[k] * X[s, col[k], :]
 * Unfused sparse aggregation.  Replaces torch.spmm(adj, support), models/SubLayers.py:46.
 * X, Y: [S, n_cols, d] and [S, n_rows, d]; X and Y must not alias.
 * d: any multiple of 4 up to 4096 (GraphConvolution takes arbitrary in/out widths, models/SubLayers.py:8-12);
 * d = 128 / 256 run tuned kernels.  The fused gated layer below needs d in {128, 256}.
 */
int cgcn_spmm(cgcn_stream_t stream, int n_rows, int n_cols, int S, int d,
              const int32_t *rowptr, const int32_t *col, const float *val, const float *row_scale,
              const float *X, float *Y, const cgcn_graph_aux *aux);

/*
 * One gated graph-convolution layer, forward (one fused launch; or two -- a feature-sliced aggregation into H and a
 * row-local launch on it -- when H is wanted and the feature table S*n*d*4 is too large for the L2s; same results):
 *     H  = diag(row_scale) Ahat X          (aggregation)
 *     U  = H W + b                          (models/SubLayers.py:43-50; the reference
 *                                            computes A (X W) + b -- same value up to fp32
 *                                            re-association)
 *     Z  = tanh(U)                          (models/ChromeModels.py:38 / :44)
 *     g  = sigmoid(Z . wg + cg)             (models/ChromeModels.py:39 / :45, nn.Linear(d,1))
 *     Xn = (1 - g) X + g Z                  (models/ChromeModels.py:40 / :46)
 * X, Xn, Z, H: [S,n,d].  gate: [S,n].  W: [d,d] stored in x out (GraphConvolution.weight).
 * b, wg: [d].  cg: [1] (device).  Z and H may be NULL for inference (they are what the
 * backward needs).  Xn must not alias X.
 * dropout_p > 0 additionally applies the inter-layer dropout of models/ChromeModels.py:42 to Xn
 * (Xn <- mask * Xn / (1-p), mask from (rng_state, stream_id)); pass 0 / NULL / 0 for none.
 * H_in (may be NULL): a previously computed H = diag(row_scale) Ahat X for this X and graph (H does not
 * depend on the layer's weights).  When given, the gather is skipped and H_in is streamed instead; H is
 * then not written (pass H = NULL, the saved tensor for the backward is H_in itself).
 * colstats (may be NULL) / colstats_rows: the first stage of the classifier head's BatchNorm batch statistics
 * (models/ChromeModels.py:58-59, nn.BatchNorm1d in training mode) of relu(Xn), taken while the tile is on chip, in the form
 * cgcn_layer_fwd_colstats_plan planned: colstats_rows = the plan's *rows_per_tile, colstats = a buffer of the plan's tile
 * count x [S][d][2] floats.  Hand (colstats, tiles, colstats_rows) to cgcn_head_train as col_stats.
 *   colstats_rows > 0 (RECORDS): per tile of colstats_rows nodes and (strand, column) the mean and the sum of squared
 *     deviations.  colstats_rows must be a multiple of 16 / S; more than 16 / S nodes per record (merged records) are
 *     produced by the two-launch route only and need H or H_in (CGCN_ERR_BAD_ARG otherwise).
 *   colstats_rows = -1 (CGCN_COLSTATS_ROWS_ACCUMULATE): the buffer holds 64-bit fixed-point integer totals (see the plan);
 *     this call zeroes them in its aggregation launch, so it takes the two-launch route: needs H or H_in, n >= 2, an 8-byte
 *     aligned buffer.
 *   colstats_rows = -2 (CGCN_COLSTATS_ROWS_ZERO_ONLY): this call produces NO statistics; its first launch zeroes the totals
 *     in `colstats` for a LATER call of the same step (the previous layer's forward prepares the last layer's buffer) ...
 *   colstats_rows = -3 (CGCN_COLSTATS_ROWS_ACCUMULATE_ZEROED): ... which then accumulates into them on ANY route, the fused
 *     kernel included (tables below the two-launch size keep their one-launch forward).  Hand the buffer to cgcn_head_train
 *     with col_stats_rows = -1 either way.
 * The mode is an argument of the call (ABI v24): nothing about it is read from process state.
 */
int cgcn_layer_fwd(cgcn_stream_t stream, int n, int S, int d,
                   const int32_t *rowptr, const int32_t *col, const float *val, const float *row_scale,
                   const float *X, const float *W, const float *b, const float *wg, const float *cg,
                   float *Xn, float *Z, float *H, float *gate,
                   float dropout_p, const unsigned long long *rng_state, unsigned int stream_id,
                   const float *H_in, float *colstats, int colstats_rows, const cgcn_graph_aux *aux);

/* Column-statistics modes of cgcn_layer_fwd / cgcn_head_train (cgcn_layer_fwd_colstats_plan's `mode`) ... */
#define CGCN_COLSTATS_RECORDS 0
#define CGCN_COLSTATS_ACCUMULATE 1
/* ... and the special values of colstats_rows (above). */
#define CGCN_COLSTATS_ROWS_ACCUMULATE (-1)
#define CGCN_COLSTATS_ROWS_ZERO_ONLY (-2)
#define CGCN_COLSTATS_ROWS_ACCUMULATE_ZEROED (-3)

/* What to allocate for the column statistics of cgcn_layer_fwd(n, S, d) in `mode`: returns the number of tiles ([S][d][2]
 * floats each; 0 = unsupported shape or mode), *rows_per_tile = the value to pass as colstats_rows / col_stats_rows.
 * CGCN_COLSTATS_RECORDS: *rows_per_tile = nodes per record (the last one may be shorter): 16 / S on tables that take the
 *   fused route, a multiple of it (one record per row-local workgroup) on tables that take the two-launch route.
 * CGCN_COLSTATS_ACCUMULATE (n >= 2; falls back to records for n < 2): *rows_per_tile = -1 -- the buffer is used as 64-bit
 *   integer fixed-point totals of sum relu(Xn) and sum relu(Xn)^2 per (strand, column), zeroed and added to inside
 *   cgcn_layer_fwd (order-independent, so bit-reproducible); cgcn_head_train, handed the same (buffer, tiles, -1), derives
 *   the BatchNorm statistics from the totals inside its main kernel and launches no finalize kernel; it also leaves the
 *   BatchNorm-BACKWARD column sums as integer totals in the same buffer (and writes the loss from a ticketed total) instead
 *   of launching a finish kernel: the cgcn_head_grad handed to cgcn_layer_bwd must then carry stat_acc = that buffer.
 *   Range of the fixed point (32 fraction bits): sum relu(Xn)^2 < 2.1e9 per column (rms |Xn| < 265 at n = 30 000); beyond
 *   it the statistics -- and the loss -- come out NaN (loudly wrong, never silently wrapped).  The caller decides per call
 *   from what it knows about its inputs: chromegcn_amd's engine bounds |Xn| by the input features per chromosome
 *   (finetune.GCNStage), its module-level entry points default to records. */
int cgcn_layer_fwd_colstats_plan(int n, int S, int d, int mode, int *rows_per_tile);

/* Test / tuning hook: feature-table size in bytes from which cgcn_layer_fwd takes the two-launch route when H is
 * given (0 = always, negative = restore the built-in default).  Process-wide. */
void cgcn_debug_set_fwd_split_bytes(long long bytes);

/* Measurement hook: how the dense fp32 products of the row-local kernels are formed.  CGCN_PRODUCTS_SPLIT (default): six
 * bf16 MFMA partial products of an EXACT three-way split of every fp32 operand (x = h + m + l, 8 + 8 + 8 significant
 * bits), fp32 accumulators -- fp32 arithmetic on the bf16 matrix cores, measured MORE accurate against float64 than the
 * chain (it rounds 8 times where the chain rounds K / 4 times; profiles/r06_bf16x6_probe.txt).  CGCN_PRODUCTS_FP32_CHAIN:
 * v_mfma_f32_16x16x4_f32, the form of rounds 1-5.  The two differ in the last bits.  Process-wide; initial value from the
 * environment (CGCN_PRODUCTS=fp32 | split); any other argument restores that initial value. */
#define CGCN_PRODUCTS_FP32_CHAIN 0
#define CGCN_PRODUCTS_SPLIT 1
void cgcn_debug_set_products(int mode);
int cgcn_debug_get_products(void);

/* Which kernels a call WOULD launch, so that a profiler prices the kernel that actually runs (bench.py's roofline);
 * nothing is launched, no GPU is needed.
 *   cgcn_debug_layer_fwd_route: the training forward (H given, no H_in) on this graph with column statistics as planned
 *     (colstats_rows as for cgcn_layer_fwd; 0 = none), under the current split threshold: 0 = the fused k_layer_fwd,
 *     1 = k_aggregate_sliced + k_layer_dense / k_layer_dense256 (large tables, hub-heavy graphs, accumulate mode),
 *     2 = k_band_aggregate + the row-local kernel (band graphs: cgcn_graph_aux::band_halfwidth; their backward's last
 *     launch is k_bwd_band instead of k_bwd_sliced).
 *   cgcn_debug_layer_bwd_route: the row-local launch of cgcn_layer_bwd: 0 = k_bwd_rowlocal256s (d = 256: four column-slab
 *     workgroups per range of 32-row tiles, both dense products in the launch; ABI <= 20: k_bwd_rowlocal256 + a second
 *     launch for dHs), 2 = k_bwd_rowlocal_ring (d = 128: row / matrix wave teams over a flag-synchronised LDS ring).
 *     (1 was the 48-row-tile kernel of ABI <= 18: no longer returned.)
 * Negative = error code (unsupported shape). */
int cgcn_debug_layer_fwd_route(int n, int S, int d, const cgcn_graph_aux *aux, int colstats_rows);
int cgcn_debug_layer_bwd_route(int n, int S, int d);

/*
 * State the fused head's backward leaves for the LAST gated layer's backward (cgcn_head_bwd with
 * dX == NULL fills dym / bnc inside its workspace; cgcn_head_workspace_layout gives their offsets).
 * With it, cgcn_layer_bwd recomputes dL/dXn per row (BatchNorm backward, ReLU mask, dropout mask)
 * instead of reading it: one launch and one [S,n,d] round trip less.
 */
typedef struct cgcn_head_grad {
  const float *dym;         /* [n,d]    */
  const float *bnc;         /* [S,2,d]  */
  const float *save_mean;   /* [S,d]    */
  const float *save_invstd; /* [S,d]    */
  const float *bn_w;        /* [d]      */
  float dropout_p;          /* head dropout probability (0 = none) */
  const unsigned long long *rng_state;
  /* in deferred mode cgcn_head_bwd leaves the second stage of dW_out / db_out to cgcn_layer_bwd, which runs it in
   * extra workgroups of its row-local kernel: */
  const float *part;        /* partials region of the head workspace (cgcn_head_workspace_layout) */
  int n_partials;           /* cgcn_head_bwd_partials(n) */
  int C;
  float *dW_out;            /* [C,d] */
  float *db_out;            /* [C]   */
  int accumulate;
  const float *dloss;       /* [1] upstream d loss when the workspace came from cgcn_head_train (whose results are for
                               d loss = 1); NULL when it came from cgcn_head_bwd with dpred (already scaled) */
  float *dbn_w;             /* [d] both set when the workspace came from cgcn_head_train: d(bn weight) / d(bn bias) */
  float *dbn_b;             /* [d] are then finished here as well (same accumulate flag); NULL after cgcn_head_bwd  */
  const void *stat_acc;     /* ABI v23: NULL, or -- when cgcn_head_train ran in accumulate mode (col_stats_rows = -1) -- the
                               SAME col_stats buffer: its backward block holds the BatchNorm-backward column sums as integer
                               totals and `bnc` is not read (cgcn_head_train launched no finish kernel to fill it) */
} cgcn_head_grad;

/*
 * Optional: the optimizer step of utils/util_methods.py:14-19's SGD (optimizer.step(), finetune.py:49) fused into
 * the LAST cgcn_layer_bwd call of a train step -- the first layer's backward, whose gather launch (dX != NULL) or
 * partial-sum launch (dX == NULL) finishes this layer's parameter sums in slab workgroups.  Those workgroups then apply
 * the update to the elements they have just finished, and further extra workgroups update every other element of
 * the flat arenas (their gradients were finished by earlier launches of the step): no cgcn_sgd_step launch.
 * param / grad / momentum_buf: flat fp32 arenas of `count` elements in which every parameter, its gradient and its
 * momentum buffer sit at the SAME offset; dW, db, dwg, dcg of this call must point into `grad`.  Semantics and the
 * rng_state counter advance are those of cgcn_sgd_step.  Needs n > 0, accumulate == 0, aux_stream == NULL, and -- when
 * dX is produced -- in_dropout_p == 0: the step advances the dropout counter inside the same launch that would read
 * it for the input-dropout mask (the first layer's input is never a dropped tensor: models/ChromeModels.py:37-42).
 */
typedef struct cgcn_sgd_fuse {
  float *param;
  const float *grad;
  float *momentum_buf;      /* NULL iff momentum == 0 */
  long long count;
  float lr, momentum, weight_decay, grad_scale;
  int nesterov;
  unsigned long long *rng_state;   /* may be NULL */
} cgcn_sgd_fuse;

/* Bytes of scratch cgcn_layer_bwd needs for (n, S, d).  0 on unsupported shapes. */
size_t cgcn_layer_bwd_workspace_bytes(int n, int S, int d);

/*
 * Backward of cgcn_layer_fwd (what autograd derives for models/ChromeModels.py:37-40 /
 * :43-46; math in SURVEY.md Appendix A).  Given dXn = dL/dXn [S,n,d] and optionally
 * dgate = dL/dgate [S,n] (NULL = 0):
 *     gamma = g (1-g) (sum_k dXn (Z - X) + dgate)
 *     dU    = (g dXn + gamma wg^T) (1 - Z^2)
 *     db = sum_rows dU,  dwg = sum_rows gamma Z,  dcg = sum gamma,  dW = H^T dU
 *     dHs   = diag(row_scale) dU W^T        (dL/dH, pre-scaled)
 *     dX    = (1-g) dXn + Ahat^T dHs
 * dX: [S,n,d], must not alias dXn; NULL = parameter gradients only (the gather over Ahat^T is skipped:
 * use it for the first layer when nobody needs d loss / d features).  dW [d,d], db [d], dwg [d], dcg [1] are overwritten
 * when accumulate == 0 and added to when accumulate != 0.  dHs is a [S,n,d] output (the gather's operand; also
 * the adjacency-saliency operand: dL/dA_ij = <dHs_i, X_j>, see cgcn_sddmm); it may be NULL when dX is NULL (then it
 * is not computed) and must not alias dX.  Sums over rows are two-stage and deterministic
 * (no float atomics): results are bit-reproducible run to run.
 * in_dropout_p > 0: X was produced by a layer that applied dropout (in_stream_id = that layer's
 * stream_id); dX is then the gradient w.r.t. the pre-dropout tensor (mask / (1-p) applied).
 * Exactly one of dXn and head must be non-NULL (head: see cgcn_head_grad).
 * aux_stream (may be NULL): a second stream on which the partial-sum reduction runs concurrently with the
 * gather kernel; forked from and joined back into `stream` with events inside this call.
 * sgd (may be NULL): see cgcn_sgd_fuse.
 */
int cgcn_layer_bwd(cgcn_stream_t stream, int n, int S, int d,
                   const int32_t *rowptr_t, const int32_t *col_t, const float *val_t, const float *row_scale,
                   const float *X, const float *Z, const float *H, const float *gate,
                   const float *W, const float *wg,
                   const float *dXn, const float *dgate,
                   float *dX, float *dHs, float *dW, float *db, float *dwg, float *dcg,
                   int accumulate, float in_dropout_p, const unsigned long long *rng_state,
                   unsigned int in_stream_id, const cgcn_head_grad *head,
                   void *workspace, size_t workspace_bytes, cgcn_stream_t aux_stream,
                   const cgcn_sgd_fuse *sgd, const cgcn_graph_aux *aux_t);

/*
 * Profiling hook: cgcn_layer_bwd one launch group at a time, so that each kernel can be bracketed with events on the
 * caller's stream (bench.py's per-kernel roofline).  phases: bit 0 = the row-local launch (k_bwd_rowlocal_ring / k_bwd_rowlocal256s),
 * bit 1 = the launch after it (k_bwd_sliced / k_bwd_band with the second-stage sums, or k_reduce_partials when dX ==
 * NULL).  phases == 3 is cgcn_layer_bwd without aux_stream / sgd.  A phase-2-only call works on the partials, dHs and
 * (head mode) dL/dXn an earlier phase-1 call left in workspace / dHs / dX.  Stateless like everything else here.
 */
int cgcn_debug_layer_bwd_phases(cgcn_stream_t stream, int n, int S, int d,
                                const int32_t *rowptr_t, const int32_t *col_t, const float *val_t, const float *row_scale,
                                const float *X, const float *Z, const float *H, const float *gate,
                                const float *W, const float *wg, const float *dXn, const float *dgate,
                                float *dX, float *dHs, float *dW, float *db, float *dwg, float *dcg,
                                int accumulate, float in_dropout_p, const unsigned long long *rng_state,
                                unsigned int in_stream_id, const cgcn_head_grad *head,
                                void *workspace, size_t workspace_bytes, int phases, const cgcn_graph_aux *aux_t);

/* Bytes of scratch cgcn_head_fwd / cgcn_head_bwd need for (n, S, d, C).  0 on unsupported shapes. */
size_t cgcn_head_workspace_bytes(int n, int S, int d, int C);

/* Byte offsets, inside that workspace, of the dym [n,d], bnc [S,2,d] and per-workgroup partials regions
 * cgcn_head_bwd fills (host pointers; see cgcn_head_grad), and the number of partial blocks. */
int cgcn_head_workspace_layout(int n, int S, int d, int C, size_t *dym_offset, size_t *bnc_offset,
                               size_t *part_offset);
int cgcn_head_bwd_partials(int n);

/*
 * Classifier head + loss of the GCN-stage step, fused (models/ChromeModels.py:48-51 applied to each
 * strand, then finetune.py:43,45,52):
 *     y_s   = dropout(BatchNorm1d(relu(X_s)))     batch statistics over the n nodes when training
 *     pred  = mean_s (y_s W_out^T + b_out)        computed as (mean_s y_s) W_out^T + b_out
 *     loss  = mean over n*C of BCE-with-logits(pred, target);   probs = sigmoid(pred)
 * X: [S,n,d].  bn_w, bn_b, run_mean, run_var: [d]; num_batches_tracked: int64[1] or NULL.  W_out: [C,d],
 * b_out: [C], target: [n,C] (0/1 floats).  C <= 256 (the reference takes C from the data, main.py:35; the training
 * kernels walk the labels in passes of 128).
 * training != 0: batch statistics, running statistics updated once per strand in strand order (what two
 *   successive ChromeGCN.forward calls do), num_batches_tracked += S, dropout with probability dropout_p
 *   driven by rng_state (see Conventions; the head uses its own fixed stream id).  Outputs
 *   dpred = d loss / d pred [n,C] (may be NULL), save_mean / save_invstd [S,d].
 * training == 0: running statistics, no dropout; dpred, save_*, rng_state unused.
 * probs: [n,C]; loss: [1].
 */
int cgcn_head_fwd(cgcn_stream_t stream, int n, int S, int d, int C, const float *X, const float *bn_w,
                  const float *bn_b, float *run_mean, float *run_var, long long *num_batches_tracked,
                  float momentum, float eps, int training, const float *W_out, const float *b_out,
                  const float *target, float dropout_p, const unsigned long long *rng_state,
                  float *probs, float *loss, float *dpred, float *save_mean, float *save_invstd,
                  void *workspace, size_t workspace_bytes);

/*
 * The eval-mode classifier head of ONE ChromeGCN.forward call per strand (models/ChromeModels.py:48-51 with the module in
 * eval mode, what `model(x, adj)` returns as its second value): logits[s] = BatchNorm1d(relu(X[s])) W_out^T + b_out with
 * the RUNNING statistics, dropout off -- no strand mean, no sigmoid, no loss, nothing updated.  X: [S,n,d] -> logits: [S,n,C].
 */
int cgcn_head_logits(cgcn_stream_t stream, int n, int S, int d, int C, const float *X, const float *bn_w,
                     const float *bn_b, const float *run_mean, const float *run_var, float eps, const float *W_out,
                     const float *b_out, float *logits);

/*
 * Training-mode head forward fused with the tile-local half of its backward (one pass over X): same outputs as
 * cgcn_head_fwd(training = 1) -- probs, loss, save_mean / save_invstd, running-stat update -- and, left in
 * `workspace` for the backward, dym = d loss / d (mean_s y_s) and the per-workgroup partials of dW_out, db_out and
 * the BatchNorm sums, all for an upstream d loss of 1, plus bnc (the BatchNorm-backward column means, likewise for
 * d loss = 1).  The same workspace is then handed to cgcn_layer_bwd via cgcn_head_grad with .dloss, .dbn_w and .dbn_b
 * set: that call finishes every head gradient.  (cgcn_head_bwd with dpred == NULL and dX == NULL is accepted and
 * launches nothing.)  d loss / d pred never touches memory.
 * col_stats (may be NULL): the colstats output of the cgcn_layer_fwd call that produced X, with its tile count and
 * rows per tile (cgcn_layer_fwd_colstats_plan; rows = -1: accumulate mode, the integer totals); the head then skips its own first pass over X.
 */
int cgcn_head_train(cgcn_stream_t stream, int n, int S, int d, int C, const float *X, const float *bn_w,
                    const float *bn_b, float *run_mean, float *run_var, long long *num_batches_tracked,
                    float momentum, float eps, const float *W_out, const float *b_out, const float *target,
                    float dropout_p, const unsigned long long *rng_state, float *probs, float *loss,
                    float *save_mean, float *save_invstd, const float *col_stats, int col_stats_tiles,
                    int col_stats_rows, void *workspace, size_t workspace_bytes);

/*
 * Profiling hook: cgcn_head_train one launch group at a time.  phases: bit 0 = batch statistics (k_head_colstats unless
 * col_stats is given, k_head_bn_finalize -- this one updates the running statistics), bit 1 = k_head_fused (one launch
 * per pass of <= 128 labels), bit 2 = k_head_train_finish.  phases == 7 is cgcn_head_train.
 */
int cgcn_debug_head_train_phases(cgcn_stream_t stream, int n, int S, int d, int C, const float *X, const float *bn_w,
                                 const float *bn_b, float *run_mean, float *run_var, long long *num_batches_tracked,
                                 float momentum, float eps, const float *W_out, const float *b_out, const float *target,
                                 float dropout_p, const unsigned long long *rng_state, float *probs, float *loss,
                                 float *save_mean, float *save_invstd, const float *col_stats, int col_stats_tiles,
                                 int col_stats_rows, void *workspace, size_t workspace_bytes, int phases);

/*
 * Backward of cgcn_head_fwd (training mode).  dloss: [1] upstream gradient of the loss or NULL (= 1).
 * Outputs dX [S,n,d] and, overwritten (accumulate == 0) or added to (accumulate != 0): dW_out [C,d],
 * db_out [C], dbn_w [d], dbn_b [d].  With dX == NULL ("deferred mode") only dbn_w, dbn_b and the dym / bnc /
 * partials state are produced; dX, dW_out and db_out are then finished by cgcn_layer_bwd given a
 * cgcn_head_grad that points at this workspace.  dpred == NULL (with dX == NULL): the workspace comes from
 * cgcn_head_train and already holds dym and the partials.  rng_state: same contents as in the forward.  Deterministic.
 */
int cgcn_head_bwd(cgcn_stream_t stream, int n, int S, int d, int C, const float *X, const float *bn_w,
                  const float *bn_b, const float *save_mean, const float *save_invstd, const float *W_out,
                  const float *dpred, const float *dloss, float dropout_p, const unsigned long long *rng_state,
                  float *dX, float *dW_out, float *db_out, float *dbn_w, float *dbn_b, int accumulate,
                  void *workspace, size_t workspace_bytes);

/*
 * Sampled dense-dense product on the graph's sparsity pattern:
 *     out[k] = sum_s < A[s,i,:], B[s,col[k],:] >     for every stored entry k of row i.
 * With A = dL/dU and B = X W of one layer this is dL/dA_ij on the pattern, i.e. the adjacency saliency
 * of scripts/visualize.py:29-49 (which materialises a dense n x n gradient on the CPU).  A, B: [S,n,d].
 * accumulate != 0: out[k] += the product (the saliency sums one product per layer).
 */
int cgcn_sddmm(cgcn_stream_t stream, int n, int S, int d, const int32_t *rowptr, const int32_t *col,
               const float *A, const float *B, float *out, int accumulate);

/*
 * The row normalisation of that saliency, scripts/visualize.py:49-55, on the pattern and in the reference's order of
 * operations: v[k] = |val[k] * raw[k]| (val == NULL: ones), out[k] = (v[k] / s_i) / m_i with s_i = the sum of row i's v (1 where
 * it is 0) and m_i = the maximum of row i's quotients (1 where it is 0).  raw, out: [nnz]; out may alias raw.
 */
int cgcn_saliency_normalize(cgcn_stream_t stream, int n, const int32_t *rowptr, const float *val, const float *raw, float *out);

/* adj_type codes of the device-side normaliser: the branches of process_graph, utils/util_methods.py:148-174 */
#define CGCN_ADJ_HIC 0
#define CGCN_ADJ_CONSTANT 1
#define CGCN_ADJ_BOTH 2
#define CGCN_ADJ_NONE 3

/*
 * Device-side process_graph (utils/util_methods.py:146-180), step 1 of 2: per-row entry counts of
 * A-hat (row_counts[n]) and their exclusive prefix sum (rowptr_out[n+1]; rowptr_out[n] = nnz(A-hat),
 * which the caller reads back once to size col_out).  Input: canonical CSR (sorted columns, no
 * duplicates) of the raw Hi-C matrix, fp32 values or NULL = ones; ignored for CONSTANT / NONE.
 */
int cgcn_graph_count(cgcn_stream_t stream, int n, int adj_type, const int32_t *rowptr_in, const int32_t *col_in,
                     const float *val_in, int32_t *row_counts, int32_t *rowptr_out);

/*
 * Step 2: columns of A-hat (sorted), values (val_out; required for BOTH, optional/ignored otherwise: HIC,
 * CONSTANT and NONE graphs are all-ones), row_scale = fp32(1 / rowsum) with 1/0 -> 0.
 * symmetric_flag (int32[1], caller sets it to 1; may be NULL): cleared if A-hat != A-hat^T.
 */
int cgcn_graph_fill(cgcn_stream_t stream, int n, int adj_type, const int32_t *rowptr_in, const int32_t *col_in,
                    const float *val_in, const int32_t *rowptr_out, int32_t *col_out, float *val_out,
                    float *row_scale, int32_t *symmetric_flag);

/* Bytes of scratch cgcn_multilabel_metrics needs (sort buffers + rocPRIM temporary storage). */
size_t cgcn_metrics_workspace_bytes(long long n, int C);

/*
 * Per-label ranking metrics of the reference's compute_metrics (utils/evals.py:89-92, utils/metrics.py),
 * computed on the device from probs [n,C] and 0/1 targets [n,C]:
 *   out[0*C + c] AUROC (roc_auc_score; NaN when label c has a single class present)
 *   out[1*C + c] area under sklearn's precision_recall_curve by the trapezoid rule (utils/metrics.py:168-176)
 *   out[2*C + c] recall at the first curve point (from the low-threshold end) with FDR <= fdr_cutoff
 *                (utils/metrics.py:148-160, cutoff 0.5 there)
 *   out[3*C + c] average precision (mean over labels = mean_average_precision, utils/metrics.py:25-26)
 * Tied scores form one curve point, as in sklearn.  n*C < 2^31.
 */
int cgcn_multilabel_metrics(cgcn_stream_t stream, long long n, int C, const float *probs, const float *targets,
                            float fdr_cutoff, float *out, void *workspace, size_t workspace_bytes);

/*
 * The same metrics for NON-NEGATIVE scores -- probabilities, which is what the reference's compute_metrics is given
 * (finetune.py:52 stores F.sigmoid(pred); utils/evals.py:26): with the sign known a (score, target) pair is a 32-bit
 * key and the library sorts every label's list with its own segmented radix sort (4 passes of 8 bits over all labels
 * at once) instead of one device-wide sort of 64-bit keys; same `out`, same workspace size
 * (cgcn_metrics_workspace_bytes), same tie rule, bit-identical results.
 * bad: device int32[1]; set to 1 when a score was negative or NaN -- `out` is then unspecified and the caller repeats
 * the call with cgcn_multilabel_metrics (chromegcn_amd.metrics does; it reads `bad` with the results, no extra sync).
 */
int cgcn_multilabel_metrics_nonneg(cgcn_stream_t stream, long long n, int C, const float *probs, const float *targets,
                                   float fdr_cutoff, float *out, int32_t *bad, void *workspace, size_t workspace_bytes);

/*
 * torch.optim.SGD step on one flat fp32 buffer (utils/util_methods.py:14-19 builds
 * SGD(lr, momentum=0.9, weight_decay=1e-6); dampening 0):
 *     d = grad_scale * grad + weight_decay * param;  buf = momentum * buf + d;
 *     param -= lr * (nesterov ? d + momentum * buf : buf)
 * momentum_buf zero-initialised reproduces torch's first step (buf = d); may be NULL when momentum == 0.
 * grad_scale: 1, or 1/k when grad holds the all-reduced SUM of k ranks' gradients (multi-GPU step group).
 * rng_state (may be NULL): its step counter is advanced by one -- this is the last kernel of a train step.
 */
int cgcn_sgd_step(cgcn_stream_t stream, long long count, float *param, const float *grad, float *momentum_buf,
                  float lr, float momentum, float weight_decay, int nesterov, float grad_scale,
                  unsigned long long *rng_state);

#ifdef __cplusplus
}
#endif
#endif /* CHROMEGCN_H */
