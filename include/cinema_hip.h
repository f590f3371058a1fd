/*
 * cinema_hip.h -- C ABI of libcinema_hip.so: the MI355X (gfx950) kernels behind CINeMA's
 * conditional-INR hot path.
 *
 * The reference (m-dannecker/CINeMA) is pure Python/PyTorch and has NO FFI layer; its
 * boundary for this path is the Python module API of models/inr_decoder.py and
 * models/siren.py.  Each entry point below therefore cites the
 * reference Python code whose arithmetic it replaces.  The Python mirror of that module
 * API (cinema_amd/models/) binds these symbols with ctypes (cinema_amd/_lib.py);
 * INTEGRATION.md shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a raw DEVICE pointer unless marked "host"; sizes are explicit;
 *   - `stream` is a hipStream_t passed as void*; every call is asynchronous on it and
 *     performs no device synchronisation;
 *   - return value: 0 = ok, otherwise a hipError_t or a negative library code;
 *     cinema_last_error() returns a thread-local description of the last failure;
 *   - the caller (PyTorch) owns every buffer; the library allocates nothing.
 *   - host state: a few per-device caches (CU count, kernel LDS attributes, the address of the no-alloc scratch), indexed
 *     by the HIP device that is current at the call, so one process may drive several devices; calls on different streams
 *     of one device may run concurrently unless an entry point says otherwise.  The debug trace selector
 *     (cinema_debug_fused_trace) and the CINEMA_* environment switches are process-wide.
 *   - precision: 0 = fp32 (exact-f32 MFMA), 1 = bf16 (bf16 MFMA, fp32 accumulate).  The
 *     element type "T" of activation / packed-weight buffers follows it.
 *   - Npad = N rounded up to a multiple of 128.  Row buffers have Npad rows; rows >= N are
 *     written as zeros (inputs) or finite don't-cares (activations) by the library.
 */
#ifndef CINEMA_HIP_H
#define CINEMA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CINEMA_MAX_LAYERS 16
#define CINEMA_PREC_F32 0
#define CINEMA_PREC_BF16 1

/* Flags for cinema_siren_backward */
#define CINEMA_BWD_SKIP_WEIGHT_GRADS 1 /* frozen MLP (validation latent fitting, build_atlas.py:99-126) */
/* Phase selection (profiling / roofline timing of one kernel at a time).  No phase bit set = all phases. */
#define CINEMA_BWD_PHASE_CHAIN 4     /* per-layer dX GEMMs with the fused sine-derivative epilogues */
#define CINEMA_BWD_PHASE_DLAT 8      /* d lat GEMM */
#define CINEMA_BWD_PHASE_DW 16       /* grouped weight-gradient GEMMs (one launch) */
#define CINEMA_BWD_PHASE_FINAL_DW 32 /* final_linear weight / bias gradients */
#define CINEMA_BWD_PHASE_MASK 60
/* Data-parallel order: a first call with WEIGHTS_FIRST runs everything but the d-latent GEMM (the flat gradient is
 * complete when it returns, so its all-reduce can start), a second call with DLAT_ONLY runs the d-latent GEMM. */
#define CINEMA_BWD_WEIGHTS_FIRST 256
#define CINEMA_BWD_DLAT_ONLY 512
/* The caller already holds zeros in grad_params / in dcoords (a training loop clears them off the critical path:
 * cinema_adamw_step_clear leaves zeroed gradients behind, cinema_step_prepare zeroes dcoords on a side stream):
 * the call then skips its own hipMemsetAsync of that buffer. */
#define CINEMA_BWD_GRADS_CLEARED 2
#define CINEMA_BWD_DCOORDS_CLEARED 64
/* nodes path: the scale s is rounded to bf16 and saved as on the per-sample path (development / A-B switch) */
#define CINEMA_NODES_ROUND_S 128

int cinema_version(void);
const char *cinema_last_error(void);

/* -------------------------------------------------------------------------------------
 * SIREN stack description.  Mirrors Siren.__init__ (models/siren.py:40-57) and the sizes
 * INR_Decoder.__init__ derives (models/inr_decoder.py:14-22).  The caller fills the first
 * block and calls cinema_siren_plan(), which fills the derived block.
 * ------------------------------------------------------------------------------------- */
typedef struct cinema_siren_desc {
  /* --- caller-filled --- */
  int32_t n_layers;        /* sine layers incl. the first one = num_hidden_layers + 1 */
  int32_t in_dim;          /* coordinate dimension (3) */
  int32_t hidden;          /* hidden_size; must be a multiple of 16 */
  int32_t lat;             /* latent_dim[0] + cond_dims (siren lat_size) */
  int32_t out_dim;         /* sum(out_dim) <= 16 */
  int32_t precision;       /* CINEMA_PREC_* */
  uint32_t modulated_mask; /* bit i set <=> layer i in modulated_layers */
  float omega_first;       /* f_om */
  float omega_hidden;      /* h_om */
  /* --- derived by cinema_siren_plan --- */
  int32_t HP;              /* hidden rounded up to 64 */
  int32_t LP;              /* lat rounded up to 64 */
  int32_t n_saved;         /* number of [Npad x HP] T tensors in the saved-activation buffer */
  int32_t n_gbuf;          /* number of [Npad x HP] T tensors in the backward scratch buffer */
  /* offsets in floats into the flat fp32 parameter / gradient buffer (natural nn.Linear
   * layouts, state_dict order of the reference: final_linear first, siren.py:50-57) */
  int64_t off_w[CINEMA_MAX_LAYERS], off_b[CINEMA_MAX_LAYERS];
  int64_t off_wl[CINEMA_MAX_LAYERS], off_bl[CINEMA_MAX_LAYERS]; /* -1 if not modulated */
  int64_t off_wf, off_bf, n_params;
  /* offsets in elements of T into the packed compute copy of the weights */
  int64_t pk_w[CINEMA_MAX_LAYERS], pk_wt[CINEMA_MAX_LAYERS];
  int64_t pk_wls[CINEMA_MAX_LAYERS], pk_wlt[CINEMA_MAX_LAYERS];
  int64_t pk_wlsT[CINEMA_MAX_LAYERS], pk_wltT[CINEMA_MAX_LAYERS];
  int64_t pk_wf, pk_total;
  /* slot indices ([Npad x HP] tensors) */
  int32_t sv_y[CINEMA_MAX_LAYERS], sv_c[CINEMA_MAX_LAYERS], sv_z[CINEMA_MAX_LAYERS], sv_s[CINEMA_MAX_LAYERS];
  int32_t gb_z[CINEMA_MAX_LAYERS], gb_s[CINEMA_MAX_LAYERS], gb_t[CINEMA_MAX_LAYERS];
  /* bf16, hidden <= 512: per-wave MFMA-fragment weight streams of the activation-stationary kernels (offsets in
   * elements into the packed copy, k32 steps per 64-feature slab); -1 / 0 when that path does not apply */
  int64_t pk_fs, pk_bs, pk_side;   /* pk_side: fp32 block of biases + first-layer weights in the kernel's LDS layout */
  int32_t fs_steps, bs_steps;
  /* bytes of scratch the caller appends behind the n_gbuf tensors of `gbuf` (fp32 partial tiles of the split
   * weight-gradient reductions; 0 when that path does not apply) */
  int64_t gb_extra_bytes;
  /* "nodes" path (cinema_node_tables / cinema_siren_forward_nodes): forward weight stream without the linear_lats
   * segments (their work moves to per-subject node tables) and the backward stream, both per 32-feature slab (16 waves
   * per workgroup); -1 / 0 when the activation-stationary path does not apply */
  int64_t pk_ns, pk_nb;
  int32_t ns_steps, nb_steps;
} cinema_siren_desc;

int cinema_siren_plan(cinema_siren_desc *d /* host */);

/* fp32 master parameters -> packed, zero-padded compute copies (W, W^T, linear_lats split
 * into scale/shift halves and their transposes, final_linear padded to 16 rows).  One
 * launch.  Replaces nothing in the reference: it is the layout change that lets
 * siren.py:30-35 run as MFMA GEMMs. */
int cinema_siren_pack(const cinema_siren_desc *d /* host */, const float *params, void *packed, void *stream);
/* The same in parts (a mask) for a training loop that packs beside other work: 1 = the copies the forward reads,
 * 2 = the copies only the backward reads (the forward can start after part 1), 4 = the copies of the nodes path
 * (forward stream without linear_lats segments, row-major linear_lats for cinema_node_tables), 8 = row-major linear_lats
 * only (cinema_nodegrad_prepare). */
int cinema_siren_pack_part(const cinema_siren_desc *d /* host */, const float *params, void *packed, int32_t part,
                           void *stream);

/* Siren.forward (models/siren.py:59-61) incl. every SineLayer.forward (siren.py:26-36):
 *   out[n, :] = final_linear(cat(x_L, lat)),  x_{l+1} = sin(w z) | sin(w z s + t).
 * coords_t [Npad x 3] fp32 (registered coordinates), lat [Npad x LP] T,
 * saved [n_saved x Npad x HP] T (kept for backward), out [Npad x out_dim] fp32.
 * Inference (models/inr_decoder.py:38-48 runs under no_grad): saved may be NULL when the descriptor runs the
 * activation-stationary kernels (bf16, hidden <= 512; pk_fs >= 0) -- nothing is written but `out`. */
int cinema_siren_forward(const cinema_siren_desc *d /* host */, const float *params, const void *packed,
                         const float *coords_t, const void *lat, int64_t N, int64_t Npad, void *saved,
                         float *out, void *stream);

/* One SineLayer on its own -- SineLayer.forward((x, latent)) -> (out, latent), models/siren.py:26-36:
 *   z = linear(x);  out = sin(omega * z)                      (lat_feat == 0)
 *   [s | t] = linear_lats(latent);  out = sin(omega * z * s + t)   (lat_feat > 0)
 * through the same layer kernels the layer-by-layer stack uses.  All pointers are device fp32, dense row-major:
 * x [N x in_feat], latent [N x lat_feat], W [out_feat x in_feat], b [out_feat], Wl [2 out_feat x lat_feat], bl [2 out_feat],
 * out [N x out_feat].  precision selects the arithmetic of the GEMMs (bf16 operands with fp32 accumulation, or fp32).
 * in_feat <= 3 runs the first-layer (coordinate) kernels, larger inputs the MFMA ones; out_feat must be a multiple of 16.
 * work: cinema_sine_layer_work_bytes(d, N) bytes of device memory (padded operands, packed weights, what the backward
 * needs); the backward call must see the work buffer of the forward call it differentiates, with W unchanged. */
typedef struct cinema_sine_layer_desc {
  int32_t in_feat, lat_feat, out_feat, precision;
  float omega;
} cinema_sine_layer_desc;
int64_t cinema_sine_layer_work_bytes(const cinema_sine_layer_desc *d, int64_t N);
int cinema_sine_layer_forward(const cinema_sine_layer_desc *d, const float *x, const float *latent, const float *W,
                              const float *b, const float *Wl, const float *bl, int64_t N, void *work, float *out, void *stream);
/* What autograd does for siren.py:26-36: dout [N x out_feat] -> dx [N x in_feat], dlatent [N x lat_feat], dW, db, dWl, dbl
 * (shapes of W, b, Wl, bl; overwritten, not accumulated). */
int cinema_sine_layer_backward(const cinema_sine_layer_desc *d, const float *W, int64_t N, void *work, const float *dout,
                               float *dx, float *dlatent, float *dW, float *db, float *dWl, float *dbl, void *stream);

/* Modulator with cnn_kernel_size = k > 0 -- Conv3d(C, C, k, padding='same') over the latent table, models/inr_decoder.py:
 * 109-123 (the default k = 0 is the identity and needs no call).  in / out / dout / din [S x C x lx x ly x lz] fp32,
 * W [C x C x k x k x k], b [C] (torch layouts).  k <= 15 (even k: (k - 1) / 2 zeros in front, the rest behind, as torch pads), C a multiple of 8.  The convolution runs as GEMMs over an im2col
 * image of the table (three for the backward), in the arithmetic `precision` selects.  work: cinema_conv_mod_work_bytes
 * bytes of device memory; the two calls are independent (the backward rebuilds what it needs from `in`).
 * The backward overwrites din, dW, db (what autograd leaves for the table, conv.weight and conv.bias). */
int64_t cinema_conv_mod_work_bytes(int64_t S, int32_t C, int32_t lx, int32_t ly, int32_t lz, int32_t k, int32_t precision);
int cinema_conv_mod_forward(const float *in, const float *W, const float *b, int64_t S, int32_t C, int32_t lx, int32_t ly,
                            int32_t lz, int32_t k, int32_t precision, void *work, float *out, void *stream);
int cinema_conv_mod_backward(const float *in, const float *W, const float *dout, int64_t S, int32_t C, int32_t lx, int32_t ly,
                             int32_t lz, int32_t k, int32_t precision, void *work, float *din, float *dW, float *db,
                             void *stream);

/* Backward of the above (what autograd does for siren.py:26-36,59-61).
 * dout [Npad x out_dim] fp32 (rows >= N must be zero), gbuf [n_gbuf x Npad x HP] T scratch followed (at the next
 * multiple of 256 bytes) by gb_extra_bytes more bytes,
 * grad_params [n_params] fp32 (zeroed by this call, then accumulated with atomics),
 * dlat [Npad x LP] fp32 (overwritten), dcoords [Npad x 3] fp32 (overwritten). */
int cinema_siren_backward(const cinema_siren_desc *d /* host */, const float *params, const void *packed,
                          const float *coords_t, const void *lat, const void *saved, const float *dout,
                          int64_t N, int64_t Npad, void *gbuf, float *grad_params, float *dlat,
                          float *dcoords, int32_t flags, void *stream);

/* -------------------------------------------------------------------------------------
 * K1: affine registration + trilinear latent gather.
 * ------------------------------------------------------------------------------------- */

/* embed2affine + euler2rot (utils.py:126-166): rows x tf_dim (6|9|12) -> rows x 12
 * (A row-major 3x3, then t). */
int cinema_affine_from_embedding(const float *embed, int64_t rows, int32_t tf_dim, float *affine, void *stream);

/* latent table (S, C, G=lx*ly*lz) -> node-major shadow (S, G, C) so that the 8 corner
 * reads of one sample are contiguous in C (replaces the per-sample grid materialisation
 * `modulator(latent_vecs)[idcs_df]`, models/inr_decoder.py:33). */
int cinema_latent_to_node_major(const float *latents, int64_t S, int32_t C, int32_t G, float *shadow, void *stream);

/* INR_Decoder.transform (models/inr_decoder.py:64-71) + [idcs_df] gather (:33) +
 * spatial_interpolation (:73-85), fused.
 *   coords [N x 3] fp32, idx [N] int32 in [0,S), affine [S|N x 12] fp32 or NULL (no tfs),
 *   cond [N x cond_dims] fp32 or NULL,
 *   coords_t [Npad x 3] fp32, lat_out [Npad x LP] T (cols >= C+cond_dims and rows >= N zero). */
int cinema_affine_trilerp_fwd(const float *coords, const int32_t *idx, const float *affine,
                              int32_t affine_per_sample, const float *shadow, const float *cond, int64_t N,
                              int64_t Npad, int64_t S, int32_t C, int32_t lx, int32_t ly, int32_t lz,
                              int32_t cond_dims, int32_t precision, float *coords_t, void *lat_out, int32_t LP,
                              void *stream);

/* The same forward for a training loop whose batches are grouped by subject and whose embeddings are a per-subject
 * table: embed [S x tf_dim] (or NULL) and the latent table in its API layout (S, C, G) go in directly -- no affine
 * table, no node-major shadow.  64-sample chunks keep their subject's table in LDS; any idx order is computed
 * correctly, only a grouped one is fast.  3x3x3 grid, C <= 256, C % 4 == 0, LP % 4 == 0 (else: error, use the call
 * above).  Outputs are bit-identical to cinema_affine_from_embedding + cinema_affine_trilerp_fwd.
 * cond_by_subject != 0: cond is the per-subject condition table [S x cond_dims] and a sample takes row idx[n] (the
 * reference gathers conditions[idcs_df] on the host side of the call, build_atlas.py:74-75); 0: per-sample rows [N x cond_dims]. */
int cinema_affine_trilerp_fwd_sorted(const float *coords, const int32_t *idx, const float *embed, int32_t tf_dim,
                                     const float *latents, const float *cond, int64_t N, int64_t Npad, int64_t S,
                                     int32_t C, int32_t lx, int32_t ly, int32_t lz, int32_t cond_dims, int32_t cond_by_subject,
                                     int32_t precision, float *coords_t, void *lat_out, int32_t LP, void *stream);

/* The head of a training step in ONE launch: cinema_siren_pack (both parts), cinema_affine_trilerp_fwd_sorted and,
 * optionally, cinema_latent_to_node_major (shadow != NULL; only the K1 backward reads the shadow) and cinema_step_prepare
 * (sums != NULL; subj = idx) as block ranges of one kernel -- independent work that separate launches would run one
 * after the other on one stream, or with a 6-10 us join on two.  bf16 stacks on the activation-stationary path only
 * (hidden <= 512, `pk_fs >= 0` after planning) and the shapes cinema_affine_trilerp_fwd_sorted takes; LP and the
 * precision come from the descriptor.  Same bits as the separate calls.
 * node_work != NULL (the work buffer of the node-space backward, below, sized for this S and Npad; Npad % 128 == 0): the
 * node rows of the current table and the interpolation weights of the batch are written too, as two more block ranges
 * -- cinema_nodegrad_prepare is then called with flags 1 | 2 (or not at all, see cinema_siren_forward_node_prep). */
/* Gradient of the per-subject condition table from the per-sample rows dcond [N x cond_dims] the backward calls leave, for
 * a batch sorted by subject: dtable[s] = sum of the rows with idx == s (what autograd's index_add does for
 * conditions[idcs_df], build_atlas.py:74-75).  One workgroup per subject, ranges by binary search, deterministic. */
int cinema_cond_table_bwd(const int32_t *idx_sorted, const float *dcond, int64_t N, int64_t S, int32_t cond_dims, float *dtable,
                          void *stream);
int cinema_step_head(const cinema_siren_desc *d /* host */, const float *params, void *packed, const float *coords,
                     const int32_t *idx, const float *embed, int32_t tf_dim, const float *latents, const float *cond,
                     int64_t N, int64_t Npad, int64_t S, int32_t C, int32_t lx, int32_t ly, int32_t lz, int32_t cond_dims,
                     int32_t cond_by_subject, float *coords_t, void *lat_out, float *shadow, void *node_work, float *dcoords,
                     int64_t n_dcoords, float *sums,
                     const int64_t *voxel, const uint8_t *labels, int64_t V, int32_t n_cls, const float *class_w,
                     float *scratch, void *stream);

/* Backward of the above:
 *   dlat [Npad x LP] fp32, dcoords_t [Npad x 3] fp32 (gradient reaching the registered
 *   coordinates through the SIREN's first layer; the interpolation term is added here),
 *   -> dlatents (S, C, G) fp32 (overwritten, deterministic per-subject reduction),
 *      dembed: per-sample [N x tf_dim] (embed_per_sample) or table [S x tf_dim] (overwritten;
 *      per-subject sum in a fixed-shape reduction), may be NULL,  dcond [N x cond_dims] or NULL.
 *   work: 4-byte scratch of at least 2*S + 2 + 13*N entries. */
int cinema_affine_trilerp_bwd(const float *coords, const int32_t *idx, const float *embed, int32_t tf_dim,
                              int32_t embed_per_sample, const float *affine, const float *shadow,
                              const float *coords_t, const float *dlat, float *dcoords_t, int64_t N, int64_t Npad,
                              int64_t S, int32_t C, int32_t lx, int32_t ly, int32_t lz, int32_t cond_dims, int32_t LP,
                              float *dlatents, float *dembed, float *dcond, int32_t *work, void *stream);

/* Same, for a batch whose samples are already grouped by subject (idx non-decreasing; the trainer sorts every
 * step-sized chunk of the epoch's shuffled list once per epoch -- a batch is a set, build_atlas.py:69-88 does not depend
 * on the order inside it): no counting sort, the per-subject ranges come from a lower_bound inside the first kernel and
 * the reductions read their rows in place.  The precondition is the caller's; it is not checked. */
int cinema_affine_trilerp_bwd_sorted(const float *coords, const int32_t *idx, const float *embed, int32_t tf_dim,
                                     int32_t embed_per_sample, const float *affine, const float *shadow,
                                     const float *coords_t, const float *dlat, float *dcoords_t, int64_t N, int64_t Npad,
                                     int64_t S, int32_t C, int32_t lx, int32_t ly, int32_t lz, int32_t cond_dims, int32_t LP,
                                     float *dlatents, float *dembed, float *dcond, int32_t *work, void *stream);

/* cinema_affine_trilerp_bwd WITHOUT the caller's work buffer.  No counterpart in the reference: it is for callers that
 * cannot allocate device memory -- code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.
 * The grouping scratch lives in a file-scope device array of the library (2*MAX_S + 2 + 13*MAX_N 4-byte words = 3.3 MiB
 * per GPU the library runs on, held until the process ends); sizes beyond the maxima are rejected with an error.  The
 * call allocates and frees nothing.  Every call on one GPU shares that GPU's array: calls that overlap on one GPU (two
 * streams, two host threads) race -- serialise them.  Same outputs as cinema_affine_trilerp_bwd.
 * Before a HIP graph capture: make ONE ordinary (eager) call on each device first.  The first call on a device looks the
 * array's address up (hipGetSymbolAddress, which may load the code object) and keeps it; that lookup must not happen
 * inside a capture.  Later calls on the device make no runtime call besides the kernel launches. */
#define CINEMA_NOALLOC_MAX_N 65536
#define CINEMA_NOALLOC_MAX_S 4096
int cinema_affine_trilerp_bwd_noalloc(const float *coords, const int32_t *idx, const float *embed, int32_t tf_dim,
                                      int32_t embed_per_sample, const float *affine, const float *shadow,
                                      const float *coords_t, const float *dlat, float *dcoords_t, int64_t N, int64_t Npad,
                                      int64_t S, int32_t C, int32_t lx, int32_t ly, int32_t lz, int32_t cond_dims, int32_t LP,
                                      float *dlatents, float *dembed, float *dcond, void *stream);

/* cinema_siren_backward for a training loop that runs the split reduction of the weight-gradient tiles elsewhere: the
 * reduction's arguments are written to `pending` (HOST memory, CINEMA_PENDING_BYTES, opaque) instead of being launched;
 * grad_params holds the bias / first-layer / final-bias gradients only until the pending reduction has run
 * (cinema_affine_trilerp_bwd_sorted_pending runs it as a block range beside the K1 coordinate gradient, which is
 * independent of it).  Nothing is pending on the paths without split tiles (fp32, hidden < 256, Npad < 4096). */
#define CINEMA_PENDING_BYTES 4096
int cinema_siren_backward_deferred(const cinema_siren_desc *d /* host */, const float *params, const void *packed,
                                   const float *coords_t, const void *lat, const void *saved, const float *dout, int64_t N,
                                   int64_t Npad, void *gbuf, float *grad_params, float *dlat, float *dcoords, int32_t flags,
                                   void *pending /* host */, void *stream);
int cinema_affine_trilerp_bwd_sorted_pending(const float *coords, const int32_t *idx, const float *embed, int32_t tf_dim,
                                             int32_t embed_per_sample, const float *affine, const float *shadow,
                                             const float *coords_t, const float *dlat, float *dcoords_t, int64_t N,
                                             int64_t Npad, int64_t S, int32_t C, int32_t lx, int32_t ly, int32_t lz,
                                             int32_t cond_dims, int32_t LP, float *dlatents, float *dembed, float *dcond,
                                             int32_t *work, const void *pending /* host */, void *stream);

/* -------------------------------------------------------------------------------------
 * "Nodes" path of the bf16 chain kernels (hidden <= 512): the FiLM modulation and the latent part of final_linear are
 * evaluated in NODE space.  linear_lats (models/siren.py:32) and the lat columns of final_linear (siren.py:61) are
 * linear, and a sample's interpolated latent (models/inr_decoder.py:73-85) is a convex combination of its subject's
 * grid nodes, so  linear_lats(lat[n]) = sum_k w[n][k] * (Wl * table[s][k]) + bl:  the projections are computed once per
 * call for the S * G node vectors (cinema_node_tables) instead of N samples, the chain kernels interpolate them with one
 * fp16 MFMA step per modulated layer and half, and per-sample latent rows never exist.  Same function, different
 * rounding (fp16 node projections and weights instead of a bf16 GEMM on bf16 latent rows).
 * Needs lx*ly*lz <= 30 grid nodes; condition values are per SUBJECT (cond_subject [S x cond_dims]; the training loop's
 * and the atlas evaluation's case).  Per-sample conditions / explicit latent rows use the per-sample entry points.
 * ------------------------------------------------------------------------------------- */
/* buffer sizes of the node tables for S subjects: T [t_elems] bf16, U [u_bytes], V [v_bytes] */
int cinema_nodes_sizes(const cinema_siren_desc *d /* host */, int64_t S, int64_t *t_elems, int64_t *u_bytes, int64_t *v_bytes);

/* INR_Decoder.transform (models/inr_decoder.py:64-71) + the interpolation WEIGHTS of spatial_interpolation (:73-85):
 *   coords [N x 3], idx [N], embed [S x tf_dim] or NULL -> coords_t [rows x 3] fp32, wrows [rows x 32] fp16
 *   (8 corner weights of the sample's cell at their nodes' fragment positions, 1.0 in the bias slot; rows >= N zero).
 * Same registered coordinates, bit for bit, as cinema_affine_trilerp_fwd. */
int cinema_node_weights(const float *coords, const int32_t *idx, const float *embed, int32_t tf_dim, int64_t N, int64_t rows,
                        int64_t S, int32_t lx, int32_t ly, int32_t lz, float *coords_t, void *wrows, void *stream);

/* Sample tiles of a batch grouped by subject (idx non-decreasing -- the caller's precondition, not checked): every
 * subject's rows are cut into tiles of at most 16 rows so that each MFMA sample tile of the chain kernels belongs to ONE
 * subject.  tiles: int32 [4 * (rows / 16 + S + 2)]: word 0 = number of tiles, then {subject, first row, rows, 0} per
 * tile.  Rows [N, rows) (rows = Npad: the padding of the row buffers) are appended to the last subject. */
int cinema_node_tiles(const int32_t *idx, int64_t N, int64_t rows, int64_t S, void *tiles, void *stream);

/* Node rows T [S x 32 x LP] bf16 from the latent table (S, C, lx, ly, lz) (+ cond_subject), then their projections
 * U (linear_lats of every modulated layer, as fp16 MFMA fragments) and V (lat columns of final_linear).
 * Needs cinema_siren_pack_part(.., 4, ..) of the current parameters. */
int cinema_node_tables(const cinema_siren_desc *d /* host */, const float *params, const void *packed, const float *latents,
                       const float *cond_subject, int64_t S, int32_t C, int32_t lx, int32_t ly, int32_t lz, int32_t cond_dims,
                       void *T, void *U, void *V, void *stream);

/* cinema_siren_forward / cinema_siren_backward on the nodes path: `lat` is replaced by (wrows, tiles, U, V, S); the
 * batch must be grouped by subject (cinema_node_tiles).
 * The backward still takes `lat` [Npad x LP] T for the per-sample d-latent / linear_lats weight-gradient GEMMs. */
int cinema_siren_forward_nodes(const cinema_siren_desc *d /* host */, const float *params, const void *packed,
                               const float *coords_t, const void *wrows, const void *tiles, const void *U, const void *V,
                               int64_t S, int64_t N, int64_t Npad, void *saved, float *out, int32_t flags, void *stream);
int cinema_siren_backward_nodes(const cinema_siren_desc *d /* host */, const float *params, const void *packed,
                                const float *coords_t, const void *lat, const void *wrows, const void *tiles, const void *U,
                                int64_t S, const void *saved, const float *dout, int64_t N, int64_t Npad, void *gbuf,
                                float *grad_params, float *dlat, float *dcoords, int32_t flags, void *pending /* host, may be NULL */,
                                void *stream);

/* -------------------------------------------------------------------------------------
 * Backward with the latent side reduced in NODE space ("node gradients"; bf16 chain kernels, hidden <= 512, at most 30
 * grid nodes, batch grouped by subject: idx non-decreasing -- the caller's precondition, not checked).
 * What autograd does for models/inr_decoder.py:64-85 + models/siren.py:26-36,59-61 in ONE call: it replaces
 * cinema_siren_backward + cinema_affine_trilerp_bwd_sorted for a training loop.  The interpolated latent of a sample is a
 * convex combination of its subject's grid nodes and linear_lats / the latent columns of final_linear are linear, so the
 * per-sample rows X[n] = [ds_j | dt_j]_j, dOut that leave the chain kernels are reduced ONCE to
 *   dU[s][k] = sum_{n of s} w[n][k] X[n]   (k: the subject's 27 nodes + a constant-1 slot)   and
 *   dwn[n][k] = X[n] . linear_lats(table[s][k] - mean_k table[s][k]),
 * after which d linear_lats.weight / .bias, d final_linear.weight[:, H:] and d table are GEMMs over S*32 node rows
 * instead of N sample rows and  d coords[n] += sum_k dwn[n][k] d w[n][k] / d coords  (no per-sample d-latent rows,
 * no per-sample latent reduction).  Same function as the per-sample calls; different rounding (bf16 interpolation
 * weights and node sums).
 *   work: cinema_nodegrad_work_bytes(d, S, Npad) bytes of device memory, zeroed once by the caller.  Besides the node
 *     tables it carries 16 replicas of the chain kernel's column sums (b_l and first-layer weight gradients, hidden 256 /
 *     512): the CHAIN stage adds into them, the DW stage's tail sums them into grad_params and clears them -- a call with
 *     the CHAIN bit alone leaves those gradients there until a call that runs the DW stage.
 *   cinema_nodegrad_prepare: per step, after the weights were packed (cinema_siren_pack_part with bits 2 | 8, or
 *     cinema_step_head) -- interpolation weights of the batch, node rows and centred node projections of the current
 *     table (`latents` (S, C, lx, ly, lz): the table K1 interpolates; cond_subject [S x cond_dims] or NULL).
 *     flags & 1: cinema_step_head already wrote the node rows of this table into `work` (one launch instead of two);
 *     flags & 2: ... and the interpolation weights of this batch; flags & 4: the projections are left to
 *     cinema_siren_forward_node_prep (with 1 | 2 | 4 the call launches nothing).
 *   cinema_siren_backward_nodegrad: flags as cinema_siren_backward (SKIP_WEIGHT_GRADS, GRADS_CLEARED, DCOORDS_CLEARED,
 *     WEIGHTS_FIRST / DLAT_ONLY: the first call leaves grad_params complete, the second one the table-side gradients;
 *     the PHASE bits time single stages on whatever earlier stages left in the buffers: CHAIN = chain kernel +
 *     k_node_grad, DW = weight-gradient launch + tail, DLAT = the table-side tail alone).
 *     Outputs (overwritten): grad_params, dlatents (S, C, lx, ly, lz), dembed [S x tf_dim] or NULL, dcond [S x cond_dims]
 *     or NULL (per-SUBJECT condition gradient).  dcoords [Npad x 3] holds the first layer's part only.
 * ------------------------------------------------------------------------------------- */
/* cinema_siren_forward + the node projections of cinema_nodegrad_prepare in ONE launch: the projections need only the
 * packed weights and the node rows (both left by cinema_step_head with node_work != NULL), so they run as tail workgroups
 * of the forward launch, on the CUs its shorter second round of workgroups leaves idle.  A training loop then calls
 * cinema_step_head(node_work) -> this -> loss -> cinema_siren_backward_nodegrad and never cinema_nodegrad_prepare. */
int cinema_siren_forward_node_prep(const cinema_siren_desc *d /* host */, const float *params, const void *packed,
                                   const float *coords_t, const void *lat, int64_t N, int64_t Npad, void *saved, float *out,
                                   void *node_work, int64_t S, int32_t lx, int32_t ly, int32_t lz, void *stream);
int cinema_nodegrad_eligible(const cinema_siren_desc *d /* host */, int32_t lx, int32_t ly, int32_t lz);
int64_t cinema_nodegrad_work_bytes(const cinema_siren_desc *d /* host */, int64_t S, int64_t Npad);
int cinema_nodegrad_prepare(const cinema_siren_desc *d /* host */, const void *packed, const float *coords, const int32_t *idx,
                            const float *embed, int32_t tf_dim, const float *latents, const float *cond_subject, int64_t N,
                            int64_t Npad, int64_t S, int32_t C, int32_t lx, int32_t ly, int32_t lz, int32_t cond_dims,
                            void *work, int32_t flags, void *stream);
int cinema_siren_backward_nodegrad(const cinema_siren_desc *d /* host */, const float *params, const void *packed,
                                   const float *coords_t, const void *saved, const float *dout, int64_t N, int64_t Npad,
                                   void *gbuf, float *grad_params, float *dcoords, int32_t flags, const float *coords,
                                   const int32_t *idx, const float *embed, int32_t tf_dim, int64_t S, int32_t C, int32_t lx,
                                   int32_t ly, int32_t lz, int32_t cond_dims, void *work, float *dlatents, float *dembed,
                                   float *dcond, void *stream);

/* -------------------------------------------------------------------------------------
 * K3: ground-truth sampling + loss + dLoss/dOutput (Criterion.forward, utils.py:88-104,
 * and the integer gather `vol[c_nz]` of data_loading/dataset.py:87-89).
 *   target source A (dense):  target [N x (sr_dims+1)] fp32, last column = label.
 *   target source B (volumes): voxel [N] int64 linear voxel index, subj [N] int32,
 *     vol_int [n_mod pointers on device -> per-modality concatenated fp32 volumes],
 *     given as intens [S*V x sr_dims] fp32 (subject-major, modality innermost) and
 *     labels [S*V] uint8; V voxels per subject.
 *   sums (fp32[4], zeroed by the call): [0] sum |d| or d^2, [1] sum w*nll, [2] sum w (pre-pass, only with class_w), [3] sum w
 *   dout [Npad x out_dim] fp32 = d total / d output with
 *     total = sr_scale*sr + seg_scale*ce   (sr_scale = sr_weight, seg_scale = seg_weight^2,
 *     the reference applies seg_weight twice, utils.py:95,103).
 *   class_w [n_cls] fp32 or NULL.  loss_metric: 0 = mse, 1 = l1.
 * ------------------------------------------------------------------------------------- */
int cinema_sample_loss_fwd_bwd(const float *out, const float *target, const int64_t *voxel, const int32_t *subj,
                               const float *intens, const uint8_t *labels, int64_t V, int64_t N, int64_t Npad,
                               int32_t sr_dims, int32_t n_cls, int32_t loss_metric, const float *class_w,
                               float sr_scale, float seg_scale, float *sums, float *dout, float *target_out,
                               void *stream);

/* The same call for a training loop that prepared `sums` beforehand with cinema_step_prepare (typically on a second
 * stream, beside the forward pass): no memset, no class-weight pre-pass -- sums[0], [1], [3] must be zero and, with
 * class_w, sums[2] = sum of class_w[label] over the batch (labels do not depend on the network output). */
int cinema_sample_loss_fwd_bwd_prepared(const float *out, const float *target, const int64_t *voxel, const int32_t *subj,
                                        const float *intens, const uint8_t *labels, int64_t V, int64_t N, int64_t Npad,
                                        int32_t sr_dims, int32_t n_cls, int32_t loss_metric, const float *class_w,
                                        float sr_scale, float seg_scale, float *sums, float *dout, float *target_out,
                                        void *stream);

/* The prepared call for one rank of a data-parallel job (reference: utils.py:57-79 evaluated on the union of the ranks'
 * batches; the reference itself is single-device).  norm = device fp32[2] = {samples, sum of class_w[label]} of the UNION
 * batch divided by the number of ranks: the means and dOut are normalised by these instead of N and sums[2], so that
 * the rank-average of the gradients is the union batch's gradient also with non-uniform class weights or ragged N
 * (without class_w pass norm[1] = norm[0]). */
int cinema_sample_loss_fwd_bwd_normed(const float *out, const float *target, const int64_t *voxel, const int32_t *subj,
                                      const float *intens, const uint8_t *labels, int64_t V, int64_t N, int64_t Npad,
                                      int32_t sr_dims, int32_t n_cls, int32_t loss_metric, const float *class_w,
                                      float sr_scale, float seg_scale, float *sums, float *dout, float *target_out,
                                      const float *norm, void *stream);

/* Per-step clearing for a training loop (build_atlas.py:69-88 does optimizer.zero_grad() + a loss per step), ONE launch:
 * zeroes dcoords [n_dcoords] fp32 (may be NULL) and sums[0..3]; with class_w it also sets sums[2] to the sum of
 * class_w[labels[subj*V + voxel]] over the N samples (the normaliser of the weighted cross-entropy, utils.py:86,95).
 * scratch: 4-byte device scratch of CINEMA_PREPARE_SCRATCH entries, zeroed ONCE by the caller before the first call
 * (per-workgroup partial sums + the arrival counter the last workgroup resets); calls sharing one scratch must not overlap.
 * Independent of the forward pass: meant for a second stream. */
#define CINEMA_PREPARE_SCRATCH 260
int cinema_step_prepare(float *dcoords, int64_t n_dcoords, float *sums, const int64_t *voxel, const int32_t *subj,
                        const uint8_t *labels, int64_t V, int64_t N, int32_t n_cls, const float *class_w, float *scratch,
                        void *stream);

/* Dense-inference epilogue (models/inr_decoder.py:50-57): clamp, per-modality min/max over
 * the whole volume (two passes: cinema_infer_minmax then cinema_infer_finish), argmax,
 * softmax, channel concat -> vol [G x (sr_dims + 1 + n_cls)] fp32.
 * minmax: fp32 [2*sr_dims] = mins then maxs; caller initialises mins to 1.0 and maxs to 0.0
 * (values are clamped to [0,1] before the reduction). */
int cinema_infer_minmax(const float *out, int64_t G, int32_t out_dim, int32_t sr_dims, float *minmax, void *stream);
int cinema_infer_finish(const float *out, int64_t G, int32_t out_dim, int32_t sr_dims, const float *minmax,
                        float *vol, void *stream);

/* The same epilogue for a volume evaluated in chunks, without a whole-volume copy of the network outputs
 * (models/inr_decoder.py:44-57 keeps the (G, out_dim) tensor; at 310^3 that is 1.43 GB): per chunk, cinema_infer_chunk
 * writes the clamped intensities, argmax and softmax into the chunk's rows of vol [G x (out_dim + 1)] and folds the chunk
 * into minmax (initialised as for cinema_infer_minmax); after the last chunk cinema_infer_rescale applies
 * (x - min) / (max - min) to the intensity channels of the whole volume.  Results equal the two-pass calls bit for bit. */
int cinema_infer_chunk(const float *out, int64_t G, int32_t out_dim, int32_t sr_dims, float *minmax, float *vol_rows,
                       void *stream);
int cinema_infer_rescale(float *vol, int64_t G, int32_t out_dim, int32_t sr_dims, const float *minmax, void *stream);

/* Connected components of a 3-d mask [X x Y x Z] uint8 with 6-connectivity (what scipy.ndimage.label computes for the
 * post-hoc masking of a reconstruction, models/inr_decoder.py:92-104): labels[v] = 0 outside the mask, else 1 + the
 * smallest linear index of v's component (so labels order like scipy's raster-scan numbering).  init != 0 starts from
 * the mask; every call runs `sweeps` propagation sweeps and leaves in *changed (device int32) how many waves still
 * updated a label in them: call again with init = 0 until it reads 0. */
int cinema_cc_label(const uint8_t *mask, int32_t X, int32_t Y, int32_t Z, int32_t *labels, int32_t *changed, int32_t sweeps,
                    int32_t init, void *stream);

/* Coordinates [count x 3] of the run [start, start + count) of the dense evaluation grid, x-major flattening of the three
 * axis vectors (utils.py:235-250 builds the whole meshgrid, 357 MB at 310^3; the chunked evaluation makes each run). */
int cinema_grid_coords(const float *ax, const float *ay, const float *az, int64_t nx, int64_t ny, int64_t nz, int64_t start,
                       int64_t count, float *coords, void *stream);

/* -------------------------------------------------------------------------------------
 * Optimiser step: torch.optim.AdamW exactly as the reference configures it
 * (build_atlas.py:432-453: one group per parameter table, decoupled weight decay, no amsgrad),
 * all groups in one launch.  `step` is the 1-based step count of the group (bias correction).
 * ------------------------------------------------------------------------------------- */
#define CINEMA_ADAMW_MAX_GROUPS 8
typedef struct cinema_adamw_group {
  float *p;        /* parameters (updated in place) */
  const float *g;  /* gradients */
  float *m;        /* exp_avg */
  float *v;        /* exp_avg_sq */
  int64_t n;       /* elements */
  float lr, beta1, beta2, eps, weight_decay;
  int32_t step;
} cinema_adamw_group;

int cinema_adamw_step(const cinema_adamw_group *groups /* host */, int32_t n_groups, void *stream);
/* Same update; the gradients are zeroed once they have been consumed (optimizer.step() + the next step's
 * optimizer.zero_grad(), build_atlas.py:82-87, in one pass over the gradient). */
int cinema_adamw_step_clear(const cinema_adamw_group *groups /* host */, int32_t n_groups, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CINEMA_HIP_H */
