// Node-space backward of the latent side ("node gradients"): internal interface between nodegrad.hip (its kernels) and
// siren.hip (which sequences them inside the backward).
//
// The interpolated latent of a sample is a convex combination of its subject's grid nodes,
//     lat[n] = sum_k w[n][k] * table[s][k]          (models/inr_decoder.py:73-85, grid_sample on a 3^3 grid)
// and everything downstream of it up to the FiLM phase is linear (models/siren.py:32, :61).  So the three gradients
// that leave the chain kernels through the per-sample rows X[n] = [ds_j | dt_j]_j (+ dOut) -- d linear_lats.weight,
// d table and d coords (through the interpolation weights) -- only need
//     dU[s][k][f]  = sum_{n of s} w[n][k] * X[n][f]        (32 node slots x F features per subject)
//     dwn[n][k]    = sum_f X[n][f] * U'[s][k][f]           (U' = linear_lats(table[s][k] - mean_k table[s][k]))
// k_node_grad reads X ONCE and forms both; the weight gradient, the table gradient and the bias gradient are then GEMMs
// over S*32 node rows instead of N sample rows, and d coords = sum_k dwn[n][k] * d w[n][k] / d coords.
//
// Work buffer (caller-owned, cinema_nodegrad_work_bytes; zeroed once by the caller, rows >= S*32 stay zero):
//   offs   int32 [S + 1]            first row of every subject in the grouped batch
//   wt     bf16 [Npad/16][32][16]   interpolation weights, per 16-row block, slot-major (MFMA operand order)
//   wrow   bf16 [Npad][32]          the same weights row-major; axt int32 [Npad][3][4]: the rows' interpolation axes
//   T      bf16 [Rp][LP]            node rows of the table (+ the subject's conditions), Rp = S*32 rounded up to 128
//   Tc     bf16 [Rp][LP]            the same rows minus the subject's mean over its nodes (noderows_device.h); T and Tc sit
//                                   right behind offs, so their places depend on S only (cinema_step_head writes them)
//   Up     bf16 [S*32][F]           centred projections U', F = n_mod * 2 * HP
//   Vp     bf16 [S*32][16]          the same through final_linear.weight[:, H:]
//   dU     bf16 [Rp][F], dUo16 bf16 [Rp][16], dUo32 fp32 [Rp][16]
//   dcn    fp32 [nparts][Npad][4]   per-(tensor, column group) parts of d coords through the interpolation weights
//   dT     fp32 [dt_parts][Rp][LP]  per-(segment, k piece) parts of d (node rows), dt_parts = 2 n_mod * dtk (+ 1: final_linear term)
//   rep    fp32 [NG_REP][rep_w]     replicas of the chain kernel's column sums (b_l gradients [n_layers][HP], then the first
//                                   layer's weight gradient [HP][4]): workgroup b adds into replica b % NG_REP instead of 256
//                                   workgroups into one address; k_ng_tail sums them into the flat gradient and clears them
#pragma once
#include "common.h"

#define NG_SLOTS 32
#define NG_ONE_SLOT 31
#define NG_REP 16

struct NgPlan {
  long S, Npad, Rp;
  int n_mod, F, nparts, ncg, HP, LP, dt_parts;
  int layer_of[CINEMA_MAX_LAYERS];
  size_t offs, T, Tc, wt, wrow, axt, Up, Vp, dU, dUo16, dUo32, dcn, dT, rep, bytes;
  int rep_w;
};
bool ng_eligible(const cinema_siren_desc *d, int lx, int ly, int lz);
NgPlan ng_plan(const cinema_siren_desc *d, int64_t S, int64_t Npad);

// T and Tc from the latent table in its API layout (a launch of its own; the step head has the same body as a block range)
int ng_launch_rows(const cinema_siren_desc *d, void *work, const float *latents, const float *cond_subject, int64_t S, int C,
                   int lx, int ly, int lz, int cond_dims, hipStream_t s);

// X -> dU, dUo, dcn (one launch).  gbuf: the backward scratch tensors (ds / dt slots, dOut16 in the last slot).
int ng_launch_grad(const cinema_siren_desc *d, const NgPlan &p, void *work, const void *gbuf, const float *coords_t,
                   const int32_t *idx, int64_t N, int lx, int ly, int lz, const void *packed, const float *params, hipStream_t s);

// Tail of the backward in one launch (block ranges; NULL / 0 switches a range off):
//   reduce:  split reduction of k_dw256's partial tiles (opaque Dw256Args, may be NULL)
//   bias:    linear_lats.bias gradients from the constant-1 slot of dU (grad_params != NULL)
//   table:   d table (S, C, G) and d cond table from the dT parts (dlatents != NULL)
//   tf:      embedding-table gradient from d coords (dembed != NULL)
struct NgTailArgsHost {
  const void *reduce;            // const Dw256Args * or NULL
  float *grad_params;            // flat gradient (bias range) or NULL
  float *dlatents; float *dcond; // (S, C, G), (S, cond_dims) or NULL
  float *dembed; const float *embed; int tf_dim; const float *coords; const float *dcoords; const int32_t *idx;
  int64_t N; int C, G, cond_dims;
};
int ng_launch_tail(const cinema_siren_desc *d, const NgPlan &p, void *work, const NgTailArgsHost &t, hipStream_t s);
