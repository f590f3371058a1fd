// Node-space modulation ("nodes" path of the bf16 chain kernels): internal interface between node.hip (tables, weights,
// reductions) and fused_nodes.hip (chain kernels).
//
// linear_lats is linear and the interpolated latent of a sample is a convex combination of its subject's grid nodes, so
//     linear_lats(lat[n]) = sum_k w[n][k] * (Wl * table[s][k]) + bl
// The per-node projections U[s][k] = Wl * table[s][k] are computed once per step for the S * G node vectors instead of
// for every sample (models/siren.py:32 applied to S*G rows instead of N), and the chain kernels interpolate them with
// ONE k32 MFMA step per modulated layer and half:  s[f][n] = sum_slot U[f][slot] * w[slot][n].
//
// Slots: 32 per subject.  slot < G = lx*ly*lz is grid node (ix*ly + iy)*lz + iz; slot 31 carries the constant 1 (its U
// column is the linear_lats bias, its reduction the bias gradient); the others are zero.  Per-subject condition values
// live in the condition columns of every node row (the weights of a sample sum to 1).
//
// w rows  [rows][32] fp16, position p of a row holds slot sigma(p):  p = 8*g + e  ->  slot 4*g + e (e < 4),
//         16 + 4*g + (e - 4) (e >= 4): the order in which the 16x16 accumulator map hands a lane its 8 slots.
// T       [S][32][LP] bf16: node rows (slot order, not permuted): latent channels, then the subject's conditions.
// U       fp16 MFMA A fragments  [mod layer j][slab of 64 features][subject][half h][ft][lane][8]  (one k32 step = 4 KB
//         per half): lane (i, gk) of fragment ft holds feature slab*64 + 32*(ft>>1) + 8*(i>>2) + 4*(ft&1) + (i&3),
//         positions 8*gk .. 8*gk+7.  This feature map (also that of the nodes path's weight streams, pk_ns / pk_nb)
//         leaves accumulator register r of fragment ft in lane (i, g) with feature slab*64 + 32*(ft>>1) + 8*g + 4*(ft&1)
//         + r of sample i: the fragment pair p = ft>>1 of a lane is one 16-byte chunk of 8 consecutive features, and the
//         four lanes g of a row cover 64 contiguous bytes -- every epilogue load / store instruction moves whole 64-byte
//         segments per row (fused.h's map gives 16 consecutive features per lane, written as two 16-byte halves).
// V       the same for final_linear.weight[:, H:] (16 output rows): [subject][lane][8] fp16, lane i = output row.
#pragma once
#include "common.h"

#define NODE_SLOTS 32
#define NODE_ONE_SLOT 31

typedef _Float16 f16_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__host__ __device__ inline int node_pos_of_slot(int slot) {      // inverse of sigma
  return slot < 16 ? 8 * (slot >> 2) + (slot & 3) : 8 * ((slot - 16) >> 2) + 4 + (slot & 3);
}

struct NodeDims { int n_mod, nslab; long S; };
inline long node_u_bytes(int n_mod, int nslab, long S) { return (long)n_mod * nslab * S * 8192 + 8192; }
inline long node_v_bytes(long S) { return S * 1024; }

bool nodes_eligible(const cinema_siren_desc *d, int lx, int ly, int lz);
