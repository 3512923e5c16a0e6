/*
 * pc3d.h — C-ABI of libpc3d_hip.so, the MI355X (gfx950) point-set kernel library behind the
 * attack-iteration hot path of LI-Yiquan/3DPointCloudAttack.
 *
 * Conventions (all entry points):
 *   - every pointer is a CALLER-OWNED DEVICE buffer (e.g. torch.Tensor.data_ptr()); the library allocates
 *     nothing and keeps no mutable global state besides a thread-local error string;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the null stream) and the call
 *     returns before completion; nothing inside synchronises, so every entry point is hipGraph-capturable;
 *   - return value 0 = ok; >0 = hipError_t of the failing runtime call; <0 = argument error
 *     (PC3D_EINVAL). No C++ exception crosses the boundary. pc3d_last_error() gives the text.
 *   - point sets are fp32 and addressed by ELEMENT strides (batch, point, channel) so both reference
 *     layouts are zero-copy: [B,N,3] is (3N,3,1); [B,3,N] is (3N,1,N).
 *   - indices are int32 on the wire (N < 2^31); the Python mirror widens to int64 where the reference
 *     returns LongTensors.
 *
 * The reference has one FFI precedent whose style this follows: `extern "C" void render_ball(...)` loaded
 * with ctypes and fed raw caller-owned buffers (reference utils/render_balls_so.cpp:12-14,
 * utils/show3d_balls.py:22,80-84). Each function below names the reference code it replaces.
 */
#ifndef PC3D_H
#define PC3D_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PC3D_OK 0
#define PC3D_EINVAL (-22)

/* Library version (major*10000 + minor*100 + patch). */
int pc3d_version(void);
/* Thread-local text of the last non-zero return on this thread ("" if none). */
const char* pc3d_last_error(void);

/* ---------------------------------------------------------------------------------------------------------
 * K1  nearest neighbour of every query point in a reference set (squared L2, direct-difference form).
 *   min_d2[b,i] = min_j |q[b,i,:]-r[b,j,:]|^2 ; idx[b,i] = lowest j attaining it.
 * Replaces the [B,N,M] materialisations + min() of
 *   attack/CW/CW_utils/distance.py:15-32,40-50,58-70 (batch_pairwise_dist + ChamferDistance/HausdorffDistance),
 *   utils/dis_utils_torch.py:8-28 (cdist + min), utils/dis_utils_numpy.py:13-38 (scipy distance_matrix + min),
 *   attack/GeoA3/knn_utils.py:10-20 with K=1 (attack/GeoA3/loss_utils.py:36-58).
 * q: B x N points, r: B x M points, strides in elements. min_d2: [B,N] f32, idx: [B,N] i32 (either may be NULL).
 * ------------------------------------------------------------------------------------------------------- */
int pc3d_nn_f32(const float* q, int64_t q_bs, int64_t q_ps, int64_t q_cs,
                const float* r, int64_t r_bs, int64_t r_ps, int64_t r_cs,
                int B, int N, int M, float* min_d2, int32_t* idx, void* stream);
/* pc3d_nn_f32 that also writes the indices as int64 (idx64 [B,N], may be NULL; idx may be NULL too): the pytorch3d-style
 * knn_points of attack/GeoA3/knn_utils.py:22-55 returns int64 indices — written here instead of by a conversion launch. */
int pc3d_nn_i64_f32(const float* q, int64_t q_bs, int64_t q_ps, int64_t q_cs,
                    const float* r, int64_t r_bs, int64_t r_ps, int64_t r_cs,
                    int B, int N, int M, float* min_d2, int32_t* idx, int64_t* idx64, void* stream);


/* Both directions in ONE launch (grid.z = 2): a->b into (dA,iA) [B,N]; b->a into (dB,iB) [B,M]. */
int pc3d_nn_bidir_f32(const float* a, int64_t a_bs, int64_t a_ps, int64_t a_cs,
                      const float* b, int64_t b_bs, int64_t b_ps, int64_t b_cs,
                      int B, int N, int M,
                      float* dA, int32_t* iA, float* dB, int32_t* iB, void* stream);

/* Both directions from ONE evaluation of every distance (the two-scan form above computes each d(a_i,b_j) twice):
 * a workgroup keeps 64*Q points of `a` in registers, scans a range of `b` from LDS, takes the row minima as before
 * and reduces the column minima across the lanes with a halving butterfly (v_permlane32/16_swap + DPP); what one
 * workgroup cannot finish (column minima per tile of `a`; row minima when `b` is split to fill the chip) goes through
 * the caller's workspace and is folded by a second, small launch. Same results as pc3d_nn_bidir_f32, bit for bit
 * (distances; indices = lowest index attaining the minimum). Replaces the same reference code as pc3d_nn_f32 for the
 * callers that need both directions: distance.py:40-50,58-70 (ChamferDistance/HausdorffDistance return both),
 * utils/dis_utils_numpy.py:23-38, utils/dis_utils_torch.py:14-28, attack/GeoA3/loss_utils.py:36-46.
 * ws: device scratch of at least pc3d_nn_bidir_shared_ws_bytes(B,N,M) bytes, 16-byte aligned, owned by the caller,
 * free to reuse once the stream has passed this call. dA/iA/dB/iB: any may be NULL. */
int64_t pc3d_nn_bidir_shared_ws_bytes(int B, int N, int M);
int pc3d_nn_bidir_shared_f32(const float* a, int64_t a_bs, int64_t a_ps, int64_t a_cs,
                             const float* b, int64_t b_bs, int64_t b_ps, int64_t b_cs,
                             int B, int N, int M,
                             float* dA, int32_t* iA, float* dB, int32_t* iB,
                             void* ws, int64_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * K8b  point-wise dense layer on fp32 MFMA:  Y[M,N] = act( gate(X)[M,K] . W[N,K]^T + bias[N] ).
 * Replaces the library GEMM + activation passes of the shared MLPs of PointNet++ set abstraction
 * (model/pointnet2_utils.py:190-197,243-257), DGCNN's EdgeConv products and conv5 (model/dgcnn.py:297-320) and
 * CurveNet's 1x1 convolutions (model/curvenet_util.py:189-193,321-331), with eval BatchNorm folded into W / bias.
 * Backward to the input (frozen weights, no weight gradient): the same entry point on W^T with X = dY and
 * gate = the layer's output Y:  dX = (dY * act'(Y)) . W   (gate_slope = 0 for ReLU, = slope for LeakyReLU).
 * X: [M,K] row stride ldx; W: [N,K] row-major; bias [N] or NULL; gate [M,K] row stride ldg or NULL (X is then read
 * as gate > 0 ? X : gate_slope * X); act: 0 none, 1 ReLU, 2 LeakyReLU(slope); Y: [M,N] row stride ldy. Exact fp32
 * (v_mfma_f32_32x32x2_f32). Any M, N, K >= 1.
 * ------------------------------------------------------------------------------------------------------- */
int pc3d_gemm_nt_f32(const float* X, int64_t ldx, const float* W, const float* bias, const float* gate, int64_t ldg,
                     float gate_slope, int M, int N, int K, int act, float slope, float* Y, int64_t ldy, void* stream);
/* The same product with an EXPLICIT tiling. pc3d_gemm_nt_f32 picks its tile shape from (N, K) alone and sums every
 * output element's products in ascending k — its result for a row never depends on how many rows share the launch.
 * variant: 0 / 1 128x128 tiles, 4 waves (double / single buffered); 2 128x64 (N <= 64); 3 256x64; 4 64x128; 5 / 6
 * 128x128, 8 waves (single / double buffered); 8 as 5 with four workgroups per CU; 11: 64x64 tiles with the K steps
 * split over two groups of waves, 12: 32x64 tiles with four K groups — for launches with few tiles and a long K
 * (CurveNet's deep levels); the K split sums an element's products in a different (fixed) order, so a caller chooses
 * it from per-cloud shapes, not from the batch (ops.gemm_variant). Anything else: PC3D_EINVAL. */
int pc3d_gemm_nt_tiled_f32(const float* X, int64_t ldx, const float* W, const float* bias, const float* gate,
                           int64_t ldg, float gate_slope, int M, int N, int K, int act, float slope, float* Y,
                           int64_t ldy, int variant, void* stream);
/* The same layer with a residual branch summed in before the activation:  Y = act(X . W^T + bias + R)  — the tail
 * of a CurveNet CIC block, relu(conv2(..) + shortcut) (model/curvenet_util.py:372-376). R: [M,N] row stride ldr.
 * Its backward: G = pc3d_gate_f32(dY, Y) once, then dR = G and dX = pc3d_gemm_nt_f32(G, W^T). */
int pc3d_gemm_nt_res_f32(const float* X, int64_t ldx, const float* W, const float* bias, const float* R, int64_t ldr,
                         int M, int N, int K, int act, float slope, float* Y, int64_t ldy, int variant, void* stream);
/* (variant: < 0 = the default tiling, else as pc3d_gemm_nt_tiled_f32) */
/* out[i] = y[i] > 0 ? g[i] : slope * g[i] over n contiguous floats (16-byte aligned buffers): the derivative of a
 * (Leaky)ReLU taken from its OUTPUT's sign, for layers whose pre-activation had more than one producer. */
int pc3d_gate_f32(const float* g, const float* y, int64_t n, float slope, float* out, void* stream);

/* Point normals from the k-NN covariance (attack/GeoA3/utility.py:43-92 estimate_normal): eigenvector of the smallest
 * eigenvalue of the 3 x 3 covariance of the k neighbours (closed form, double arithmetic), sign fixed against the
 * summed centred neighbours (:73-75). Replaces a batched 3 x 3 symeig (rocsolver: 5 ms per call at B=32, N=1024).
 * x: [B,N] points (element strides); idx: [B,N,K1] int32 neighbour lists with the point itself first (K1 = k + 1,
 * as pc3d_knn_f32 returns them for a self-query); out: [B,N] normals (element strides). */
int pc3d_estimate_normal_f32(const float* x, int64_t x_bs, int64_t x_ps, int64_t x_cs, const int32_t* idx,
                             int B, int N, int K1, float* out, int64_t o_bs, int64_t o_ps, int64_t o_cs, void* stream);
/* GeoA3's curvature proxy (attack/GeoA3/loss_utils.py:60-90): kappa[b,i] = mean over the k = K1 - 1 neighbours j
 * (idx[b,i,1:], the first entry is the point itself) of |<(p_j - p_i) / max(|p_j - p_i|, 1e-12), n_i>|, and its
 * backward to the points (gx [B,N,3] contiguous, overwritten; float atomics). x, nrm: [B,N] points / normals with
 * element strides; idx [B,N,K1] int32. One launch each way instead of ~30. */
int pc3d_kappa_f32(const float* x, int64_t x_bs, int64_t x_ps, int64_t x_cs, const float* nrm, int64_t n_bs, int64_t n_ps,
                   int64_t n_cs, const int32_t* idx, int B, int N, int K1, float* out, void* stream);
/* pc3d_kappa_f32 with the normals taken THROUGH an index: point i uses normal nidx[b,i] of the M source normals (nrm
 * [B,M] with element strides; the reference gathers the nearest original point's normal first, loss_utils.py:72-82) and
 * the normals used are written to nout [B,3,N] (contiguous, channels first) — what pc3d_kappa_bwd_f32 then takes. */
int pc3d_kappa_gather_f32(const float* x, int64_t x_bs, int64_t x_ps, int64_t x_cs, const float* nrm, int64_t n_bs,
                          int64_t n_ps, int64_t n_cs, int M, const int64_t* nidx, const int32_t* idx, int B, int N, int K1,
                          float* out, float* nout, void* stream);
int pc3d_kappa_bwd_f32(const float* x, int64_t x_bs, int64_t x_ps, int64_t x_cs, const float* nrm, int64_t n_bs,
                       int64_t n_ps, int64_t n_cs, const int32_t* idx, const float* gout, int B, int N, int K1, float* gx,
                       int deterministic, void* stream);
/* (deterministic = 1: every wavefront accumulates into its own LDS slab and the slabs are combined in a fixed order —
 * bit-identical gradients run to run and for a cloud alone or inside a batch; 0: slabs shared by the wavefronts / global
 * float atomics, order-dependent in the last bits. (3 + 4) N floats of LDS, N <= 8192.) */

/* GeoA3's per-sample loss assembly as one launch each way (attack/GeoA3/GeoA3_attack.py:139-181 on top of
 * loss_utils.py:36-58,92-105): from the adv->ori nearest-neighbour squared distances d_ao [B,N] (and indices idx_ao
 * [B,N] int64), the ori->adv distances d_oa [B,M] (NULL = pseudo-Chamfer), the curvature proxies k_adv [B,N] / k_ori
 * [B,M] (both NULL = no curvature term), the classification loss cls [B] and the trade-off constants scale [B]:
 *   out[0,b] dis  = mean d_ao (+ mean d_oa)     out[1,b] hd = max d_ao     out[2,b] curv = mean (k_adv - k_ori[idx_ao])^2
 *   out[3,b] constrain = w_dis dis + w_hd hd + w_curv curv                 out[4,b] loss_n = cls + scale constrain
 * out [5,B]; hd_arg [B] int32 = first arg-max of d_ao (where torch.max routes the gradient). The backward takes the
 * upstream gradients of all five outputs (g_out [5,B]) and overwrites g_d_ao [B,N], g_d_oa [B,M] (may be NULL),
 * g_k_adv [B,N] (may be NULL) and g_cls [B]. */
int pc3d_geoa3_terms_f32(const float* d_ao, const float* d_oa, const float* k_adv, const float* k_ori,
                         const int64_t* idx_ao, const float* cls, const float* scale, int B, int N, int M, float w_dis,
                         float w_hd, float w_curv, float* out, int32_t* hd_arg, void* stream);
int pc3d_geoa3_terms_bwd_f32(const float* g_out, const float* k_adv, const float* k_ori, const int64_t* idx_ao,
                             const float* scale, const int32_t* hd_arg, int B, int N, int M, float w_dis, float w_hd,
                             float w_curv, float* g_d_ao, float* g_d_oa, float* g_k_adv, float* g_cls, void* stream);

/* Best-attack bookkeeping of one GeoA3 iteration in one launch (attack/GeoA3/GeoA3_attack.py:307-330): label_out[b] =
 * first arg-max of logits[b,:] (NaN counts as the maximum, like torch.argmax); success = (label == target[b]) when
 * targeted, (label != target[b]) otherwise; where success and metric[b] < best_loss[b]: best_loss, best_bs = search_step,
 * best_step = step and best_attack[b,:] = iterate[b,:] (n3 floats per sample, contiguous); where success and metric[b] <
 * iter_best_loss[b]: iter_best_loss and iter_best_score = label. All state is updated in place. */
int pc3d_geoa3_record_f32(const float* logits, int ld, int B, int ncls, const int64_t* target, int targeted,
                          const float* metric, const float* iterate, int n3, int64_t search_step, int64_t step,
                          float* best_loss, float* best_attack, int64_t* best_bs, int64_t* best_step,
                          float* iter_best_loss, int64_t* iter_best_score, int64_t* label_out, void* stream);

/* First layer of a set-abstraction MLP without the grouped input tensor (model/pointnet2_utils.py:118-135,190-197): the
 * layer is linear in [x_j - c_s ; f_j], so  W1 [x_j - c_s ; f_j] + b1 = P[idx[s,j]] + Bc[s]  with P = [x | f] W1^T per
 * POINT (B*NA rows, pc3d_gemm_nt_f32) and Bc[s] = b1 - (Wx x)[centroid s].
 *   pc3d_group_act_f32     : H[b,s,j,:] = act(P[b,idx[b,s,j],:] + Bc[b,s,:])   act = LeakyReLU(slope), slope 0 = ReLU
 *   pc3d_group_act_bwd_f32 : gP[b,idx] += g', gBc[b,s] = sum_j g'  with g' = act'(H) gH (gP is zero-filled here; float
 *                            atomics, ball-query padding merged on chip).
 * P [B,NA,C], Bc [B,S,C], idx [B,S,K] int32 (outside [0,NA): zero row / no gradient), H, gH [B,S,K,C]; C % 4 == 0,
 * C <= 512 for the backward. */
int pc3d_group_act_f32(const float* P, const float* Bc, const int32_t* idx, int B, int NA, int S, int K, int C,
                       float slope, float* H, void* stream);
int pc3d_group_act_bwd_f32(const float* gH, const float* H, const int32_t* idx, int B, int NA, int S, int K, int C,
                           float slope, float* gP, float* gBc, int deterministic, void* stream);
/* (deterministic = 1: gP through the ordered LDS scatter described at pc3d_scatter_rows_det_f32 instead of float atomics) */
/* Layers 1 + 2 of a set-abstraction MLP in one launch, without the [B,S,ns,C1] layer-1 output:
 *   Y[(b,s,j), :] = act(W act_in(P[b, idx[b,s,j], :] + Bc[b,s,:]) + bias)
 * = pc3d_group_act_f32 followed by pc3d_gemm_nt_f32, with the rows of the GEMM's X operand generated on load (P [B*NA, K]
 * with row stride ldp, Bc [B*S, K], idx [B,S,ns] int32, outside [0,NA): zero row of P). act_in = LeakyReLU(slope_in),
 * 0 = ReLU; act / slope as pc3d_gemm_nt_f32. Y [B*S*ns, N] with row stride ldy. mask (may be NULL; needs K % 4 == 0):
 * [B*S*ns, K/4] bytes, bit c % 4 of byte c / 4 = "element c of the generated row had a positive pre-activation" — the
 * only thing the backward of act_in needs from the tensor that is no longer stored:
 *   pc3d_group_act_bwd_mask_f32 = pc3d_group_act_bwd_f32 reading that mask instead of H. */
int pc3d_gemm_nt_gather_f32(const float* P, int64_t ldp, const float* Bc, const int32_t* idx, int B, int NA, int S, int ns,
                            float slope_in, const float* W, const float* bias, int N, int K, int act, float slope, float* Y,
                            int64_t ldy, uint8_t* mask, uint32_t* ymask, void* stream);
/* pc3d_group_max_linear_bwd_f32 with the channels split over `ksplit` (1 or 4) thread groups of a workgroup, the partial
 * sums added in chunk order: for launches with few groups (a group-all layer: ONE group per cloud, C3 = 1024 channels to
 * walk serially). A row's summation order depends on ksplit, so the caller chooses it from the per-cloud shape, never
 * from the batch. C3 % 32 == 0. */
int pc3d_group_max_linear_bwd_ks_f32(const float* gout, const float* out, const int64_t* arg, const float* W, int G, int ns,
                                     int C2, int C3, const float* xin, float* gx, int ksplit, void* stream);
/* ymask (may be NULL; needs N % 32 == 0): [B*S*ns, N/32] words, bit n % 32 of word n / 32 of a row = "Y[row, n] > 0 before
 * the activation" — the sign of layer 2's output for ITS backward, so that Y itself need not be kept after the next
 * layer has consumed it:  pc3d_group_max_linear_bwd_mask_f32 = pc3d_group_max_linear_bwd_f32 with that mask for xin. */
int pc3d_group_max_linear_bwd_mask_f32(const float* gout, const float* out, const int64_t* arg, const float* W, int G, int ns,
                                       int C2, int C3, const uint32_t* xmask, float* gx, void* stream);
int pc3d_group_act_bwd_mask_f32(const float* gH, const uint8_t* mask, const int32_t* idx, int B, int NA, int S, int K, int C,
                                float slope, float* gP, float* gBc, int deterministic, void* stream);

/* The WHOLE set-abstraction MLP after its per-point first layer + the max over the group in ONE launch
 * (model/pointnet2_utils.py:173-199): out[g,c] = relu(max_j (W3 relu(W2 relu(P[b, idx[g,j], :] + Bc[g,:]) + b2))[c] + b3[c]),
 * arg[g,c] = the winning member j (lowest on ties) — pc3d_gemm_nt_gather_f32 followed by the group-max form of the last
 * layer, without the [B*S*ns, C2] layer-2 output in memory (268 MB per level at SSG's sizes). Results are bit-identical
 * to that two-launch form (same operand order, same fp32 MFMA sequence). P [B*NA, C1] row stride ldp, Bc [B*S, C1],
 * idx [B,S,ns] int32 (outside [0,NA): zero row of P), W2 [C2,C1], W3 [C3,C2]; ns in {32, 64, 128}; C1 in {32, 64, 128},
 * C2 a multiple of 32 up to 128, C3 a multiple of 32. mask1 [B*S*ns, C1/4] bytes and mask2 [B*S*ns, C2/32] words receive the sign bits
 * of the generated layer-1 rows / of the layer-2 pre-activations (as pc3d_gemm_nt_gather_f32 writes them): all the
 * backward (pc3d_group_max_linear_bwd_mask_f32, pc3d_gemm_nt_f32 on W2^T, pc3d_group_act_bwd_*) needs. */
int pc3d_sa_chain_f32(const float* P, int64_t ldp, const float* Bc, const int32_t* idx, int B, int NA, int S, int ns,
                      const float* W2, const float* b2, int C1, int C2, const float* W3, const float* b3, int C3,
                      uint8_t* mask1, uint32_t* mask2, float* out, int64_t* arg, void* stream);
/* The same launch over a TABLE of row units: the 16- / 32-row units of the grouped rows that hold nothing but the ball
 * query's padding (copies of a group's first point, model/pointnet2_utils.py:84-104 — they can neither change nor win the
 * group max) are left out and the others packed four to a tile, whole groups per tile: fewer tiles instead of wasted
 * products (SSG: 13.5 of 32 and 22 of 64 rows per group are listed points). Same out / arg; mask1 / mask2 are written for
 * the kept units only (the backward never reads the others: their rows carry exact zeros).
 *   pc3d_sa_chain_table_unit: rows per unit the launch expects for a shape — 16 (the resident-weight kernel, 64-row tiles),
 *   8 (the streaming kernel, groups of <= 64 rows: an 8-row quarter of a 32 x 32 MFMA accumulator tile is a register group
 *   of its own, so the max is taken per 8 rows), 32 (the streaming kernel, groups of 128 rows), 0 (no table).
 *   pc3d_sa_blocks_i32: builds the table from idx [B,S,ns] — flags [B*S] bytes (scratch: bit u = unit u of the group is
 *   kept), tb int32 unit ids in original row space (-1 = empty slot): unit 8 -> [B*S*8] (64-row tiles of eight slots, at
 *   most eight units per group), unit 16 -> [B*S*4] (64-row tiles of four slots), unit 32 -> [ceil(B*S*ns/128)*4]
 *   (128-row tiles of four slots; at most four units per group for 16 / 32); ntiles [1] — two small launches (idx is known
 *   as soon as the ball query has run). Whole groups per tile; tiles past ntiles are undefined for units 8 / 16. B*S <= 64 K
 *   groups (48 K for unit 32): beyond that the entry point fails and the caller launches pc3d_sa_chain_f32 over every row.
 *   pc3d_sa_chain_tb_f32: tb / ntiles NULL = pc3d_sa_chain_f32; `unit` must be what pc3d_sa_chain_table_unit says. */
int pc3d_sa_chain_table_unit(int S, int ns, int C1, int C2, int C3);
int pc3d_sa_blocks_i32(const int32_t* idx, int B, int S, int ns, int unit, uint8_t* flags, int32_t* tb, int32_t* ntiles,
                       void* stream);
int pc3d_sa_chain_tb_f32(const float* P, int64_t ldp, const float* Bc, const int32_t* idx, int B, int NA, int S, int ns,
                         const float* W2, const float* b2, int C1, int C2, const float* W3, const float* b3, int C3,
                         uint8_t* mask1, uint32_t* mask2, float* out, int64_t* arg, const int32_t* tb, const int32_t* ntiles,
                         int unit, void* stream);

/* The backward of a set-abstraction chain (pc3d_sa_chain_f32) from the max down to the points, on the ACTIVE rows only.
 * A row of a group that wins no channel of the max carries no gradient — 58-65 % of the rows at the single-scale
 * classifier's levels (the ball query pads a group with copies of its first point, and a copy never wins): the three
 * launches below neither write, read nor multiply those rows (model/pointnet2_utils.py:190-198 differentiated).
 *   pc3d_group_max_linear_bwd_sparse_f32 = pc3d_group_max_linear_bwd_mask_f32 that also writes amask [G, ceil(ns/32)] words
 *       (bit j of group g = "row j won at least one channel") and stores ONLY those rows of gx [G*ns, C2]; the other
 *       rows of gx stay undefined.
 *   pc3d_gemm_nt_groupsum_f32: Y = X . Wt^T (pc3d_gemm_nt_f32 without bias / activation) AND the groups pass of
 *       pc3d_group_act_bwd_rev_f32 on Y in one launch — X = gx above [B*S*ns, K] (row stride ldx), Wt [C1, K] = W2
 *       transposed, Y [B*S*ns, C1] = the gradient of the generated layer-1 rows (NOT masked by mask1; written: the active
 *       rows, and zeros for inactive rows that are not padding copies j > 0 with idx[g,j] == idx[g,0] — those copies are
 *       not in the reverse lists of pc3d_group_reverse_i32, the points pass reaches them through `tail` only), gBc [B*S, C1] = per group the sum of mask1 * Y over its rows, tail [B*S, C1] the same sum over the rows
 *       j > 0 that repeat the group's first index (idx [B,S,ns]); mask1 [B*S*ns, C1/4] bytes from the forward. amask NULL =
 *       every row is active. A tile's active rows are compacted before the product (half the MFMA blocks at those
 *       levels). Bit-identical to the separate launches on full tensors: the skipped terms are exact zeros, the others
 *       are summed in the same order. ns in {32, 64, 128}; C1 in {32, 64, 128}; K a multiple of 32.
 *   pc3d_group_act_bwd_points_f32: the points pass of pc3d_group_act_bwd_rev_f32 alone (gP [B,NA,C] from gH, its sign
 *       bytes, the groups pass's tail and the reverse index off / lst); amask (may be NULL) [B*S, ceil(K/32)]: rows of gH
 *       whose bit is clear are skipped. */
int pc3d_group_max_linear_bwd_sparse_f32(const float* gout, const float* out, const int64_t* arg, const float* W, int G, int ns,
                                         int C2, int C3, const uint32_t* xmask, float* gx, uint32_t* amask, void* stream);
int pc3d_gemm_nt_groupsum_f32(const float* X, int64_t ldx, const float* Wt, const uint8_t* mask1, const int32_t* idx,
                              const uint32_t* amask, int B, int S, int ns, int C1, int K, float* Y, float* gBc, float* tail,
                              void* stream);
/* pc3d_gemm_nt_groupsum_f32 with amask over PACKED tiles: a tile is a run of whole groups whose active rows fill 128 rows
 * (a first one-workgroup launch assigns groups to tiles from amask's bit counts), instead of 128 consecutive rows of
 * which a third are active. Same results bit for bit. scratch: B*S + 4 + ceil(B*S / 4) int32 (overwritten). ns in {32, 64};
 * amask must not be NULL. */
int pc3d_gemm_nt_groupsum_packed_f32(const float* X, int64_t ldx, const float* Wt, const uint8_t* mask1, const int32_t* idx,
                                     const uint32_t* amask, int B, int S, int ns, int C1, int K, float* Y, float* gBc,
                                     float* tail, int32_t* scratch, void* stream);
int pc3d_group_act_bwd_points_f32(const float* gH, const uint8_t* mask, const float* tail, const int32_t* off,
                                  const int32_t* lst, const uint32_t* amask, int B, int NA, int S, int K, int C, float slope,
                                  float* gP, void* stream);

/* The coordinate part of a set-abstraction layer's first 1x1 convolution (model/pointnet2_utils.py:118-135,190-197: the
 * Conv2d over [xyz_j - centre_s ; feat_j] is linear, so it splits into a per-POINT part P = Wx x + Wf f and a per-CENTRE
 * part Bc = b1 - Wx c; the Wf part is a GEMM with the coordinate part as its residual operand). Three-column products:
 *   pc3d_affine3_f32      out[b,n,c] = bias[c] + sign * (W[c,0] x + W[c,1] y + W[c,2] z);  x a [B,N,3] view with element
 *                         strides (x_bs, x_ps, x_cs) — a channels-first [B,3,N] tensor is read in place; W [C,3], bias [C]
 *                         or NULL, C % 4 == 0, out [B*N, C] with row stride ldo.
 *   pc3d_affine3_bwd_f32  out[b,n,:] = add[b,n,:] + sign * sum_c g[b,n,c] W[c,:];  g [B*N, C] row stride ldg, add (may be
 *                         NULL) and out [B,N,3] views with their own strides (the gradient of a channels-first tensor is
 *                         written in that layout). Fixed summation order.
 *   pc3d_scatter_points_det_f32  out[b, idx[b,s], :] += val[b,s,:] in ascending s: the gradient of the centres
 *                         new_xyz = xyz[fps_idx] (:113, index_points) folded into the gradient of xyz; idx [B,S] int32,
 *                         val [B,S,3] contiguous, out as above. One wavefront owns a cloud's column: no float atomics on
 *                         memory, repeated centres are summed in order. */
int pc3d_affine3_f32(const float* x, int64_t x_bs, int64_t x_ps, int64_t x_cs, int B, int N, const float* W,
                     const float* bias, float sign, int C, float* out, int64_t ldo, void* stream);
int pc3d_affine3_bwd_f32(const float* g, int64_t ldg, int B, int N, int C, const float* W, float sign,
                         const float* add, int64_t a_bs, int64_t a_ps, int64_t a_cs,
                         float* out, int64_t o_bs, int64_t o_ps, int64_t o_cs, void* stream);
int pc3d_scatter_points_det_f32(const int32_t* idx, const float* val, int B, int S, int N, float* out, int64_t o_bs,
                                int64_t o_ps, int64_t o_cs, void* stream);
/* out[g,c] = max_r Y[g,r,c], arg[g,c] = the first row holding it (a NaN wins, as torch.max): the group max of a
 * set-abstraction layer whose last layer is too wide for the fused form (model/pointnet2_utils.py:198). Y [G,ns,C]. */
int pc3d_rows_max_f32(const float* Y, int64_t G, int ns, int C, float* out, int64_t* arg, void* stream);

/* The same backward WITHOUT float atomics, as a gather over a reverse index of the grouping.
 * pc3d_group_reverse_i32: per cloud, the list of grouped rows that reference each point (a counting sort on the device:
 *   two passes of integer atomics + a scan). idx [B,S,K] int32 as above; cnt [B,NA] int32 scratch; off [B,NA+1] int32
 *   (list of point p = lst[b][off[b,p] .. off[b,p+1])); lst [B, L] int32 with L = pc3d_group_reverse_list_len(S, K) =
 *   S*K + S. Entry codes: s*K + j = row (s,j) of the cloud; S*K + s = "the sum of the rows of group s that repeat its
 *   first index" (the ball query's padding), listed under that first index. Entries outside [0,NA) are in no list.
 *   Every list is sorted ascending after the fill (the atomics hand out slots in arrival order), so the gather below
 *   sums in ONE order: deterministic gradients. It depends on idx only: build it once per forward, beside the MLPs.
 * pc3d_group_act_bwd_rev_f32: gBc and the per-group padded-tail sums in one pass over the groups (tail [B,S,C] scratch),
 *   then gP[b,p,:] = sum over the list of p — every row of gP written once, no zero fill. The sign of the activation
 *   comes from H [B,S,K,C] or, when H is NULL, from the bit mask of pc3d_gemm_nt_gather_f32. C % 4 == 0, C <= 512. */
int64_t pc3d_group_reverse_list_len(int S, int K);
int pc3d_group_reverse_i32(const int32_t* idx, int B, int NA, int S, int K, int32_t* cnt, int32_t* off, int32_t* lst,
                           void* stream);
int pc3d_group_act_bwd_rev_f32(const float* gH, const float* H, const uint8_t* mask, const int32_t* idx, const int32_t* off,
                               const int32_t* lst, int B, int NA, int S, int K, int C, float slope, float* gP, float* gBc,
                               float* tail, void* stream);

/* Backward of  [max_n | mean_n] LeakyReLU(Y[b,n,:], slope)  followed straight by the backward of the layer that produced Y
 * (model/dgcnn.py:317-320 after conv5, model/curvenet.py:64-67 after conv0), in ONE GEMM: dX = dY W with the rows of dY
 * generated on load from the pre-activation Y [B*Npts, K] (row stride ldy_in), the upstream gradient g [B, 2K] (max part,
 * then mean part) and the arg-max rows arg [B, K] of pc3d_act_pool_f32 — pc3d_act_pool_bwd_f32 + pc3d_gemm_nt_f32 without
 * the [B*Npts, K] gradient tensor. W [N, K] row-major (the producing layer's weight TRANSPOSED, as for any backward on
 * pc3d_gemm_nt_f32); dX [B*Npts, N] with row stride ldx_out. K % 4 == 0. */
int pc3d_gemm_nt_poolbwd_f32(const float* Y, int64_t ldy_in, const float* g, const int32_t* arg, int B, int Npts,
                             float slope_pool, const float* W, int N, int K, float* dX, int64_t ldx_out, void* stream);

/* Row reductions of a [B,N] f32 matrix into out[B].
 *   op:  0 = mean, 1 = max, 2 = sum        pre: 0 = identity, 1 = sqrt(max(x,0)) applied per element first
 * mean/max of squared distances = ChamferDistance/HausdorffDistance (distance.py:44-49,64-69);
 * pre=1 gives the Euclidean variants of dis_utils_{numpy,torch}.py. Deterministic (fixed tree order). */
int pc3d_rowreduce_f32(const float* x, int B, int N, int op, int pre, float* out, void* stream);

/* Backward through the gathered NN pairs of pc3d_nn_bidir_f32 (autograd of distance.py:40-50,58-70 and of
 * every loss built on per-point NN distances).  With upstream per-point weights wA on dA and wB on dB
 * (element strides (w_bs, w_ps) — a stride of 0 broadcasts, so "mean over points" needs no expanded tensor —
 * and scalar multipliers sA, sB):
 *   grad_a[i] = 2 sA wA[i] (a_i - b_iA[i]) + sum_{j: iB[j]==i} 2 sB wB[j] (a_i - b_j)
 *   grad_b[j] = 2 sB wB[j] (b_j - a_iB[j]) + sum_{i: iA[i]==j} 2 sA wA[i] (b_j - a_i)
 * wA/wB may be NULL (that direction contributes nothing). grad_a/grad_b are OVERWRITTEN; either may be NULL.
 * deterministic != 0: ordered accumulation (bitwise reproducible) instead of float atomics. */
int pc3d_nn_bwd_f32(const float* a, int64_t a_bs, int64_t a_ps, int64_t a_cs,
                    const float* b, int64_t b_bs, int64_t b_ps, int64_t b_cs,
                    int B, int N, int M,
                    const int32_t* iA, const float* wA, int64_t wA_bs, int64_t wA_ps, float sA,
                    const int32_t* iB, const float* wB, int64_t wB_bs, int64_t wB_ps, float sB,
                    float* grad_a, int64_t ga_bs, int64_t ga_ps, int64_t ga_cs,
                    float* grad_b, int64_t gb_bs, int64_t gb_ps, int64_t gb_cs,
                    int deterministic, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * K8  PointNet per-point MLP 3 -> 64 -> 128 -> C3 (1x1 conv, eval-mode BatchNorm folded into W/b by the caller,
 * ReLU after layers 1-2 and optionally after the pool) fused with max-pool over the N points.
 * Replaces model/pointnet.py:34-37 (STN3d tower, relu_last=1) and :110-123 (PointNetfeat trunk, relu_last=0),
 * i.e. three Conv1d+BN(+ReLU) and torch.max(x, 2) with their [B,C,N] activations.
 *   x      B x N points (element strides); T optional [B,3,3] row-major input transform, x'[n,:] = x[n,:] @ T[b]
 *          (the torch.bmm of model/pointnet.py:106-109), NULL = identity
 *   W1 [64,3] b1[64]  W2 [128,64] b2[128]  W3 [C3,128] b3[C3]   row-major, BN already folded; C3 % 32 == 0
 *   part_val/part_idx  caller workspace [B, ceil(N / pc3d_pointmlp3_tile_points()), C3] f32 / i32
 *   pooled [B,C3] f32 (after the optional ReLU), argidx [B,C3] i32 = lowest point index attaining the max;
 *          pass both NULL to keep only the per-tile partials (no fold launch)
 *   mask1 [B,N] u64, mask2 [B,N,4] u32 (both or neither)  the ReLU decisions of layers 1 and 2 per point (bit c of
 *          mask1 = channel c of layer 1 is positive; word j bit r of mask2 = channel 32j+r of layer 2 is positive),
 *          which the backward launch consumes instead of recomputing the two layers
 * fp32 throughout: layer 2/3 run on v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains).
 * ------------------------------------------------------------------------------------------------------- */
int pc3d_pointmlp3_tile_points(void);
int pc3d_pointmlp3_bwd_tile_points(void);
int pc3d_pointmlp3_max_fwd_f32(const float* x, int64_t x_bs, int64_t x_ps, int64_t x_cs, int B, int N,
                               const float* T, const float* W1, const float* b1, const float* W2,
                               const float* b2, const float* W3, const float* b3, int C1, int C2, int C3,
                               int relu_last, float* part_val, int32_t* part_idx,
                               float* pooled, int32_t* argidx, uint64_t* mask1, uint32_t* mask2, void* stream);
/* The same launch with the input transform computed in its prologue: T[b] = th_W [9,th_K] . th_in[b] + th_b — the
 * last layer of STN3d (model/pointnet.py:45-47: fc3, with the flattened identity added into th_b) — instead of a
 * launch of its own between the two towers; T_out [B,9] receives it (the backward launch reads it as its T). */
int pc3d_pointmlp3_max_fwd_th_f32(const float* x, int64_t x_bs, int64_t x_ps, int64_t x_cs, int B, int N,
                                  const float* th_in, const float* th_W, const float* th_b, int th_K, float* T_out,
                                  const float* W1, const float* b1, const float* W2, const float* b2, const float* W3,
                                  const float* b3, int C1, int C2, int C3, int relu_last, float* part_val,
                                  int32_t* part_idx, float* pooled, int32_t* argidx, uint64_t* mask1, uint32_t* mask2,
                                  void* stream);

/* Backward-to-input of the above (weights are frozen during an attack: no weight gradients, SURVEY A-14).
 * g_pooled [B,C3] is the upstream gradient on `pooled`; with relu_last the caller zeroes it where pooled <= 0.
 * T == NULL: grad_x receives the gradient w.r.t. the tower input. T given: the kernel chains through x' = x @ T:
 * grad_x receives the gradient w.r.t. the RAW points x, and part_gT (if non-NULL, workspace
 * [B, ceil(N / pc3d_pointmlp3_bwd_tile_points()), 16], 9 used) the per-tile partial sums of dL/dT (row-major 3x3);
 * their sum over tiles is the gradient that flows on into the STN head. accumulate != 0: grad_x += (used to add the
 * STN tower's contribution on top of the trunk's).
 * W2T is W2 transposed ([64,128] row-major; lets the W2^T product read its operand rows contiguously).
 * mask1 / mask2 are the forward launch's outputs of those names (required).
 * The max-pool routes each channel to one point, so the layer-3 dgrad is a sparse ordered gather: deterministic. */
int pc3d_pointmlp3_max_bwd_f32(const float* x, int64_t x_bs, int64_t x_ps, int64_t x_cs, int B, int N,
                               const float* T, const float* W1, const float* b1, const float* W2,
                               const float* b2, const float* W3, const float* W2T, int C1, int C2, int C3,
                               const int32_t* argidx, const uint64_t* mask1, const uint32_t* mask2,
                               const float* g_pooled,
                               float* grad_x, int64_t gx_bs, int64_t gx_ps, int64_t gx_cs,
                               float* part_gT, int accumulate, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Classifier heads. Y[b,o] = epi(sum_k X[b,k] W[o,k] + bias[o]) for small row counts (the B samples of a batch),
 * fp32 MFMA. X may be given as P partial slabs per row ([B,P,K], summed on load; P=1 = plain matrix; ldx = row
 * stride in floats). epi: optional (Leaky)ReLU (relu = 1: Y = Y > 0 ? Y : slope * Y; slope 0 = ReLU), then optional
 * gate (Y = gate[b,o] > 0 ? Y : gate_slope * Y — the activation mask of a saved forward output, which makes the same
 * kernel the backward of Linear + (Leaky)ReLU when W is passed transposed: dX_prev = gate_prev(dY . W)).
 * Replaces F.linear / BatchNorm1d(eval, folded) / ReLU of model/pointnet.py:38-47,144-147 (LeakyReLU(0.2): the DGCNN
 * head, model/dgcnn.py:322-326) and their autograd.
 * ------------------------------------------------------------------------------------------------------- */
int pc3d_linear_f32(const float* X, int ldx, int P, int B, int K, const float* W, const float* bias, int O,
                    int relu, float slope, const float* gate, int ldg, float gate_slope, float* Y, int ldy, void* stream);
/* The same layer with its X operand produced on the fly by a tiny pre-layer (the backward of STN3d's fc3, 9 -> 256,
 * folded into the launch of fc2's backward, model/pointnet.py:44-45 and their autograd):
 *   X[b,k] = gate_pre[b,k] > 0 ? sum_{j<J} (sum_p parts[b,p,j]) Wp[j,k] : 0 ;   Y[b,o] = gate(sum_k X[b,k] W[o,k])
 * parts [B,P,Jp] (J <= Jp columns used, J <= 16), Wp [J,K] row-major, gate_pre [B,K] (row stride ldgp), K % 16 == 0. */
int pc3d_linear_pre_f32(const float* parts, int P, int Jp, int J, const float* Wp, const float* gate_pre, int ldgp, int B,
                        int K, const float* W, int O, const float* gate, int ldg, float* Y, int ldy, void* stream);

/* log_softmax (model/pointnet.py:148) + argmax + adversarial loss on the log-probabilities and its gradient w.r.t.
 * the LOGITS, one launch. kind 0 = UntargetedLogitsAdvLoss, 1 = LogitsAdvLoss, 2 = CrossEntropyAdvLoss, 3 = minus kind 2
 * (GeoA3's untargeted classification loss, attack/GeoA3/GeoA3_attack.py:125-127)
 * (attack/CW/CW_utils/adv_utils.py:64-80, 17-33, 42-51); per-sample loss[b] (before the batch mean), pred[b],
 * logp [B,ncls], g_logits [B,ncls] = scale * dloss_b/dlogits (scale = 1/B reproduces .mean()). Outputs may be NULL.
 * kind + 4: the loss is taken on `logits` AS GIVEN (no log-softmax) — what the reference's functors compute on whatever the
 * victim returns, log-probabilities (PointNet, PointNet++, DGCNN) or raw logits (CurveNet); kind 6 is then nll_loss. */
int pc3d_cls_loss_f32(const float* logits, int ld, int B, int ncls, const int64_t* target, int kind, float kappa,
                      float scale, float* logp, int64_t* pred, float* loss, float* g_logits, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * K9  clip / projection of the perturbation pc - ori, one launch (attack/CW/CW_utils/clip_utils.py).
 *   mode 0: per-point L2 norm <= budget  — ClipPointsLinf (:43-56; the reference's "Linf" is a per-point L2 clip,
 *           SURVEY A-10); with `normal` != NULL the inner-point tangent projection ProjectInnerPoints (:67-108)
 *           runs first = ProjectInnerClipLinf (:124-136). budget <= 0 skips the clip (projection only).
 *   mode 1: global L2 norm of the whole perturbation <= budget — ClipPointsL2 (:16-29). `normal` ignored.
 * All tensors B x K points with element strides; out may alias pc.
 * ------------------------------------------------------------------------------------------------------- */
int pc3d_clip_f32(const float* pc, int64_t pc_bs, int64_t pc_ps, int64_t pc_cs,
                  const float* ori, int64_t o_bs, int64_t o_ps, int64_t o_cs,
                  const float* normal, int64_t n_bs, int64_t n_ps, int64_t n_cs,
                  int B, int K, int mode, float budget,
                  float* out, int64_t out_bs, int64_t out_ps, int64_t out_cs, void* stream);

/* K10 (+K9)  one torch.optim.Adam step (weight_decay 0, amsgrad off) on the adversarial points, optionally fused
 * with the mode-0 clip/projection against `ori` (ori NULL = plain Adam). Replaces opt.step() + clip_func of
 * attack/CW/CW_attack.py:169-174 and attack/KNN/KNN_attack.py:129-136.
 * m, v (exp_avg, exp_avg_sq) share p's strides and are updated in place. The step number t (>= 1) comes from the
 * device word *step_dev when non-NULL (hipGraph replay), else from step_host. g2 (may be NULL): a second gradient,
 * summed with g on load — the victim's and the distance term's branches of loss.backward() (attack/KNN/KNN_attack.py:
 * 125-129), which autograd would add in a launch of its own. */
int pc3d_adam_clip_step_f32(float* p, int64_t p_bs, int64_t p_ps, int64_t p_cs,
                            const float* g, int64_t g_bs, int64_t g_ps, int64_t g_cs,
                            const float* g2, int64_t g2_bs, int64_t g2_ps, int64_t g2_cs,
                            float* m, float* v,
                            const float* ori, int64_t o_bs, int64_t o_ps, int64_t o_cs,
                            const float* normal, int64_t n_bs, int64_t n_ps, int64_t n_cs,
                            int B, int K, double lr, double beta1, double beta2, double eps, float budget,
                            const int32_t* step_dev, int step_host, void* stream);

/* *ctr += delta on the stream (advances the Adam step word between graph replays). */
int pc3d_i32_add(int32_t* ctr, int delta, void* stream);

/* Dense pairwise distances out[b,i,j] (contiguous [B,N,M] f32): mode 0 = |x_i-y_j|^2, mode 1 = Euclidean.
 * For callers that really consume the matrix: _Distance.batch_pairwise_dist (distance.py:15-32),
 * dis_utils_torch.pairwise_distances (:8-11), dis_utils_numpy.pairwise_distances (:13-20),
 * pointnet2_utils.square_distance (model/pointnet2_utils.py:19-38). Direct-difference form. */
int pc3d_pairwise_f32(const float* x, int64_t x_bs, int64_t x_ps, int64_t x_cs,
                      const float* y, int64_t y_bs, int64_t y_ps, int64_t y_cs,
                      int B, int N, int M, int mode, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * K2/K4  K nearest reference points (xyz, squared L2, direct-difference form) of every query, ascending;
 * ties put the lower reference index first. 1 <= K <= min(64, M). dists/idx: [B,N,K] f32 / i32 (either may be NULL).
 * Replaces the [B,N,M] matrix + topk of attack/CW/CW_utils/dist_utils.py:133-144 (KNNDist),
 * attack/GeoA3/knn_utils.py:10-55 (knn_points), attack/AOF/TAOF_attack.py:13-28, model/dgcnn.py:194-200 on xyz,
 * model/curvenet_util.py:10-17. Self-kNN (q == r) returns the point itself first (distance 0), like the reference.
 * ------------------------------------------------------------------------------------------------------- */
int pc3d_knn_f32(const float* q, int64_t q_bs, int64_t q_ps, int64_t q_cs,
                 const float* r, int64_t r_bs, int64_t r_ps, int64_t r_cs,
                 int B, int N, int M, int K, float* dists, int32_t* idx, void* stream);
/* The neighbour graph of ONE cloud set (self-kNN, K = k + 1 columns, self first) with the two views of it that
 * CurveNet's blocks use (model/curvenet_util.py:10-17 `knn`, :232 `idx[:, :, :k]`, walk.py `idx[:, :, 1:]`) written by the
 * same launch instead of two slicing copies per resolution: idx [B,N,K]; idx_noself [B,N,K-1] (may be NULL) = columns
 * 1 .. K-1; idx_first [B,N,k2] (may be NULL, k2 <= K) = columns 0 .. k2-1. Same search and tie rule as pc3d_knn_f32. */
int pc3d_knn_graph_i32(const float* pts, int64_t p_bs, int64_t p_ps, int64_t p_cs, int B, int N, int K, int32_t* idx,
                       int32_t* idx_noself, int32_t* idx_first, int k2, void* stream);
/* The same two searches with a HINT: hint [B,N,K] int32 (may be NULL, may be `idx` itself — a wavefront reads the rows of
 * its queries before it writes them) holds K reference indices per query from an earlier search of nearly the same clouds:
 * the previous iteration of an attack loop (the reference re-runs its matrix + topk from scratch every iteration,
 * model/curvenet_util.py:10-17, attack/CW/CW_utils/dist_utils.py:133-144). Their largest distance bounds the K-th
 * distance, so only candidates at or below it are ever inserted (K + a few instead of K ln(M / K) per query). The RESULT
 * does not depend on the hint — distances, indices and tie order are those of pc3d_knn_f32; a hint row that is out of
 * range, repeats an index or has a non-finite bound makes its wavefront run the unhinted scan. */
int pc3d_knn_hint_f32(const float* q, int64_t q_bs, int64_t q_ps, int64_t q_cs,
                      const float* r, int64_t r_bs, int64_t r_ps, int64_t r_cs,
                      int B, int N, int M, int K, float* dists, int32_t* idx, const int32_t* hint, void* stream);
int pc3d_knn_graph_hint_i32(const float* pts, int64_t p_bs, int64_t p_ps, int64_t p_cs, int B, int N, int K, int32_t* idx,
                            int32_t* idx_noself, int32_t* idx_first, int k2, const int32_t* hint, void* stream);


/* Backward of the K distances with upstream w [B,N,K]: grad_q dense, grad_r scattered (float atomics; when
 * deterministic != 0 in a fixed order: with det_ws = B*N*K*3 floats of scratch the edges' contributions are recorded
 * and summed by the ordered LDS scatter of pc3d_scatter_rows_det_f32, with det_ws NULL every reference point scans
 * the whole index list). Both are OVERWRITTEN; either may be NULL. When q and r are the same cloud (self-kNN) the
 * caller adds the two results. */
int pc3d_knn_bwd_f32(const float* q, int64_t q_bs, int64_t q_ps, int64_t q_cs,
                     const float* r, int64_t r_bs, int64_t r_ps, int64_t r_cs,
                     int B, int N, int M, int K, const int32_t* idx, const float* w,
                     float* grad_q, int64_t gq_bs, int64_t gq_ps, int64_t gq_cs,
                     float* grad_r, int64_t gr_bs, int64_t gr_ps, int64_t gr_cs,
                     int deterministic, float* det_ws, void* stream);
/* Self-kNN form of that backward (q and r are the SAME cloud x, the kNN attack's regulariser): grad [B,N,3 strides] =
 * the dense term + the scattered term in one buffer; w_scale [B] (may be NULL) multiplies every w[b,...] — an upstream
 * per-sample gradient folded into the launch. deterministic / det_ws as above. */
int pc3d_knn_self_bwd_f32(const float* x, int64_t x_bs, int64_t x_ps, int64_t x_cs, int B, int N, int K,
                          const int32_t* idx, const float* w, const float* w_scale, float* grad, int64_t g_bs, int64_t g_ps,
                          int64_t g_cs, int deterministic, float* det_ws, void* stream);
/* The kNN-distance outlier penalty of the kNN attack (attack/CW/CW_utils/dist_utils.py:112-160 KNNDist.forward) from the
 * K1 = k + 1 sorted self-kNN distances d [B,N,K1] (entry 0 = the point itself): value_i = mean_{j>=1} d[i,j]; thr =
 * mean_i value + alpha * std_i value (unbiased); loss[b] = mean_i value_i [value_i > thr]; and the weights its backward
 * hands to pc3d_knn_self_bwd_f32: w[b,i,j] = [value_i > thr] / (N k) for j >= 1, 0 for j = 0 (the mask is a constant of
 * the graph, :146-151). One workgroup per sample, fixed-order reductions. */
int pc3d_knn_outlier_loss_f32(const float* d, int B, int N, int K1, float alpha, float* loss, float* w, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * K11  per-iteration bookkeeping of the CW-family loops on the device (attack/CW/CW_attack.py:129-153): per-sample
 * dist = ||adv-ori||_F, success = (pred != label) [untarget] or (pred == label); where success && dist < bestdist
 * update (bestdist, bestscore); where success && dist < o_bestdist update (o_bestdist, o_bestscore) and copy adv[b]
 * into o_bestattack[b]. input_val (may be NULL) always receives adv (the iterate this pass started from, :133).
 * dist_val[B] (may be NULL) receives dist; *step (may be NULL) is incremented by one (the Adam step word).
 * Replaces the D2H copy of the whole cloud + Python per-sample loop of every iteration.
 * ------------------------------------------------------------------------------------------------------- */
int pc3d_cw_bookkeep_f32(const float* adv, int64_t a_bs, int64_t a_ps, int64_t a_cs,
                         const float* ori, int64_t o_bs, int64_t o_ps, int64_t o_cs, int B, int K,
                         const int64_t* pred, const int64_t* label, int untarget,
                         float* bestdist, int64_t* bestscore, float* o_bestdist, int64_t* o_bestscore,
                         float* o_bestattack, int64_t ba_bs, int64_t ba_ps, int64_t ba_cs,
                         float* input_val, int64_t iv_bs, int64_t iv_ps, int64_t iv_cs,
                         float* dist_val, int32_t* step, void* stream);

/* CW update in one launch: g = g_model + d/dadv[ mean_b w[b] * D(adv_b, ori_b) ]; Adam (as pc3d_adam_clip_step_f32);
 * per-point clip to `budget` (ClipPointsLinf; <= 0: none). dist_kind 0: none; 1: L2Dist (needs l2norm[B] =
 * ||adv-ori||_F, e.g. dist_val of the bookkeeping launch); 2: ChamferDist('adv2ori') (needs nn_idx [B,K] from
 * pc3d_nn_f32). Replaces loss.backward() of the distance term + opt.step() + clip_func (CW_attack.py:161-174). */
int pc3d_cw_step_f32(float* p, int64_t p_bs, int64_t p_ps, int64_t p_cs,
                     const float* g, int64_t g_bs, int64_t g_ps, int64_t g_cs, float* m, float* v,
                     const float* ori, int64_t o_bs, int64_t o_ps, int64_t o_cs, int B, int K,
                     double lr, double beta1, double beta2, double eps, float budget,
                     const int32_t* step_dev, int step_host, int dist_kind, const float* w,
                     const float* l2norm, const int32_t* nn_idx, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * K5  farthest-point sampling (model/pointnet2_utils.py:60-81; model/curvenet_util.py:69-90 with start 0).
 * One workgroup per cloud performs all S dependent arg-max steps on chip. start[B] = first sampled index per
 * cloud (the reference draws it with torch.randint every forward, SURVEY A-4; NULL = 0). out [B,S] i32.
 * Same fp32 arithmetic as the reference (((dx*dx+dy*dy)+dz*dz), running min initialised to 1e10, lowest index on
 * ties) so the index sequence is reproduced bit for bit. N <= 8192.
 * ------------------------------------------------------------------------------------------------------- */
int pc3d_fps_f32(const float* xyz, int64_t x_bs, int64_t x_ps, int64_t x_cs, int B, int N, int S,
                 const int32_t* start, int32_t* out, void* stream);
/* The same sampling with the workgroup size named (64, 128, 256, 512 or 1024 threads per cloud, N <= 32 * threads; results
 * are identical; pc3d_fps_f32 takes 64 up to N = 512, else 256): for tests and measurements. */
int pc3d_fps_threads_f32(int threads, const float* xyz, int64_t x_bs, int64_t x_ps, int64_t x_cs, int B, int N, int S,
                         const int32_t* start, int32_t* out, void* stream);
/* The same sampling on ONE wavefront per cloud with exact pruning (csrc/fps_pruned.hip): the cloud is split into rows of 64
 * spatially close points (a k-d partition built inside the launch), and a step updates only the rows whose bounding box is
 * closer to the new centre than the row's largest running distance. Same arithmetic per point, ties to the lowest index:
 * the index sequence equals pc3d_fps_threads_f32's bit for bit. N <= 4096. Measured no faster than the full-update kernel
 * (DESIGN.md §3.9), so pc3d_fps_f32 does not choose it; kept for the evidence and as the base of further work. */
int pc3d_fps_pruned_f32(const float* xyz, int64_t x_bs, int64_t x_ps, int64_t x_cs, int B, int N, int S,
                        const int32_t* start, int32_t* out, void* stream);


/* K6  ball query (model/pointnet2_utils.py:84-104): out[b,s,:] = the first `nsample` point indices in ascending
 * order with |xyz_i - centers_s|^2 <= radius^2, padded with the first hit (N if there is none). out [B,S,nsample] i32.
 * Direct-difference distances (the reference's -2ab+a^2+b^2 differs by fp32 rounding at the ball's surface). */
int pc3d_ball_query_f32(const float* xyz, int64_t x_bs, int64_t x_ps, int64_t x_cs,
                        const float* centers, int64_t c_bs, int64_t c_ps, int64_t c_cs,
                        int B, int N, int S, float radius, int nsample, int32_t* out, void* stream);
/* The same query on a named kernel (pc3d_ball_query_f32 chooses by problem size; results are identical): kernel 1 = a
 * wavefront per centre, 2 = a centre per lane with the cloud split over the workgroup's four wavefronts (N <= 65535,
 * nsample <= 96). For tests and measurements. */
int pc3d_ball_query_kernel_f32(int kernel, const float* xyz, int64_t x_bs, int64_t x_ps, int64_t x_cs,
                               const float* centers, int64_t c_bs, int64_t c_ps, int64_t c_cs,
                               int B, int N, int S, float radius, int nsample, int32_t* out, void* stream);

/* K7  group gather, channels-last (model/pointnet2_utils.py:41-57,121-131): out[b,s,j,:] =
 * [xyz[b,idx[b,s,j]] - centers[b,s] (3, xyz may be NULL), feat[b,idx[b,s,j],:] (D, feat [B,N,D] contiguous or NULL)].
 * centers NULL = no subtraction (plain index_points). out [B,S,ns,3+D] contiguous. */
int pc3d_group_gather_f32(const float* xyz, int64_t x_bs, int64_t x_ps, int64_t x_cs, const float* feat, int D,
                          const int32_t* idx, const float* centers, int64_t c_bs, int64_t c_ps, int64_t c_cs,
                          int B, int N, int S, int ns, float* out, void* stream);

/* Backward of K7: grad_xyz [B,N,3] / grad_feat [B,N,D] (contiguous, OVERWRITTEN, either may be NULL) receive the
 * scatter-add of g_out [B,S,ns,(3)+D]; center_idx [B,S] (index of each centre in xyz, or NULL) receives minus the
 * sum over its group. det_ws NULL: float atomics (order-dependent in the last bits); det_ws = B*S*3 floats of scratch:
 * the ordered LDS scatter of pc3d_scatter_rows_det_f32 (deterministic). */
int pc3d_group_gather_bwd_f32(const float* g_out, const int32_t* idx, const int32_t* center_idx, int B, int N, int S,
                              int ns, int D, int has_xyz, float* grad_xyz, float* grad_feat, float* det_ws, void* stream);

/* Backward to x of out[g,c] = max_r relu(x[g,r,:] . W[c,:] + b[c]) — the last 1x1 conv + ReLU + max over the group of
 * a set-abstraction layer (model/pointnet2_utils.py:190-197, :243-257). gout/out [G,C3], arg [G,C3] int64 = winning row
 * inside the group (torch.max indices), W [C3,C2]; gx [G,ns,C2] overwritten. Sparse row accumulation in ascending
 * channel order (deterministic) instead of autograd's dense product on a one-nonzero-per-channel tensor. ns <= 128.
 * xin (may be NULL): x itself when x is the ReLU output of the previous layer — gx is zeroed where xin <= 0, which folds
 * that layer's ReLU backward into this launch. */
/* Forward of that operator without the [G*ns, C3] activation: out[g,c] = max_r relu(x[g,r,:] . W[c,:] + b[c]),
 * arg[g,c] = the winning row (int64, lowest on ties; unspecified where out == 0). x [G,ns,C2], ns <= 128,
 * C2 % 8 == 0 and <= 128, C3 % 32 == 0. fp32 MFMA. */
int pc3d_group_linear_max_f32(const float* x, const float* W, const float* b, int G, int ns, int C2, int C3,
                              float* out, int64_t* arg, void* stream);
/* The same with the kernel named (identical results): 0 = the library chooses, 1 = a workgroup per group, 2 = the
 * tiled GEMM main loop with a group-max epilogue (ns in {32, 64, 128} only, else PC3D_EINVAL). */
int pc3d_group_linear_max_kernel_f32(int kernel, const float* x, const float* W, const float* b, int G, int ns, int C2,
                                     int C3, float* out, int64_t* arg, void* stream);
int pc3d_group_max_linear_bwd_f32(const float* gout, const float* out, const int64_t* arg, const float* W,
                                  int G, int ns, int C2, int C3, const float* xin, float* gx, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * K3  DGCNN dynamic graph (model/dgcnn.py:194-227,299-313).
 * pc3d_knn_feat_f32: idx[b,i,:] = the K nearest points of i in C-dimensional feature space (self included, nearest
 * first, lowest index on ties) for channels-last features x [B,N,C]; C % 8 == 0, C <= 128, K <= 64, any N. The
 * distances |xi-xj|^2 = |xi|^2 + |xj|^2 - 2 xi.xj of 32 queries x 128 references at a time are formed by fp32 MFMA in
 * LDS and scanned against K-lists held across the lanes; nothing of size N*N reaches HBM (the reference writes
 * [B,N,N] and calls topk).
 * pc3d_gather_max_f32: out[b,i,c] = max (sign[c] < 0: min; sign NULL: max) over j in idx[b,i,:] of P[b,j,c], with
 * the winning j in arg (may be NULL) — the neighbour reduction of an EdgeConv expressed as W[xj-xi; xi] = P_j + Q_i.
 * pc3d_gather_max_bwd_f32: gP[b,arg,c] += g (gP overwritten; deterministic = 0: global float atomics, 1: one wavefront
 * per (cloud, channel slice) accumulates in LDS in source order — see pc3d_scatter_rows_det_f32).
 * ------------------------------------------------------------------------------------------------------- */
int pc3d_knn_feat_f32(const float* x, int B, int N, int C, int K, int32_t* idx, void* stream);
int pc3d_gather_max_f32(const float* P, const int32_t* idx, const float* sign, int B, int N, int C, int K,
                        float* out, int32_t* arg, void* stream);
int pc3d_gather_max_bwd_f32(const float* g, const int32_t* arg, int B, int N, int C, float* gP, int deterministic,
                            void* stream);
/* The same reduction for S query rows over N source points (idx, out, arg, g: [B,S,..]; P, gP: [B,N,C]; indices are
 * clamped to [0,N-1]): CurveNet's MaskedMaxPool, max over the ball-query neighbours of every FPS centroid
 * (model/curvenet_util.py:469-484). */
int pc3d_gather_max_rows_f32(const float* P, const int32_t* idx, int B, int N, int S, int C, int K, float* out,
                             int32_t* arg, void* stream);
int pc3d_gather_max_rows_bwd_f32(const float* g, const int32_t* arg, int B, int N, int S, int C, float* gP,
                                 int deterministic, void* stream);

/* -------------------------------------------------------------------------------------------------------
 * Deterministic scatter-add (new; the reference's autograd of index_points / get_graph_feature / knn_gather —
 * model/pointnet2_utils.py:41-57, model/dgcnn.py:203-227, attack/GeoA3/knn_utils.py:58-86 — is deterministic on its CPU
 * path, a scatter with global float atomics is not):
 *   out[b, tgt[b,r], c] = sum over the records r = 0 .. R-1, in that order, of m(val[b,r,c])
 * with m = identity, or the derivative of a LeakyReLU taken from the sign of act[b,r,c]. ONE wavefront owns the LDS tile
 * of all N destination rows x a slice of channels and walks the records in order (ds_add_f32 of one wave execute in
 * issue order), so the result is a pure function of the inputs: the same run to run, under hipGraph replay, and for a
 * cloud alone or inside a batch. tgt [B,R] int32 (outside [0,N): skipped, or clamped when clamp = 1); val [B,R,ldv],
 * act [B,R,lda] or NULL; out [B,N,ldo] overwritten (accumulate = 1: added to). N <= 40960 (a CU's LDS).
 * The deterministic modes of the backward entry points above run on this kernel.
 * ------------------------------------------------------------------------------------------------------- */
/* Sorted reverse index of a gather (CSR): for idx [B,E] int32 with values in [0,NA) (clamp = 1: clamped into it; 0:
 * others are in no list), lst[b, off[b,t] .. off[b,t+1]) = the entries e with idx[b,e] == t in ASCENDING e. cnt [B,NA]
 * int32 scratch, off [B,NA+1], lst [B,E]. pc3d_rev_gather_sum_f32: out[b,t,:] = sum over that segment, front to back, of
 * m(val[b,e,:]) (m as for pc3d_scatter_rows_det_f32) — the same sums as the scatter below, every destination row in
 * parallel; what the deterministic backward of the wide many-edge gathers (LPFA) runs on. */
int pc3d_rev_index_i32(const int32_t* idx, int B, int E, int NA, int clamp, int32_t* cnt, int32_t* off, int32_t* lst,
                       void* stream);
int pc3d_rev_gather_sum_f32(const float* val, int64_t ldv, const float* act, int64_t lda, float slope, const int32_t* off,
                            const int32_t* lst, int B, int E, int NA, int C, float* out, int64_t ldo, void* stream);
int pc3d_scatter_rows_det_f32(const int32_t* tgt, const float* val, int64_t ldv, const float* act, int64_t lda, float slope,
                              int B, int R, int N, int C, float* out, int64_t ldo, int accumulate, int clamp, void* stream);
/* One EdgeConv layer's epilogue (model/dgcnn.py:299-313): PQ [B,N,2C] = [P | Q] from ONE GEMM against [U;V];
 * out[b,i,c] = leaky_slope(max_j P[b,idx[b,i,j],c] + Q[b,i,c]), arg = the winning j. C % 4 == 0.
 * Backward: gPQ [B,N,2C] overwritten: dQ = g * leaky'(out), dP scattered to arg (float atomics). */
int pc3d_edge_max_f32(const float* PQ, const int32_t* idx, int B, int N, int C, int K, float slope,
                      float* out, int32_t* arg, void* stream);
/* The same launch also writing out into a column slice of a wider buffer (out2 + (b N + i) ld2, 16-byte aligned, ld2 % 4 == 0):
 * DGCNN concatenates its four EdgeConv outputs for conv5 (model/dgcnn.py:315 `torch.cat`) — written here, the copy goes. */
int pc3d_edge_max_cat_f32(const float* PQ, const int32_t* idx, int B, int N, int C, int K, float slope,
                          float* out, int32_t* arg, float* out2, int64_t ld2, void* stream);
/* Global pooling head (model/dgcnn.py:317-320, model/curvenet.py:64-67): z = leaky_slope(Y) (slope 0: ReLU),
 * out[b, 0:C] = max_i z[b,i,:], out[b, C:2C] = mean_i z[b,i,:], arg [B,C] = lowest arg-max row; one pass over
 * Y [B,N,C]. Backward: gY overwritten in one pass from gout [B,2C]. C % 4 == 0. Deterministic. */
int pc3d_act_pool_f32(const float* Y, int B, int N, int C, float slope, float* out, int32_t* arg, void* stream);
int pc3d_act_pool_bwd_f32(const float* Y, const float* gout, const int32_t* arg, int B, int N, int C, float slope,
                          float* gY, void* stream);
/* g [B,N,C] with row stride ldg >= C floats: the slice of a wider gradient (the backward of DGCNN's torch.cat over the four
 * EdgeConv outputs hands over [B,N,512] column slices) is read in place. */
int pc3d_edge_max_bwd_f32(const float* g, int64_t ldg, const float* out, const int32_t* arg, int B, int N, int C,
                          float slope, float* gPQ, int deterministic, void* stream);
/* The deterministic form with the channel-slice width of its owner-wave kernel named (slice & 255: 1, 2, 4, 8, 16; 0 = the
 * library chooses; identical results for every width — a wavefront per (cloud, slice) sums the points in order). Two more bits
 * for measurements: 256 = fp32 instead of fp64 LDS tiles, 512 = workgroups in launch order instead of XCD bands. */
/* The deterministic form for an EdgeConv output with TWO consumers (DGCNN: conv5 through the concatenation, and the next
 * layer): the upstream gradient is g + g2 (one fp32 add per element, on load), g2 [B,N,ldg2 >= C] — the sum autograd would
 * form in a launch of its own (model/dgcnn.py:299-315 under autograd). */
int pc3d_edge_max_bwd_sum_f32(const float* g, int64_t ldg, const float* g2, int64_t ldg2, const float* out,
                              const int32_t* arg, int B, int N, int C, float slope, float* gPQ, void* stream);
int pc3d_edge_max_bwd_slice_f32(const float* g, int64_t ldg, const float* out, const int32_t* arg, int B, int N, int C,
                                float slope, float* gPQ, int slice, void* stream);

/* K16  guided curve walk of CurveNet (model/walk.py:74-153 `Walk.forward`), one launch per step and direction: one or
 * two curves per wavefront, lane j scores neighbour j. feats [B,N,C] (C in {8,16,32,64}), adj [B,N,k] (k <= 64, self
 * excluded), start [B,cn]. agent_w [2C] / agent_b [1]: the 1x1 agent conv + eval BatchNorm folded, neighbour part
 * first (walk.py:128-131); mom_w [2,2C] / mom_b [2]: the momentum conv folded, current-feature columns first
 * (walk.py:104-115). Per step: descriptor blend by the 2-way softmax momentum, neighbour scores damped by
 * clamp(1 + cos(last move, candidate move), 0, 1) (walk.py:55-72), hard arg-max pick (lowest slot on ties) with the
 * softmax as its straight-through gradient. Outputs: curves [B,cn,L,C] (the reference's [B,C,cn,L] transposed) and,
 * for the backward, nodes / pick [B,cn,L], pre [B,cn,L,C], mom [B,cn,L,2].
 * Backward: gfeats [B,N,C] and coef [B,N] are ACCUMULATED into (zero them first); the full gradient is
 * gfeats + coef (x) agent_w[0:C] (the rank-1 score term is left to the caller as one dense pass). Float atomics.
 * deterministic = 1: the steps record their contributions in ws and the entry point sums them in record order with
 * the ordered LDS scatter (pc3d_scatter_rows_det_f32); gfeats and coef are then OVERWRITTEN, and gfeats already
 * INCLUDES the rank-1 term coef (x) agent_w[0:C] (added by the scatter on its way out: no dense pass for the caller).
 * ws: pc3d_curve_walk_bwd_ws_floats(B, cn, C, L, k, deterministic) floats of scratch (the gradients that travel from
 * step to step; in deterministic mode also the records). */
int pc3d_curve_walk_fwd_f32(const float* feats, const int32_t* adj, const int32_t* start, const float* agent_w,
                            const float* agent_b, const float* mom_w, const float* mom_b, int B, int N, int C, int k,
                            int cn, int L, float* curves, int32_t* nodes, int32_t* pick, float* pre, float* mom,
                            void* stream);
int pc3d_curve_walk_bwd_f32(const float* gcurves, const float* feats, const int32_t* adj, const float* agent_w,
                            const float* agent_b, const float* mom_w, const float* mom_b, int B, int N, int C, int k,
                            int cn, int L, const float* curves, const int32_t* nodes, const int32_t* pick,
                            const float* pre, const float* mom, float* gfeats, float* coef, float* ws, int deterministic,
                            void* stream);
int64_t pc3d_curve_walk_bwd_ws_floats(int B, int cn, int C, int L, int k, int deterministic);

/* K17  the two bandwidth-bound halves of CurveNet's local point-feature aggregation (model/curvenet_util.py:199-236,
 * `LPFA.group_feature` / `LPFA.forward`) around its 1x1-conv GEMM; channels-last, C % 4 == 0:
 *   edge_act: E[b,i,j,:] = leaky_slope(A[b,idx[b,i,j],:] + Bc[b,i,:])   A, Bc [B,N,C], idx [B,N,K] -> E [B,N,K,C]
 *             backward: gBc [B,N,C] overwritten, gA [B,N,C] ACCUMULATED (zero it first; float atomics)
 *   act_mean: out[b,i,:] = mean_j leaky_slope(Z[b,i,j,:])                Z [B,N,K,C] -> out [B,N,C]
 *             backward: gZ overwritten (deterministic). */
int pc3d_edge_act_f32(const float* A, const float* Bc, const int32_t* idx, int B, int N, int K, int C, float slope,
                      float* E, void* stream);
int pc3d_edge_act_bwd_f32(const float* gE, const float* E, const int32_t* idx, int B, int N, int K, int C, float slope,
                          float* gA, float* gBc, int deterministic, const int32_t* rev_off, const int32_t* rev_lst,
                          void* stream);
/* (deterministic = 0: gA must be zero-filled by the caller, float atomics; 1: gA overwritten — by a gather through the
 * sorted reverse index of idx when rev_off / rev_lst (pc3d_rev_index_i32 with E = N*K, NA = N, clamp = 1) are given,
 * else by the ordered LDS scatter) */
int pc3d_act_mean_f32(const float* Z, int B, int N, int K, int C, float slope, float* out, void* stream);
int pc3d_act_mean_bwd_f32(const float* Z, const float* gout, int B, int N, int K, int C, float slope, float* gZ,
                          void* stream);
/* The whole LPFA block for ONE 1x1-conv layer of equal width (model/curvenet_util.py:204-236 with mlp_num = 1, what every
 * CIC block of the classifier uses) in one launch each way, without any [B,N,K,C] tensor:
 *   out[b,i,:] = mean_j LeakyReLU_s2( W . LeakyReLU_s1(A[b,idx[b,i,j],:] + Bc[b,i,:]) + bias )
 * A, Bc, out, gout, gA, gBc [B,N,C]; idx [B,N,K] int32 (clamped); W [C,C] (out, in), Wt its transpose; C in
 * {16,32,64,128}, K <= 30. Backward: gA, gBc overwritten — gA through float atomics when edge_scratch is NULL, else
 * deterministically: the per-edge gradients go to edge_scratch [B,N,K,C] and every gA row sums its edges in ascending edge
 * order — a gather through the sorted reverse index of idx (rev_off / rev_lst from pc3d_rev_index_i32, E = N*K, NA = N,
 * clamp = 1) when given, else the ordered LDS scatter. */
int pc3d_lpfa_fused_f32(const float* A, const float* Bc, const int32_t* idx, const float* W, const float* bias, int B,
                        int N, int K, int C, float slope1, float slope2, float* out, void* stream);
int pc3d_lpfa_fused_bwd_f32(const float* gout, const float* A, const float* Bc, const int32_t* idx, const float* W,
                            const float* Wt, const float* bias, int B, int N, int K, int C, float slope1, float slope2,
                            float* gA, float* gBc, float* edge_scratch, const int32_t* rev_off, const int32_t* rev_lst,
                            void* stream);

/* K18  per-cloud half of CurveNet's curve aggregation (model/curvenet_util.py:379-437 `CurveAggregation.forward`):
 * curves [B,cn,cl,C] (channels-last) -> attention keys Kp [B,C,R] and values Vp [B,R,C], R = cn + cl, such that the
 * block's output is leaky(x + softmax_rows(x^T Kp[:, :cn]) Vp[:cn] + softmax_rows(x^T Kp[:, cn:]) Vp[cn:]).
 * w_att [C] (line_conv_att), Wa / Wb / Wc [mid,C] (conva / convb / convc), Wn / Wl [mid,mid] (convn / convl),
 * Wd [C,2 mid] + bd [C] (convd with its eval BatchNorm folded; bd is added to the first cn value rows).
 * LDS-resident, weights included: pc3d_curve_agg_lds_bytes(cn, cl, C, mid, backward) bytes, which must not exceed the
 * CU's 160 KB (error otherwise; the classifier's blocks need 58 / 97 KB). Backward: gcurves [B,cn,cl,C] overwritten
 * from gKp / gVp; deterministic. */
int64_t pc3d_curve_agg_lds_bytes(int cn, int cl, int C, int mid, int backward);
int pc3d_curve_agg_kv_f32(const float* curves, const float* w_att, const float* Wa, const float* Wb, const float* Wn,
                          const float* Wl, const float* Wc, const float* Wd, const float* bd, int B, int cn, int cl,
                          int C, int mid, float* Kp, float* Vp, void* stream);
int pc3d_curve_agg_kv_bwd_f32(const float* gKp, const float* gVp, const float* curves, const float* w_att,
                              const float* Wa, const float* Wb, const float* Wn, const float* Wl, const float* Wc,
                              const float* Wd, const float* bd, int B, int cn, int cl, int C, int mid, float* gcurves,
                              void* stream);

/* K19  channels-last glue of a CurveNet CIC block (csrc/curvenet_cl.hip); all tensors fp32, C % 4 == 0.
 *   att_scale (model/curvenet_util.py:452-455, CurveGrouping): att[p] = sigmoid(x[p,:] . w), xs[p,:] = x[p,:] att[p]
 *     for M rows; backward to x of a gradient on xs (att itself only feeds the top-k selection).
 *   topk_desc (:457): indices of the K largest scores of every cloud, descending, ties to the lower index (the
 *     reference asks torch.topk for sorted=False, whose order is device dependent — DESIGN.md A-15). N <= 8192.
 *   curve_attn (:425-437, CurveAggregation's per-point half): with the keys Kp [B,C,R] / values Vp [B,R,C] of
 *     pc3d_curve_agg_kv_f32 (R = cn + cl),  out = LeakyReLU_slope(x + softmax(x Kp[:, :cn]) Vp[:cn] + softmax(x Kp[:, cn:])
 *     Vp[cn:]).  C in {8,16,32,64}, R <= 128. Backward: gx [B,N,C], gKp, gVp (overwritten); ws = device scratch of
 *     pc3d_curve_attn_bwd_ws_floats(..) floats.
 *   lpfa_prep (:204-236, LPFA): A = x + p G1^T, Bc = p G2^T + t - x for M rows (x [M,C], p [M,3], G1, G2 [C,3], t [C]),
 *     the two per-point terms of leaky((x_j - x_i) + xyz2feature([p_i; p_j; p_j - p_i])) = leaky(A_j + Bc_i); backward
 *     gx = gA - gBc, gp = gA G1 + gBc G2. */
int pc3d_att_scale_f32(const float* x, const float* w, int64_t M, int C, float* xs, float* att, void* stream);
int pc3d_att_scale_bwd_f32(const float* g, const float* x, const float* att, const float* w, int64_t M, int C, float* gx,
                           void* stream);
int pc3d_topk_desc_f32(const float* score, int B, int N, int K, int32_t* idx, void* stream);
int pc3d_curve_attn_f32(const float* x, const float* Kp, const float* Vp, int B, int N, int C, int cn, int cl,
                        float slope, float* out, void* stream);
int64_t pc3d_curve_attn_bwd_ws_floats(int B, int N, int C, int cn, int cl);
int pc3d_curve_attn_bwd_f32(const float* gout, const float* out, const float* x, const float* Kp, const float* Vp, int B,
                            int N, int C, int cn, int cl, float slope, float* gx, float* gKp, float* gVp, float* ws,
                            void* stream);
int pc3d_lpfa_prep_f32(const float* x, const float* pts, const float* G1, const float* G2, const float* t, int64_t M,
                       int C, float* A, float* Bc, void* stream);
int pc3d_lpfa_prep_bwd_f32(const float* gA, const float* gBc, const float* G1, const float* G2, int64_t M, int C,
                           float* gx, float* gpts, void* stream);

/* K12  dense graph Laplacian L = D - A of the symmetrised kNN graph with Gaussian weights A_ij = exp(-|pi-pj|^2)
 * (attack/AOF/TAOF_attack.py:31-52, attack/AOF/Eval_AOF.py:72-93). idx [B,N,K] from pc3d_knn_f32 (self included, as
 * in the reference's topk); L [B,N,N] f32 is overwritten. Only the N*K graph edges are evaluated. */
int pc3d_graph_laplacian_f32(const float* xyz, int64_t x_bs, int64_t x_ps, int64_t x_cs,
                             const int32_t* idx, int B, int N, int K, float* L, void* stream);

/* K12b  AOF's per-iteration spectral re-projection (attack/AOF/TAOF_attack.py:114-126 at the start of a binary step,
 * :164-170 after every optimiser step): coeff = adv . V, lfc = coeff[..., :lp] . V[..., :lp]^T, hfc = coeff[..., lp:] .
 * V[..., lp:]^T — three torch.bmm with a 3-row left operand in the reference. Here two HBM-bound launches that read V^T
 * and V once each: adv, coeff (scratch), lfc, hfc [B,3,N] contiguous; V [B,N,N] (column j = eigenvector j, as
 * torch.symeig / torch.linalg.eigh return it) and Vt = V^T [B,N,N], both contiguous. Sums in a fixed order. */
int pc3d_spectral_reproject_f32(const float* adv, const float* V, const float* Vt, int B, int N, int lp, float* coeff,
                                float* lfc, float* hfc, void* stream);
/* The building block: out[b,c,r] = sum_k vec[b,c,k] * mat[b,r,k] for the 3 rows of vec [B,3,K] against every row of
 * mat [B,R,K]; with out_hi != NULL the columns k < split go to out_lo and the others to out_hi (both [B,3,R]). */
int pc3d_rowdot3_f32(const float* mat, const float* vec, int B, int R, int K, int split, float* out_lo, float* out_hi,
                     void* stream);

/* Classifier tail in one launch: logits = c2 W3^T + b3 (fc3), log_softmax / pred / adversarial loss as
 * pc3d_cls_loss_f32, and g_c2 = (g_logits W3) * (c2 > 0) (fc3 backward + ReLU mask of fc2's activation). K2 <= 256,
 * ncls <= 64. *step (may be NULL) is incremented by one. */
int pc3d_cls_tail_f32(const float* c2, int B, int K2, const float* W3, const float* b3, int ncls,
                      const int64_t* target, int kind, float kappa, float scale, float* logp,
                      int64_t* pred, float* loss, float* g_c2, int32_t* step, void* stream);

/* pc3d_cw_bookkeep_f32 + pc3d_cw_step_f32 as ONE launch (one workgroup per sample): bookkeeping on the current
 * iterate, then total gradient + Adam + clip on that sample's points. o_bestattack / input_val / m / v share adv's
 * strides; input_val / dist_val may be NULL. The step number is READ from *step_dev (or step_host). */
int pc3d_cw_update_f32(float* adv, int64_t a_bs, int64_t a_ps, int64_t a_cs,
                       const float* ori, int64_t o_bs, int64_t o_ps, int64_t o_cs, int B, int K,
                       const int64_t* pred, const int64_t* label, int untarget,
                       float* bestdist, int64_t* bestscore, float* o_bestdist, int64_t* o_bestscore,
                       float* o_bestattack, float* input_val, float* dist_val,
                       const float* g, int64_t g_bs, int64_t g_ps, int64_t g_cs, float* m, float* v,
                       double lr, double beta1, double beta2, double eps, float budget,
                       const int32_t* step_dev, int step_host, int dist_kind, const float* w,
                       const int32_t* nn_idx, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PC3D_H */
