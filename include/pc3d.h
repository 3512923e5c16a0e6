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

/* Both directions in ONE launch (grid.z = 2): a->b into (dA,iA) [B,N]; b->a into (dB,iB) [B,M]. */
int pc3d_nn_bidir_f32(const float* a, int64_t a_bs, int64_t a_ps, int64_t a_cs,
                      const float* b, int64_t b_bs, int64_t b_ps, int64_t b_cs,
                      int B, int N, int M,
                      float* dA, int32_t* iA, float* dB, int32_t* iB, void* stream);

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

#ifdef __cplusplus
}
#endif
#endif /* PC3D_H */
