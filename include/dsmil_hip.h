/*
 * dsmil_hip.h — C-ABI of libdsmil_hip.so: the MI355X (gfx950) implementation of the two DSMIL
 * hot paths.  Plain pointers and sizes only; every pointer named "device" is HBM memory owned
 * by the caller; nothing is allocated, freed or synchronised inside the library; all work is
 * enqueued on the HIP stream handed in (hipStream_t passed as void*), in order, and is
 * hipGraph-capturable.  Every function returns 0 on success or a negative DSMIL_E_* code;
 * dsmil_strerror() names it.  No C++ exceptions cross this boundary.
 *
 * The reference (binli123/dsmil-wsi) has no FFI of its own: its boundary is the nn.Module API of
 * dsmil.py.  Each entry point below therefore cites the reference forward it replaces, and
 * INTEGRATION.md shows the ctypes stub a maintainer of the reference would add.
 */
#ifndef DSMIL_HIP_H
#define DSMIL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSMIL_ABI_VERSION 5
#define DSMIL_Q_DIM 128 /* query width hard-coded at dsmil.py:31,33 */

enum {
    DSMIL_OK = 0,
    DSMIL_E_INVALID = -1,     /* null pointer / non-positive size / bad flag           */
    DSMIL_E_UNSUPPORTED = -2, /* shape or dtype outside what the kernels implement      */
    DSMIL_E_WORKSPACE = -3,   /* workspace smaller than dsmil_*_workspace_bytes() says  */
    DSMIL_E_LAUNCH = -4,      /* hipGetLastError() != hipSuccess after a launch         */
    DSMIL_E_ALIGN = -5        /* a pointer is not aligned as the kernels require        */
};

enum { DSMIL_F32 = 0, DSMIL_BF16 = 1 };

/* Parameters of FCLayer + BClassifier, row-major fp32 device pointers, named after the
 * reference state_dict keys (dsmil.py:9,31,33,44):
 *   fc_w  [C,K]  fc_b [C]      i_classifier.fc.0.{weight,bias}      (may be NULL when the
 *                               caller supplies instance logits, see classes_in)
 *   q0_w  [128,K] q0_b [128]   b_classifier.q.0.*  (b_classifier.q.* when nonlinear == 0)
 *   q2_w  [128,128] q2_b [128] b_classifier.q.2.*  (ignored when nonlinear == 0)
 *   fcc_w [C,C,Kv] fcc_b [C]   b_classifier.fcc.*  (Conv1d(C,C,kernel_size=Kv))
 */
typedef struct dsmil_agg_params {
    const float* fc_w;
    const float* fc_b;
    const float* q0_w;
    const float* q0_b;
    const float* q2_w;
    const float* q2_b;
    const float* fcc_w;
    const float* fcc_b;
    int32_t K;         /* feature width of feats (512; 166 MUSK1; 1024 tree)           */
    int32_t Kv;        /* width of the value rows (== K unless passing_v)               */
    int32_t C;         /* number of classes                                             */
    int32_t nonlinear; /* dsmil.py:30-33: 1 = Linear-ReLU-Linear-Tanh query, 0 = Linear  */
} dsmil_agg_params;

/* Replaces MILNet.forward / FCLayer.forward + BClassifier.forward (dsmil.py:10-12,46-62,70-74)
 * for a BATCH of n_bags independent bags stored back to back ("varlen"):
 *   feats    device [total_rows, K] fp32 row-major; bag b owns rows offsets[b]..offsets[b+1]-1
 *   vals     device [total_rows, Kv] fp32 — V of dsmil.py:48; pass feats (or NULL) when
 *            v = Identity (passing_v=False, the only form any reference script uses)
 *   offsets  device int64 [n_bags+1], offsets[0] == 0, non-decreasing, every bag >= 1 row
 *   max_rows the largest bag length (host value; sizes the launch grid)
 *   classes_in  device [total_rows, C] or NULL.  NULL: instance logits are computed here
 *            (FCLayer) and written to classes_out.  Non-NULL: BClassifier.forward(feats, c)
 *            with caller-supplied c (attention_map.py:85); classes_out may then be NULL.
 * Outputs (device, fp32 unless noted):
 *   classes_out [total_rows, C]   instance logits           (dsmil.py:11)
 *   A           [total_rows, C]   attention, softmax over each bag's instances (dsmil.py:56)
 *   B           [n_bags, C, Kv]   bag embeddings            (dsmil.py:57-59)
 *   pred        [n_bags, C]       bag logits                (dsmil.py:60-61)
 *   idx         int64 [n_bags, C] critical-instance index, bag-local (dsmil.py:52, row 0 of
 *                                 the descending sort; lowest index wins on exact ties)
 *   ws / ws_bytes  scratch of at least dsmil_agg_workspace_bytes(...) bytes, 256-B aligned
 */
int dsmil_agg_forward(const float* feats, const float* vals, const int64_t* offsets,
                      int32_t n_bags, int64_t total_rows, int64_t max_rows,
                      const dsmil_agg_params* p, const float* classes_in, float* classes_out,
                      float* A, float* B, float* pred, int64_t* idx, void* ws, size_t ws_bytes,
                      void* stream);

size_t dsmil_agg_workspace_bytes(int32_t n_bags, int64_t total_rows, int32_t K, int32_t Kv,
                                 int32_t C);

/* bf16-storage variant (BASELINE.json configs[2]: "2-class DSMIL aggregator bf16"; the reference
 * itself is fp32 only).  feats/vals are bfloat16 [total_rows,K] / [total_rows,Kv]; the query MLP
 * runs on bf16 MFMA with f32 accumulation from `packed` (dsmil_agg_pack_bf16: q0_w/q2_w rounded
 * to bf16, RNE); instance logits, scores, softmax, value sum and the bag head accumulate in f32
 * from the fp32 pointers in *p (the caller passes bf16-rounded values there if it wants the
 * "everything rounded to bf16" semantics of module.bfloat16()).  Outputs are fp32.
 * Requires K % 8 == 0 and Kv % 4 == 0 (else DSMIL_E_UNSUPPORTED). */
size_t dsmil_agg_packed_bf16_bytes(int32_t K);
int dsmil_agg_pack_bf16(const float* q0_w, const float* q2_w, int32_t K, void* packed, void* stream);
int dsmil_agg_forward_bf16(const void* feats_bf16, const void* vals_bf16, const int64_t* offsets,
                           int32_t n_bags, int64_t total_rows, int64_t max_rows,
                           const dsmil_agg_params* p, const void* packed, const float* classes_in,
                           float* classes_out, float* A, float* B, float* pred, int64_t* idx, void* ws,
                           size_t ws_bytes, void* stream);

/* ---- one bag sharded by INSTANCES over several GPUs ----------------------
 * Each rank holds a contiguous row range of ONE bag and exchanges C*(2+K) floats twice instead of
 * all-gathering the feature rows.  Per rank:
 *   1. dsmil_agg_shard_argmax: classes_out[rows,C] = FCLayer(feats) (dsmil.py:11) and, per class, the
 *      shard's best (value, shard-local index) of dsmil.py:52 (lowest index on ties).
 *      -> exchange (value, global index, the feature row feats[best_idx]) and keep, per class, the
 *         bag-wide winner's row: crit_rows[C,K].
 *   2. dsmil_agg_shard_attend: q_max = q(crit_rows) (dsmil.py:53-54), then the fused query/score/
 *      value-sum kernel over this shard: A_unnorm[rows,C] = exp(s - m_shard), ml[C,2] = (m_shard,
 *      sum_n exp(s - m_shard)), B_unnorm[C,Kv] = sum_n exp(s - m_shard) V[n].
 *      -> merge over ranks: m = max m_r, w_r = exp(m_r - m), l = sum l_r w_r, B = sum B_r w_r / l,
 *         A = A_unnorm w_r / l on each rank, pred = Conv1d head on B (dsmil.py:57-61).
 * Workspace: dsmil_agg_workspace_bytes(1, rows, K, Kv, C).  fp32 only. */
int dsmil_agg_shard_argmax(const float* feats, int64_t rows, const dsmil_agg_params* p,
                           float* classes_out, float* best_val, int64_t* best_idx, void* ws,
                           size_t ws_bytes, void* stream);
int dsmil_agg_shard_attend(const float* feats, const float* vals, int64_t rows,
                           const dsmil_agg_params* p, const float* crit_rows, float* A_unnorm,
                           float* ml, float* B_unnorm, void* ws, size_t ws_bytes, void* stream);

/* Which MFMA form dsmil_agg_forward uses for the fp32 query MLP (dsmil.py:31-33,49):
 *   6 (default) — bf16 MFMA over exact three-plane cuts of both fp32 operands (csrc/agg_split.h), the six
 *                 largest of the nine plane products; the three left out are together < 2^-20 of |x*w|,
 *                 below the rounding an fp32 dot product of this length carries anyway
 *   9           — all nine plane products: every fp32 product formed exactly (env DSMIL_MLP=s9)
 *   0           — v_mfma_f32_32x32x2_f32 (env DSMIL_MLP=f32)
 * The product library always uses 6; experiment builds (libdsmil_hip_expt.so) read the environment variable DSMIL_MLP
 * once per process. */
int dsmil_agg_mlp_form(void);

/* Where the critical instance's query q_max (dsmil.py:53-54) is computed on the few-rows fp32 path (a lone bag, a training
 * step: kernel k_attend_hs): mode 1 (default) = by the first C workgroups of the attend launch itself, handed to the
 * tiles through an agent-scope release/acquire flag (only when the whole launch is resident at once; larger batches take
 * the separate launch anyway); mode 0 = always the separate k_qmax launch between the logits pass and the attend kernel.
 * Both produce the same bits; the switch exists so that a test can compare them (tests/test_agg_gpu.py, hand-off stress).
 * Process-wide.  Returns the previous mode; any other `mode` only queries. */
int dsmil_agg_inline_query(int mode);

/* Which kernel a BATCH of fp32 bags (>= 512 tiles of 128 rows; v = Identity; K a multiple of 128 up to 512) takes for
 * dsmil.py:49-57.  All modes are the same fp32-class arithmetic (tests/test_agg_gpu.py compares them):
 *   2 (default)  k_attend_f3 (csrc/agg_f3.h) for the two-layer query with C <= 2: the query weights stay in registers for
 *                the whole launch, 32-row tiles resident in LDS from the query MLP to the value sum (every feature byte read
 *                once), fp16 MFMA over two-plane cuts of the row-scaled operands, three plane products, one partial per
 *                (workgroup, bag); every other case as mode 1;
 *   1            k_attend_f2 (csrc/agg_f2.h): the same arithmetic on 64-row tiles, weights streamed from L2 per tile;
 *   0            k_query_attend_split of rounds 2-4 (bf16 MFMA, exact three-plane cuts, six products, the tile read twice).
 * Process-wide; returns the previous mode; any other `mode` only queries. */
int dsmil_agg_batch_form(int mode);

/* (ABI 5) Workgroups of the persistent batch kernels (k_attend_f3 / k_attend_f2 / k_attend_bf16_res).  The default is the
 * CONSTANT 256 (the CUs of an unpartitioned MI355X), not the visible CU count: the run of tiles a workgroup owns fixes which
 * tiles share a partial and with it the fp32 summation order, so the outputs are bit-identical on every box (partition
 * mode, masked CUs) — tests/test_agg_gpu.py checks two other grids against the fp64 oracle.  n in 1..1024 sets it
 * (process-wide) and returns the previous value; any other n only queries. */
int dsmil_agg_persistent_grid(int n);

/* (ABI 5, round 6) The kernels around the persistent attend kernel of a bf16 BATCH (K = 512, C <= 2; dsmil.py:50-54 — the
 * instance logits, the critical instance and its query — and the combine of dsmil.py:57-61).  The co-resident forms
 * (k_logits_pipe, a 4-wave k_qmax, the lean k_finish) hold <= 80 registers per lane and a few KiB of LDS, so one wave of them
 * fits on every SIMD beside the resident k_attend_bf16_res workgroup of ANOTHER stream's batch (432 of a SIMD's 512
 * registers): with two or more streams in flight the second read of the features runs under the MFMA kernel of the batch in
 * front instead of behind it (+12-17 % bags/s); alone on the chip they are ~6 % slower than the plain forms.
 *   1 (default)  co-resident forms when the last few batch calls arrived on more than one stream, else the plain forms;
 *   2            always the co-resident forms;      0   never (k_logits_stream, 16-wave k_qmax, k_finish).
 * Outputs are bit-identical in every mode (tests/test_agg_bf16_gpu.py).  Process-wide; returns the previous mode; any other
 * `mode` only queries. */
int dsmil_agg_logits_form(int mode);

/* Compute units of the current device as the library sees them (256 on MI355X); <= 0 without a device.  Diagnostic: recorded
 * by bench.py and the soak so that a result can be tied to the box it ran on. */
int dsmil_device_cus(void);

/* Options of dsmil_agg_forward_ex (all optional; a NULL opts or an all-zero struct = dsmil_agg_forward):
 *   packed_split  the plane-cut query weights of forms 6 / 9 prepared ONCE per weight set instead of on every
 *                 forward (BClassifier.q changes only at optimizer.step(), train_tcga.py:73): dsmil_agg_pack_split
 *                 cuts q0_w [128,K] and q2_w [128,128] (NULL when nonlinear == 0) into a buffer of
 *                 dsmil_agg_packed_split_bytes(K, nonlinear) bytes, 16-B aligned.  Ignored by form 0.
 *   row_map       int64 [total_rows]: logical row i of the batch (bag b, instance i - offsets[b]) lives at physical
 *                 row row_map[i] of feats / vals.  This is train_tcga.py:78-83 `dropout_patches` (a random subset /
 *                 permutation of a bag's rows, `feats[random_indices]`) as an index list folded into the kernels' row
 *                 loads instead of a gathered 20 MB copy.  classes_out / A / idx stay in LOGICAL order (= the order of
 *                 the reference's gathered tensor).
 *   packed_f2     (ABI 4) the same query weights in the form BATCHES of fp32 bags use since round 5 (kernel k_attend_f2:
 *                 two fp16 planes of the power-of-two scaled weights, csrc/agg_f2.h), prepared once per weight set by
 *                 dsmil_agg_pack_f2 into dsmil_agg_packed_f2_bytes(K) bytes, 16-B aligned; NULL = cut inside the forward. */
typedef struct dsmil_agg_opts {
    const void* packed_split;
    const int64_t* row_map;
    const void* packed_f2;
} dsmil_agg_opts;
size_t dsmil_agg_packed_split_bytes(int32_t K, int32_t nonlinear);
int dsmil_agg_pack_split(const float* q0_w, const float* q2_w, int32_t K, void* packed, void* stream);
size_t dsmil_agg_packed_f2_bytes(int32_t K);
int dsmil_agg_pack_f2(const float* q0_w, const float* q2_w, int32_t K, void* packed, void* stream);
int dsmil_agg_forward_ex(const float* feats, const float* vals, const int64_t* offsets,
                         int32_t n_bags, int64_t total_rows, int64_t max_rows,
                         const dsmil_agg_params* p, const dsmil_agg_opts* opts, const float* classes_in,
                         float* classes_out, float* A, float* B, float* pred, int64_t* idx, void* ws,
                         size_t ws_bytes, void* stream);

/* The training objective of ONE bag, train_tcga.py:67-71 (also train_mil.py):
 *   max_prediction = max_n ins_prediction[n,:]  (= classes[idx_c, c], idx = the forward's critical instances)
 *   loss = 0.5 BCEWithLogitsLoss(bag_prediction, y) + 0.5 BCEWithLogitsLoss(max_prediction, y)   (mean over C)
 * and its gradients with respect to the two logit vectors, in one launch (the reference spends ~10 small kernels
 * and their autograd nodes on it):  loss[1], max_pred[C], g_pred[C] = dloss/dpred, g_max[C] = dloss/dmax_pred
 * (each may be NULL except loss).  `classes` / `pred` / `idx` are the forward's outputs for this bag, `label` [C]
 * fp32 0/1.  C <= 64. */
int dsmil_agg_loss_head(const float* classes, const float* pred, const int64_t* idx, const float* label,
                        int32_t C, float* loss, float* max_pred, float* g_pred, float* g_max, void* stream);

/* FCLayer.forward alone (dsmil.py:10-12): classes[total_rows, C] = feats @ fc_w^T + fc_b. */
int dsmil_fc_forward(const float* feats, int64_t total_rows, int32_t K, int32_t C,
                     const float* fc_w, const float* fc_b, float* classes, void* stream);

/* ---- aggregator backward (training step) -----------------------------------------------------
 * Replaces what autograd derives for `loss.backward()` in train_tcga.py:67-72 / train_mil.py for
 * ONE bag through MILNet.forward (dsmil.py:70-74): gradients of every parameter of FCLayer
 * (dsmil.py:6-12) and BClassifier (dsmil.py:27-62) given the upstream gradients of the forward's
 * outputs.  Device pointers, fp32, each gradient has its parameter's shape and is OVERWRITTEN
 * (the caller accumulates into .grad if it needs to).  g_classes / g_A / g_B may be NULL (treated
 * as zero; with g_classes NULL the fc_* gradients are not written).  The arg-max indices are
 * constants of the graph, as in torch.sort + index_select (dsmil.py:51-53).  `A`, `B`, `idx` are
 * the forward's outputs for this bag.  With passing_v (BClassifier.v = Dropout+Linear+ReLU,
 * dsmil.py:39-44) the caller hands the value rows in `vals` and asks for their gradient in
 * `g_vals` [N, Kv] (= A gB^T, NULL to skip), which it pushes through its own v layer. */
typedef struct dsmil_agg_grads {
    float* fc_w;   /* [C, K]      */
    float* fc_b;   /* [C]         */
    float* q0_w;   /* [128, K]    */
    float* q0_b;   /* [128]       */
    float* q2_w;   /* [128, 128]  (unused when !nonlinear) */
    float* q2_b;   /* [128]       */
    float* fcc_w;  /* [C, C, Kv]  */
    float* fcc_b;  /* [C]         */
} dsmil_agg_grads;

size_t dsmil_agg_backward_workspace_bytes(int64_t N, int32_t K, int32_t Kv, int32_t C);
/* dsmil_agg_backward_ex adds: g_max [C] — the SPARSE instance-stream gradient of the training objective (the max
 * over instances touches one row per class: g_fc_w[c] += g_max[c] * feats[idx_c], g_fc_b[c] += g_max[c]; may be
 * combined with a dense g_classes or replace it) and row_map (see dsmil_agg_opts; N logical rows).  A, g_A, g_vals,
 * g_classes are in logical row order. */
int dsmil_agg_backward_ex(const float* feats, const float* vals, int64_t N, const dsmil_agg_params* p,
                          const float* A, const float* B, const int64_t* idx, const float* g_classes,
                          const float* g_max, const float* g_pred, const float* g_A, const float* g_B,
                          const dsmil_agg_grads* g, float* g_vals, const int64_t* row_map, void* ws,
                          size_t ws_bytes, void* stream);
int dsmil_agg_backward(const float* feats, const float* vals, int64_t N, const dsmil_agg_params* p,
                       const float* A, const float* B, const int64_t* idx, const float* g_classes,
                       const float* g_pred, const float* g_A, const float* g_B,
                       const dsmil_agg_grads* g, float* g_vals, void* ws, size_t ws_bytes,
                       void* stream);

/* ---- one training step per C call (ABI 3) --------------------------------------------------------
 * Replaces the body of the reference's training loop for one bag, train_tcga.py:60-75 (train_mil.py:44-56 likewise):
 *     optimizer.zero_grad()
 *     ins_prediction, bag_prediction, _, _ = milnet(bag_feats)                     (:67)
 *     max_prediction, _ = torch.max(ins_prediction, 0)                             (:68)
 *     loss = 0.5 * BCEWithLogitsLoss(bag_prediction, y) + 0.5 * BCEWithLogitsLoss(max_prediction, y)   (:69-71)
 *     loss.backward(); optimizer.step()                                            (:72-73)
 * with optimizer = torch.optim.Adam(milnet.parameters(), lr, betas, weight_decay) (:241; amsgrad = maximize = False).
 * The forward, the loss head, the backward and ONE Adam kernel over the eight parameter tensors are enqueued on
 * `stream`; nothing synchronises.  The parameters in *p are UPDATED IN PLACE (they are the optimiser's tensors), as
 * are the moment tensors in *opt; `loss` (device, 1 float) receives the step's loss (the caller reads it for the
 * progress line of :74 — the step's only host sync).
 *   feats    device [rows, K] fp32, the bag;  row_map int64 [N] or NULL (dropout_patches, :78-83, as an index list)
 *   N        instances that enter the bag (= rows when row_map is NULL)
 *   label    device [C] fp32, the bag label (0/1)
 *   p        MILNet(FCLayer, BClassifier) parameters, v = Identity (Kv == K), C <= 64
 *   opt      Adam state: exp_avg / exp_avg_sq = 8 device pointers each in the order fc_w, fc_b, q0_w, q0_b, q2_w, q2_b,
 *            fcc_w, fcc_b (the q2 entries are ignored when !nonlinear); step = the 1-based index of THIS update
 *            (torch's state['step'] after its increment); hyper-parameters as Python floats (double)
 *   ws       dsmil_agg_train_step_workspace_bytes(N, K, C, nonlinear) bytes, 256-B aligned
 * dsmil_adam_step is the optimiser kernel alone (n_tensors <= DSMIL_ADAM_MAX_TENSORS; numel[i] == 0 skips entry i). */
#define DSMIL_ADAM_MAX_TENSORS 8
typedef struct dsmil_adam_state {
    float* const* exp_avg;
    float* const* exp_avg_sq;
    int64_t step;
    double lr, beta1, beta2, eps, weight_decay;
} dsmil_adam_state;
size_t dsmil_agg_train_step_workspace_bytes(int64_t N, int32_t K, int32_t C, int32_t nonlinear);
int dsmil_agg_train_step(const float* feats, int64_t N, const int64_t* row_map, const float* label,
                         const dsmil_agg_params* p, const dsmil_adam_state* opt, float* loss, void* ws,
                         size_t ws_bytes, void* stream);
int dsmil_adam_step(int32_t n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                    float* const* exp_avg_sq, const int64_t* numel, int64_t step, double lr, double beta1,
                    double beta2, double eps, double weight_decay, void* stream);

/* ---- patch embedder: ResNet-18 with InstanceNorm2d, fc = Identity --------------------------
 * Replaces the torchvision backbone that compute_feats.py:157,170 builds and dsmil.IClassifier
 * wraps (dsmil.py:21-25): feats[B,512] = flatten(avgpool(resnet18_IN(x))), and, when `classes`
 * is non-NULL, classes[B,C] = feats @ fc_w^T + fc_b (IClassifier.fc, dsmil.py:24).
 *   x_nchw   device [B,3,H,W] fp32 in [0,1] (what VF.to_tensor yields, compute_feats.py:35-39)
 *   conv1_w  device [64,3,7,7] fp32 — feature_extractor.conv1.weight, used as is
 *   packed   device buffer of dsmil_resnet18_packed_bytes() bytes filled by dsmil_resnet18_pack()
 *            from the other 19 conv weights
 *   ws       scratch of dsmil_resnet18_workspace_bytes(B,H,W) bytes, 256-B aligned
 * conv_w[20]: device pointers to the 20 bias-free conv weights [Cout,Cin,k,k] fp32 in torchvision
 * state_dict order (conv1; layerL.0.conv1, layerL.0.conv2, [layerL.0.downsample.0], layerL.1.conv1,
 * layerL.1.conv2 for L = 1..4) — the order compute_feats.py:226-231 relies on.  conv_w[0] is not
 * read by the packer. */
size_t dsmil_resnet18_packed_bytes(void);
int dsmil_resnet18_pack(const float* const* conv_w, float* packed, void* stream);
size_t dsmil_resnet18_workspace_bytes(int32_t B, int32_t H, int32_t W);
int dsmil_resnet18in_forward(const float* x_nchw, int32_t B, int32_t H, int32_t W,
                             const float* conv1_w, const float* packed, const float* fc_w,
                             const float* fc_b, int32_t C, float* feats, float* classes, void* ws,
                             size_t ws_bytes, void* stream);

/* The same forward fed with DECODED images: x_nhwc is uint8 [B,H,W,3] (row-major, RGB interleaved, as
 * PIL / numpy hold a patch).  The ToTensor step of the reference's loader (compute_feats.py:35-39:
 * VF.to_tensor = HWC uint8 -> CHW float32 / 255, IEEE division) is fused into the stem's input
 * staging, so results are bit-identical to dsmil_resnet18in_forward on the converted tensor while
 * the host->device copy and the first HBM read shrink 4x. */
int dsmil_resnet18in_forward_u8(const uint8_t* x_nhwc, int32_t B, int32_t H, int32_t W,
                                const float* conv1_w, const float* packed, const float* fc_w,
                                const float* fc_b, int32_t C, float* feats, float* classes, void* ws,
                                size_t ws_bytes, void* stream);

/* The same trunk with FROZEN-statistics norms — eval-mode nn.BatchNorm2d, i.e. the reference's
 * `--norm_layer batch` / ImageNet-pretrained extractor (compute_feats.py:149-154, run under
 * i_classifier.eval()).  Each of the 20 norms is y = (x - m[c]) * r[c] with
 *   r = weight / sqrt(running_var + eps),  m = running_mean - bias / r        (r != 0)
 * folded by the caller; bn_mean / bn_rstd are the 20 per-channel arrays concatenated in conv order
 * (dsmil_resnet18_norm_channels() = 4800 floats each).  All conv / pool / residual kernels are the
 * InstanceNorm ones; only the statistics step is replaced.  x is fp32 NCHW, or uint8 NHWC when
 * x_is_u8_nhwc != 0 (see dsmil_resnet18in_forward_u8). */
int32_t dsmil_resnet18_norm_channels(void);
int dsmil_resnet18bn_forward(const void* x, int32_t x_is_u8_nhwc, int32_t B, int32_t H, int32_t W,
                             const float* conv1_w, const float* packed, const float* bn_mean,
                             const float* bn_rstd, const float* fc_w, const float* fc_b, int32_t C,
                             float* feats, float* classes, void* ws, size_t ws_bytes, void* stream);

/* Generic trunk: BasicBlock depth 18 (blocks [2,2,2,2], 20 convs) or 34 ([3,4,6,3], 36 convs), Bottleneck depth 50
 * ([3,4,6,3], 53 convs) or 101 ([3,4,23,3], 104 convs; stride on the 3x3 conv, torchvision's v1.5) — the
 * reference's `--backbone resnet18|resnet34|resnet50|resnet101` (compute_feats.py:155-167).  conv_w is the trunk's
 * dsmil_resnet_num_convs(depth) conv tensors in state_dict order; bn_mean / bn_rstd are NULL for
 * InstanceNorm or the folded frozen-BatchNorm arrays (dsmil_resnet_norm_channels(depth) floats, see
 * dsmil_resnet18bn_forward); x is fp32 NCHW or uint8 NHWC.  Workspace as dsmil_resnet18_workspace_bytes
 * (the activation shapes do not depend on the depth).  The *18* entry points above are these with
 * depth = 18.  feats is [B, dsmil_resnet_feature_dim(depth)] (512 for BasicBlock trunks, 2048 for Bottleneck trunks); the
 * workspace of a Bottleneck trunk is larger: dsmil_resnet_workspace_bytes(depth, B, H, W). */
size_t dsmil_resnet_workspace_bytes(int32_t depth, int32_t B, int32_t H, int32_t W);
int32_t dsmil_resnet_feature_dim(int32_t depth);
int32_t dsmil_resnet_num_convs(int32_t depth);
/* Which matrix pipe the trunk's convolutions run on (for roofline accounting): plane products per fp32 MAC of the
 * Winograd convs (3x3 stride 1) and of the direct convs (3x3 stride 2, 1x1) — 3 = fp16 MFMA over two-plane cuts (round 5,
 * the product form), 9 / 6 = bf16 MFMA over exact three-plane cuts, 0 = v_mfma_f32_32x32x2_f32.  The product library has
 * one form (3 / 3); experiment builds read DSMIL_WINO / DSMIL_CONV once per process. */
int dsmil_resnet_mfma_forms(int32_t* wino_products, int32_t* direct_products);
int32_t dsmil_resnet_norm_channels(int32_t depth);
size_t dsmil_resnet_packed_bytes(int32_t depth);
int dsmil_resnet_pack(int32_t depth, const float* const* conv_w, float* packed, void* stream);
int dsmil_resnet_forward(int32_t depth, const void* x, int32_t x_is_u8_nhwc, int32_t B, int32_t H,
                         int32_t W, const float* conv1_w, const float* packed, const float* bn_mean,
                         const float* bn_rstd, const float* fc_w, const float* fc_b, int32_t C,
                         float* feats, float* classes, void* ws, size_t ws_bytes, void* stream);

/* OPT-IN reduced precision (round 5; no reference counterpart: the reference embeds in fp32, compute_feats.py:70-76).
 * precision = 0: the calls above (fp32-class: every conv as three fp16 plane products of two-plane cuts, csrc/resnet_fwd.hip
 * PlaneProducts<3>).  precision = 1: every conv operand — activations behind the norm + ReLU, and the weights — is rounded to ONE
 * fp16 plane (11 significand bits, round to nearest); products accumulate in f32 on the same MFMA; activations between the layers,
 * InstanceNorm statistics and the pooling stay fp32.  Feature error against the fp32-class path: ~2e-3 abs on features of O(1)
 * (tools/form_error_study.py `f16x1`, tests/test_resnet_gpu.py) — NOT the 1e-4 parity bar; for callers who ask for it
 * (compute_feats.py --precision half).  The packed image must come from dsmil_resnet_pack_ex with the SAME precision (same size as
 * dsmil_resnet_packed_bytes). */
int dsmil_resnet_pack_ex(int32_t depth, const float* const* conv_w, float* packed, int32_t precision, void* stream);
/* (ABI 5, round 6) precision = 2: the OPT-IN bf16-ACTIVATION trunk (csrc/resnet_b16.h; BASELINE.md's "bf16 MFMA / f32 accumulate"
 * row): behind the stem (which runs as in precision 1) every activation is stored in bf16 — NHWC with a one-pixel zero border —
 * and every conv is ONE bf16 MFMA product per MAC with f32 accumulation; InstanceNorm statistics are f32.  ResNet-18 / 34 with
 * InstanceNorm and patches between 64 x 64 and ~1000 pixels wide only (everything else: DSMIL_E_UNSUPPORTED); features agree with
 * precision 0 to bf16 rounding (max ~2e-2, mean ~3e-3 on features of magnitude ~1), NOT to the 1e-4 bar; 1.75-2x the rate of
 * precision 0.
 * precision = 3: the SAME trunk on fp16 activations (11 significant bits instead of 8: features within ~2.6e-3 of precision 0,
 * the class of precision 1, at the rate of precision 2) — activations and conv sums must stay inside fp16's +-65504: true for
 * InstanceNorm trunks with ordinary weights; a caller checks the first forward of a weight set for non-finite features (the
 * Python binding does, and falls back to precision 1).  compute_feats.py --precision half takes it where it applies.
 * The packed image of precisions 2 / 3 is LARGER: size it with dsmil_resnet_packed_bytes_ex(depth, precision) (0 = unsupported
 * depth / precision; precision 0 / 1 = dsmil_resnet_packed_bytes).  Workspace as for the other precisions.  The reference has
 * no such switch; compute_feats.py --precision bf16 exposes precision 2. */
size_t dsmil_resnet_packed_bytes_ex(int32_t depth, int32_t precision);
int dsmil_resnet_forward_ex(int32_t depth, const void* x, int32_t x_is_u8_nhwc, int32_t B, int32_t H, int32_t W,
                            const float* conv1_w, const float* packed, const float* bn_mean, const float* bn_rstd,
                            const float* fc_w, const float* fc_b, int32_t C, float* feats, float* classes, void* ws,
                            size_t ws_bytes, int32_t precision, void* stream);

/* ---- background filters of the reference's tilers on decoded tiles -------------------------
 * tiles_nhwc: device uint8 [B,H,W,3] (W <= 1024).  out: device uint64 [B,4] = per tile
 *   {sum over band 0, band 1, band 2 of PIL's ImageFilter.FIND_EDGES image, sum of img_as_ubyte(rgb2hsv(img)[...,1])}.
 * deepzoom_tiler.py:56-61 keeps a tile when mean(out[0..2]) / tile_size^2 > threshold (default 15);
 * test_crop_single.py:17-24 keeps it when out[3] / (H*W) >= t (t = 30 at its call site).  The sums are exact
 * integers: forming the ratios in float64 on the host reproduces the reference's decisions bit for bit. */
int dsmil_tile_stats(const uint8_t* tiles_nhwc, int32_t B, int32_t H, int32_t W, uint64_t* out, void* stream);

/* ---- batched baseline-JPEG decode of a slide's tiles (ABI 5) -----------------------------------------
 * Replaces the per-tile `Image.open(path)` of the reference's loaders (compute_feats.py:28,107; attention_map.py:69-79:
 * Pillow / libjpeg-turbo in DataLoader worker processes) for baseline JPEG tiles (what deepzoom_tiler.py:64 writes): the
 * COMPRESSED bytes of a batch go to the device and are decoded there into uint8 NHWC — the input of
 * dsmil_resnet_forward_ex(x_is_u8_nhwc = 1) — bit for bit what Pillow's defaults produce (islow IDCT, fancy upsampling,
 * YCbCr -> RGB; a grey image gives R = G = B).  Scope: SOF0, 8 bit, Huffman, one interleaved scan, 1 or 3 components, luma
 * sampling 1x1 / 2x1 / 2x2 with 1x1 chroma, restart intervals, arbitrary tables, at most DSMIL_JPEG_MAX_QTABLES distinct
 * quantisation and DSMIL_JPEG_MAX_HTABLES distinct Huffman tables per batch (a tiler writes the same ones into every tile).
 *
 *   dsmil_jpeg_parse   HOST function (no device work): data = the files of the batch back to back in HOST memory, offsets
 *                      [n + 1]; fills `plan` (host memory, dsmil_jpeg_plan_bytes(n) bytes, 16-B aligned): one
 *                      dsmil_jpeg_image per file at byte offset 16 of the plan (status = DSMIL_OK, or DSMIL_E_UNSUPPORTED /
 *                      DSMIL_E_INVALID for a file outside the scope — the caller decodes THOSE with Pillow) followed by the
 *                      batch's de-duplicated tables.  The caller copies data and plan to the device as they are.
 *   dsmil_jpeg_decode  data (data_bytes = offsets[n] bytes), plan: the DEVICE copies; every image with status DSMIL_OK must be width x height; out_nhwc
 *                      device uint8 [n, height, width, 3] (rows of images with another status are left untouched);
 *                      status: device int32 [n] = the record's status, or DSMIL_E_INVALID when the entropy-coded data
 *                      turned out corrupt (the image is then undefined); ws: dsmil_jpeg_workspace_bytes(n, height, width,
 *                      data_bytes) bytes, 256-B aligned.  A memset and four launches on `stream`, no host synchronisation. */
#define DSMIL_JPEG_MAX_QTABLES 256   /* distinct quantisation tables per batch (128 B each in the plan) */
#define DSMIL_JPEG_MAX_HTABLES 64    /* distinct Huffman tables per batch (8.4 KiB each in the plan) */
typedef struct dsmil_jpeg_image {
    int64_t ecs_begin, ecs_end;   /* entropy-coded segment: byte offsets into `data` */
    int32_t width, height;
    int32_t ncomp;                /* 1 (grey) or 3 (YCbCr) */
    int32_t hsamp, vsamp;         /* luma sampling factors (chroma is 1x1) */
    int32_t restart_interval;     /* MCUs between RSTn markers, 0 = none */
    int32_t qt[3];                /* per component: index of its quantisation table in the plan */
    int32_t dc[3], ac[3];         /* per component: indices of its Huffman tables in the plan */
    int32_t status;               /* DSMIL_OK, DSMIL_E_UNSUPPORTED, DSMIL_E_INVALID */
} dsmil_jpeg_image;
size_t dsmil_jpeg_plan_bytes(int32_t n);
size_t dsmil_jpeg_workspace_bytes(int32_t n, int32_t height, int32_t width, int64_t data_bytes);
int dsmil_jpeg_parse(const uint8_t* data, const int64_t* offsets, int32_t n, void* plan);
int dsmil_jpeg_decode(const uint8_t* data, int64_t data_bytes, const void* plan, int32_t n, int32_t height, int32_t width,
                      uint8_t* out_nhwc, int32_t* status, void* ws, size_t ws_bytes, void* stream);

/* ---- the reference's feature files (round 6; the data format on both sides of the embedder -> aggregator path) -----------
 * compute_feats.py:80-82 writes a bag's features as `pd.DataFrame(feats).to_csv(path, index=False, float_format='%.4f')` and
 * train_tcga.py:27-32 reads them back; pandas formats the 5.1 M numbers of a 10 000 x 512 bag in 5.4 s — 30-60x the time the
 * device needs to compute them.  dsmil_csv_format_f32 is a HOST function (no device work) that writes the SAME BYTES for the
 * data rows: x float32 [rows, cols] in host memory (rows `row_stride` elements apart), `decimals` in 0..9 (the reference: 4);
 * fields separated by ',', rows ended by '\n', NaN an empty field, infinities 'inf' / '-inf', negative zero '-0.0000' (what
 * pandas writes).  out: at least 64 bytes per value.  Returns the number of bytes written, DSMIL_E_WORKSPACE when `cap` is too
 * small, DSMIL_E_INVALID for bad arguments.  Thread-safe (the caller formats row blocks in parallel). */
int64_t dsmil_csv_format_f32(const float* x, int64_t rows, int64_t cols, int64_t row_stride, int32_t decimals, char* out, int64_t cap);
/* ... and the way back (train_tcga.py:27-32 `pd.read_csv(path)`, then `torch.tensor(..., dtype=torch.float32)`): text = the
 * DATA rows of a feature file (the caller skips the header line), `cols` fields per row -> out float32 [max_rows, cols].  HOST
 * function, thread-safe (the caller parses chunks that end at a line break in parallel).  A '%.4f' field parses to the double
 * pandas' parser gives (one correctly rounded division by 10^4) and is cast to float32 as torch casts it; exponents, 'inf',
 * 'nan', long fields go through strtod; an empty field is NaN; blank lines are skipped.  Returns the rows parsed, or
 * DSMIL_E_INVALID (a field that is not a number, a row of another width, more than max_rows rows: let pandas read the file). */
int64_t dsmil_csv_parse_f32(const char* text, int64_t nbytes, int64_t cols, float* out, int64_t max_rows);
/* The loader's file reads (compute_feats.py:21-56: DataLoader workers open every tile file) — HOST function, thread-safe: n files
 * (paths: NUL-terminated strings back to back, path_off[i] = offset of path i) back to back into `out`; with out == NULL only
 * their sizes (the caller sizes the buffer from the returned total).  sizes[i] = bytes of file i, -1 when it cannot be opened or
 * read.  Returns the total bytes, DSMIL_E_WORKSPACE when `cap` is too small, DSMIL_E_INVALID for bad arguments.  One call per
 * group of files from a Python thread holds no interpreter lock: 25 us of interpreter time per file otherwise, serialised. */
int64_t dsmil_read_files(const char* paths, const int64_t* path_off, int32_t n, uint8_t* out, int64_t cap, int64_t* sizes);

const char* dsmil_strerror(int code);
int dsmil_abi_version(void);
/* Rows per workgroup the launcher picks for the dominant kernel (k_query_attend). */
int dsmil_agg_tile_rows(int32_t n_bags, int64_t total_rows);

/* Measurement hooks (bench.py's roofline leg; no reference counterpart).  While enabled, every
 * launch of a dominant kernel is bracketed by hipEventRecord on its own launch stream (up to 4096
 * launches per channel: 0 = k_query_attend of the aggregator, 1 = k_conv of the embedder);
 * dsmil_profile_collect() synchronises on them, returns the summed kernel time and the launch
 * count of one channel, and resets it.  Not for use under graph capture. */
int dsmil_profile_enable(int on);
int dsmil_profile_collect(int channel, double* total_ms, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* DSMIL_HIP_H */
