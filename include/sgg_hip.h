/* sgg_hip.h -- C ABI of libsgg_hip.so: the MI355X (gfx950) scene-graph hot path.
 *
 * Drop-in boundary for the SGCls / PredCls IMP forward of bknyaz/sgg.  Every entry point takes raw
 * DEVICE pointers, sizes and a hipStream_t (as void*), allocates nothing, never synchronises and returns
 * 0 or a negative SGG_ERR_* code.  The caller (sgg_amd/, a Python host over torch for memory and streams)
 * owns every buffer.  Citations are file:line in the reference tree (/root/reference).
 *
 * Element types: SGG_F32 (fp32 storage, fp32 MFMA/VALU arithmetic -- the 1e-3 parity mode), SGG_BF16
 * (bf16 storage, fp32 accumulate -- the wording of BASELINE config 2) and SGG_F16 (IEEE half storage, fp32
 * accumulate: same kernels and rates as bf16 with 8x less rounding error -- the default throughput mode).
 *
 * Feature-map / RoI-feature layout is channels-last (NHWC): the same logical tensors the reference holds
 * as NCHW, handed to Python as permuted views.
 */
#ifndef SGG_HIP_H_
#define SGG_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGG_ABI_VERSION 14   /* bump whenever a prototype below changes: tests/abi.lock pins (version, digest of the prototypes) */

enum { SGG_F32 = 0, SGG_BF16 = 1, SGG_F16 = 2,
       /* a PAIR of f16 planes x = hi + lo (hi = f16(x), lo = f16(x - hi): 22 significand bits) -- the operand / activation format of the x3
        * mode (fp32-grade products on the 16-bit matrix cores: hi.hi + hi.lo + lo.hi accumulated in fp32, the MFMA loop walks the three
        * plane combinations itself).  A row of K values is [hi (K_pad) | lo (K_pad)], K_pad = K rounded up to 64; an NHWC pixel of C
        * channels is [hi (C) | lo (C)].  Accepted ONLY where an entry point says so (sgg_gemm / sgg_gemm_splitk / sgg_gemm_addrows /
        * sgg_conv3x3_relu as operand and output type, sgg_maxpool2x2, sgg_split2, sgg_conv1_1 as output type). */
       SGG_PAIR16 = 3 };
enum { SGG_ACT_NONE = 0, SGG_ACT_RELU = 1 };
enum {
    SGG_OK = 0,
    SGG_ERR_ARG = -1,      /* bad size / alignment / null pointer          -> ValueError  */
    SGG_ERR_DTYPE = -2,    /* unsupported element type                     -> TypeError   */
    SGG_ERR_LAUNCH = -3,   /* hipLaunch failed                             -> RuntimeError */
    SGG_ERR_CAPACITY = -4, /* caller-provided output capacity too small    -> ValueError  */
    SGG_ERR_SPAN = -5      /* a GEMM / conv operand spans >= 4 GiB (the kernels address rows as uniform base + 32-bit lane offset):
                              split the rows over several calls (sgg_amd.ops.gemm / conv3x3_relu do)  -> ValueError */
};

int sgg_abi_version(void);
const char* sgg_build_info(void);

/* ---- a-1  detector transform: [3P] GeneralizedRCNNTransform called at sgg_models/rel_model_base.py:183 ----
 * One image: (x-mean)/std, bilinear resize to (rh,rw) (align_corners=False; identity when (rh,rw)==(h,w)),
 * written into the zero-bordered NHWC4 fp32 plane out[(Hp+2),(Wp+2),4] of batch slot b (border and the pad
 * region rh..Hp / rw..Wp must be zero: the caller memsets the buffer once). */
int sgg_image_prep(const float* img_chw, int h, int w, int rh, int rw, float* out_nhwc4, int b, int Hp, int Wp,
                   void* stream);
/* f-2: the same from the decoded image itself, u8 [h0,w0,3] (RGB, HWC): SquarePad to S = max(h0,w0) with the fill
 * colour int(mean*256) (dataloaders/image_transforms.py:8-13) and ToTensor's /255 (dataloaders/visual_genome.py:264-266)
 * are fused in, so the host hands over 1 byte per sample instead of 4.  (rh,rw) = resized size of the S x S image. */
int sgg_image_prep_u8(const uint8_t* img_hwc, int h0, int w0, int rh, int rw, float* out_nhwc4, int b, int Hp, int Wp,
                      void* stream);
/* The whole batch in one launch: host arrays of n device image pointers and per-image (h0, w0) source sizes, (rh, rw) resized
 * sizes and a u8 flag (1: decoded u8 [h0,w0,3] as sgg_image_prep_u8, 0: f32 [3,h0,w0] as sgg_image_prep); image k goes to plane k. */
int sgg_image_prep_batch(const void* const* imgs, const int* h0, const int* w0, const int* rh, const int* rw,
                         const unsigned char* is_u8, int n, float* out, int Hp, int Wp, void* stream);

/* ---- a-2  VGG-16 features: [3P] vgg16.features minus the last pool, rel_model_base.py:92-93,184,310-312 ----
 * Activations live in zero-bordered NHWC buffers [B, H+2*pad, W+2*pad, C]. */
int sgg_conv1_1(const float* in_nhwc4, const float* w /*[64][27] (ky,kx,c)*/, const float* bias, void* out, int B,
                int H, int W, int out_dtype, void* stream);
/* conv1_1 + ReLU + conv1_2 + ReLU [+ MaxPool2d(2)] in ONE launch (16-bit modes; vgg16.features[0:4] / [0:5], rel_model_base.py:310-312): conv1_1's
 * 64-channel full-resolution output is computed per tile inside LDS and never written.  in_nhwc4 as for sgg_conv1_1; w2 [64][3][3][64];
 * w1_frags: conv1_1's weights [64][27] (k = (ky*3+kx)*3 + c) as MFMA fragments, 4096 bytes written by sgg_conv1_pack_weights (once per
 * weight change; 16-byte aligned, like b1). */
int sgg_conv1_pack_weights(const float* w1 /*[64][27]*/, void* frags /*4096 bytes*/, int dtype, void* stream);
int sgg_conv1_block(const float* in_nhwc4, const void* w1_frags, const float* b1, const void* w2, const float* b2, void* out, int out_pad,
                    int B, int H, int W, int pool, int dtype, void* stream);
/* The x3 mode's 3x3 convolution (pad 1) + bias + ReLU [+ MaxPool2d(2)] on PAIR planes (SGG_PAIR16), patch kernel form: `in` [B, H+2, W+2, 2 Cin]
 * zero-bordered with pixel = [hi (Cin) | lo (Cin)]; `w3` f16 [Cout][9][3 Cin] with tap = [hi | lo | hi] (sgg_split3 mode 1 of the [Cout * 9, Cin]
 * weight matrix); `out` [B, H+2p, W+2p, 2 Cout] (pool = 1: [B, H/2+2p, W/2+2p, 2 Cout]), hi / lo made from the fp32 accumulator after bias,
 * ReLU and the pool.  H, W >= 64, Cout % 128 == 0 and Cin % 32 == 0 (conv_pp.hip) or Cout % 64 == 0 and Cin % 64 == 0 (conv_spatial.hip),
 * else SGG_ERR_ARG (use sgg_conv3x3_relu with dtype SGG_PAIR16).
 * Replaces: [3P] vgg16.features' conv + ReLU (+ pool) pairs at fp32 grade (sgg_models/rel_model_base.py:184,310-312). */
int sgg_conv3x3_relu_x3(const void* in, const void* w3, const float* bias, void* out, int out_pad, int B, int H, int W, int Cin, int Cout,
                        int pool, void* stream);

/* pool = 1: the following MaxPool2d(2) is fused into the epilogue -- out is the pooled plane [B, H/2+2p, W/2+2p, Cout]
 * (H, W even; LDS-patch kernel only: returns SGG_ERR_ARG for shapes that kernel does not take).  out_dtype: element type of `out`;
 * != dtype (the x3 mode: f16 split operands in, f32 out) runs on the implicit-GEMM kernels, without the fused pool. */
int sgg_conv3x3_relu(const void* in /*pad 1*/, const void* w /*[Cout][3][3][Cin]*/, const float* bias, void* out,
                     int out_pad, int B, int H, int W, int Cin, int Cout, int pool, int dtype, int out_dtype, void* stream);
int sgg_maxpool2x2(const void* in /*pad 1*/, void* out, int out_pad, int B, int H, int W, int C, int dtype,
                   void* stream);

/* ---- a-3  pair indexing ----
 * eval: RelModelBase.get_rel_inds, rel_model_base.py:147-163 (row-major nonzero of same-image, off-diagonal,
 *       optionally IoU>0 (bbox_overlaps, lib/pytorch_misc.py:60)).  rel_inds i64[cap,3]=(img,subj,obj).
 * train: proposal_assignments_gtbox, lib/proposal_assignments_gtbox.py:7-80, no-sampling path: every same-image
 *       ordered pair, label = predicate of the FG relation on it (one row per FG relation) else 0, sorted by
 *       (img,subj,obj).  gt_rels i64[R,4]=(img,subj_local,obj_local,pred); img_first i32[num_im] = first box of
 *       each image.  rel_labels i64[cap,4].
 * `count` (device int32) receives the number of rows; rows beyond cap are not written and the call still
 * returns SGG_OK (the host compares count with cap).  work = int32 scratch of >= N+2+N*N (train) / N+2 (eval). */
int sgg_pair_index_eval(const int64_t* im_inds, const float* boxes /*[N,4] or NULL*/, int N, int require_overlap,
                        int64_t* rel_inds, int cap, int* count, int* work, void* stream);
int sgg_pair_index_train(const int64_t* im_inds, int N, const int64_t* gt_rels, int R, const int* img_first,
                         int64_t* rel_labels, int cap, int* count, int* work, void* stream);
/* sgdet training: the tables lib/rel_assignments.py:60-76 builds per image, for the whole batch in one launch (the
 * sampling that follows draws from numpy's RandomState in the reference and stays on the host, in that order).
 * det_boxes f32[N,4], det_img i64[N], det_labels i64[N]; gt_boxes f32[G,4], gt_classes i64[G,2]=(img,cls).
 * gt_iou f32[N,G]: box_iou(det, gt) in torchvision's fp32 operation order, -1 where the images differ;
 * match u8[N,G] = same image & same class & IoU >= fg_thresh (:61); poss u8[N,N] = same image, both labels != 0 and
 * 0 < IoU < 1 (filter_non_overlap, :66) or i != j (:69-71). */
int sgg_rel_assign_tables(const float* det_boxes, const int64_t* det_img, const int64_t* det_labels, int N,
                          const float* gt_boxes, const int64_t* gt_classes, int G, float fg_thresh,
                          int filter_non_overlap, float* gt_iou, uint8_t* match, uint8_t* poss, void* stream);
/* CSR lists of the edges by subject node (out_ptr/out_ids) and by object node (in_ptr/in_ids); edges keep ascending
 * order inside a node.  ptr i32[N+1], ids i32[E].  im_inds (optional, i64[N] node->image): when given, rel_inds must be
 * sorted by image (as both pair-index calls emit it) and only the node's own image segment is scanned.
 * so (optional) i32[E,2] = (subject, object) per edge; flags (optional) i32[1]: flags[0] = 1 iff the edge list is sorted
 * by subject (out_ids is the identity). */
int sgg_edge_csr(const int64_t* rel_inds /*[E,3]*/, int E, int N, const int64_t* im_inds, int* out_ptr, int* out_ids,
                 int* in_ptr, int* in_ids, int* so, int* flags, void* stream);

/* ---- a-4  RoIAlign (+ fused union box): RelModelBase.node_edge_features, rel_model_base.py:245-260, and
 * [3P] torchvision roi_align(output_size=7, sampling_ratio=2, aligned=False, spatial_scale) ----
 * fmap [B,H,W,C] NHWC.  rois f32[Nroi,5]=(img,x1,y1,x2,y2).  pairs == NULL: out[r] = align(rois[r]), R = Nroi.
 * pairs i64[R,2]: out[r] = align(union(rois[p0], rois[p1])) (rel_model_base.py:248-250).
 * add_ec (optional, f32[R,C]): out[r,c,ph,pw] += add_ec[r,c]  (the broadcast add of lib/get_union_boxes.py:101,
 * fused).  out [R,C,P,P] -- the reference's layout, so fc6 consumes it with un-permuted weights. */
int sgg_roi_align_fwd(const void* fmap, int B, int H, int W, int C, const float* rois, int Nroi, const int64_t* pairs,
                      int R, float spatial_scale, int P, int sampling, const float* add_ec, void* out, int dtype,
                      void* stream);
/* Backward of the call above into the feature map (needed where fmap requires grad: the GAN feature-augmentation path,
 * main.py:141): d_fmap f32[B,H,W,C] += RoIAlign^T(d_out) -- a gather (one wave per feature-map cell walks the RoIs in
 * ascending order: one writer per cell, fixed summation order, NO atomics: bit-reproducible); the caller zeroes d_fmap (or accumulates
 * the node and the union-box call into one map).  Same rois / pairs / scale arguments as the forward.  layout 0: d_out is the
 * forward's [R,C,P,P]; layout 1: channels-last [R,P,P,C] (sgg_permute_ncp_to_npc; 16-byte loads -- what the host wrapper passes). */
int sgg_roi_align_bwd(const void* d_out, int B, int H, int W, int C, const float* rois, int Nroi, const int64_t* pairs, int R,
                      float spatial_scale, int P, int sampling, float* d_fmap, int dtype, int layout, void* stream);

/* ---- a-5  union-mask raster ----
 * raster 0 (edge_model 'motifs', the default): draw_union_boxes, lib/draw_rectangles/draw_rectangles.pyx:12-67 -- coverage of
 *   each box inside the pair's union box.  raster 1 (edge_model 'raw_boxes'): draw_union_boxes_grid,
 *   lib/get_union_boxes.py:69-116 -- each box drawn in image coordinates normalised by im_wh f32[B,2] = (w, h) of its image
 *   (rois[:,0] selects the row), through [P] F.grid_sample(ones, bilinear, zeros padding, align_corners=False).
 * rois f32[N,5], pairs i64[E,2] -> out f32[E,2,P,P] + offset (the caller's `- 0.5`, lib/get_union_boxes.py:67,80). */
int sgg_union_rects_fwd(const float* rois, const int64_t* pairs, int E, int P, float offset, float* out, int raster,
                        const float* im_wh, void* stream);
/* Same raster, emitted directly as the 4 stride-16 7x7 patches the (typo'd) conv stack reads
 * (lib/get_union_boxes.py:40-43,52): out[E*4, Kpad] with k = c*49+ky*7+kx, zero padded, raster-0.5 inside. */
int sgg_union_rect_patches(const float* rois, const int64_t* pairs, int E, int P, void* out, int Kpad, int raster,
                           const float* im_wh, int dtype, void* stream);
/* MaxPool2d(3,2,1) over the 2x2 map = max over 4 consecutive rows (lib/get_union_boxes.py:55). in[E*4,C] -> out[E,C] */
int sgg_max4_rows(const void* in, void* out, int E, int C, int dtype, void* stream);
/* x[r,c,p] += add[r,c] in place (lib/get_union_boxes.py:101 when the add is not fused in RoIAlign). */
int sgg_bcast_add(void* x, const float* add_rc, int R, int PP, int C, int dtype, void* stream);

/* ---- a-7  dense projections (nn.Linear): rel_model_stanford.py:29-37,103-107; rel_model_base.py:110-111 ----
 * C[M,N] = post_scale * act(A[M,K] . W[N,K]^T + bias) + post_shift.   A may be split along K in two pieces:
 * A[:, :K1] from A (lda) and A[:, K1:] from A2 (lda2) (K1 == K when A2 is NULL).  K, K1 multiples of 64 (bf16)
 * / 32 (f32); lda, lda2, ldw multiples of 8 elements; 16-byte aligned bases.  bias/post_* are f32[N] or NULL.
 * W2 (optional, needs A2): the weights of the second K segment as their own [N, K-K1] matrix (row stride ldw2);
 * when NULL, W holds all K columns. */
int sgg_gemm(const void* A, int lda, const void* A2, int lda2, int K1, const void* W, int ldw, const void* W2, int ldw2,
             const float* bias,
             const float* post_scale, const float* post_shift, void* C, int ldc, int M, int N, int K, int act,
             int in_dtype, int out_dtype, void* stream);

/* Split-K form for short-M contractions (fc6 on the object rows: M = 32B): `splits` workgroups per output tile reduce
 * K ranges into workspace f32[splits, M, N]; one reduce + epilogue pass writes C[M,N] (row stride ldc, a multiple of 8; N % 8 == 0). */
int sgg_gemm_splitk(const void* A, int lda, const void* W, int ldw, const float* bias, const float* post_scale,
                    const float* post_shift, void* C, int ldc, int M, int N, int K, int act, int in_dtype, int out_dtype, int splits,
                    float* workspace, void* stream);

/* Weight-gradient contraction without transposed operand copies: C[N, K] = A[Mred, N]^T . B[Mred, K], bf16 in, f32 / bf16 out
 * (d W = dY^T X of every nn.Linear on the path, main.py:118).  The reduction rows are staged as they lie and reach the MFMA
 * through ds_read_b64_tr_b16.  Mred % 64 == 0, N % 128 == 0, K % 128 == 0; row strides in elements, multiples of 8.
 * splits > 1: split over the reduction rows, workspace f32[splits, N, K].  Whole 256 x 256 output tiles, 128 of them or more, and
 * splits == 1 run on the ping-pong kernel's TN form (below).
 * sgg_gemm_tn256: that form behind its own entry -- N % 256 == 0, K % 256 == 0, ANY Mred (the last Mred mod 32 reduction rows pass
 *   through pad_ws, 64 (N + K) bytes, as a zero-padded K-tile; pad_ws may be NULL when Mred % 32 == 0), optionally with the group addend
 *   of sgg_gemm_groupadd in the epilogue: C[n][k] += gadd[n][(k + col0) / group] (gadd NULL: none).  The fc6 weight gradient of
 *   loss.backward() (main.py:118) over the unordered box pairs reads dY and the pooled features as they lie through this entry. */
int sgg_gemm_tn(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int Mred, int N, int K, int in_dtype /* SGG_BF16 or SGG_F16 */,
                int out_dtype, int splits, float* workspace, void* stream);
int sgg_gemm_tn256(const void* A, int lda, const void* B, int ldb, const float* gadd, int ld_gadd, int group, int col0, void* C, int ldc,
                   int Mred, int N, int K, int in_dtype /* SGG_BF16 or SGG_F16 */, int out_dtype, void* pad_ws, void* stream);

/* ---- a-8  IMP gather / gate / scatter: RelModelStanford.message_pass, rel_model_stanford.py:74-91 ----
 * The four gates of an edge e = (s, o) are sigmoids of Linear(2H, 1) on [vertex ; edge] (:41-45, 78-89): separable, so they arrive as
 * DOT PRODUCTS made by the kernels that write the state rows (sgg_gru_gate_fwd / sgg_gru_gate_proj_fwd, `dots`):
 *   g_k(e) = sigmoid(node_dots[node, k] + edge_dots[e, k] + gate_b[k]),  k = 0 sub_vert (node s), 1 obj_vert (node o), 2 out_edge (s), 3 in_edge (o).
 * The edge inputs e_in[e] = g_sub v[s] + g_obj v[o] (:78-81) are never formed: they are consumed only by the edge GRU's input projection,
 * W_ih e_in[e] = g_sub (W_ih v[s]) + g_obj (W_ih v[o]), so the nodes are projected once (P = v W_ih^T, sgg_gemm) and
 * sgg_gru_gate_proj_fwd forms the pre-activations from P[s], P[o] (a-9 below).  What this section exports is the step's READ stream,
 *   out[0][n] = sum_{s(e)=n} g_a(e) x[e],   out[1][n] = sum_{o(e)=n} g_b(e) x[e],   (a, b) = (pair, pair + 1)
 * (sum_ctx: their sum in out[0], out is [N,H]) -- the two dense one-hot [N,E] matmuls of :60-66, 91 as segmented sums:
 *   pair 2, x = e_i: the context ctx = ctx_out + ctx_in (the node GRU consumes the halves as a K-split GEMM operand);
 *   pair 0, x = d_gi (gradient of the edge GRU's input pre-activations): the gradient of P in the backward.
 * Every row of x is read once when the host promises (img_ptr from sgg_graph_ptr, max_nodes <= 64, max_edges <= sgg_imp_sliced_capacity)
 * graphs grouped by image with edges sorted by (graph, subject) -- both pair-index calls emit that: a workgroup owns (graph, 64-byte slice
 * of the rows); from sgg_imp_ctx_mfma_min_units() (graph, 128-byte slice) units on, 16-bit graphs of <= 32 nodes / <= 1024 edges run as a
 * block-sparse gate-matrix product on the matrix cores (persistent workgroups, LDS-DMA ring; gates rounded to the rows' format).
 * Without the promise (img_ptr NULL) the CSR lists of sgg_edge_csr are walked: any edge list, every row read twice.
 * No atomics anywhere: results are bit-reproducible. */
int sgg_imp_sliced_capacity(int H, int dtype);
int sgg_imp_ctx_mfma_min_units(void);
int sgg_imp_ctx_fwd(const void* x /*[E,H]*/, const int* so /*[E,2] (sgg_edge_csr)*/, const int* out_ptr, const int* out_ids, const int* in_ptr,
                    const int* in_ids, const int* img_ptr /*or NULL*/, int B, int N, int E, int H, const float* node_dots /*[N,4]*/,
                    const float* edge_dots /*[E,4]*/, const float* gate_b /*[4]*/, int pair /*0 or 2*/, void* out /*[2,N,H] or [N,H]*/,
                    int max_edges, int max_nodes, int sum_ctx, int dtype, void* stream);
/* ---- unordered box pairs (edge_pairs.hip).  The union box of (subject, object) and (object, subject) is one box: its RoIAlign row and
 * the long part of fc6 on it (sgg_models/rel_model_base.py:245-260, rel_model_stanford.py:104 compute both per edge) are computed
 * once per unordered pair {i < j} of an image, slot u = ubase[img] + i (2n - i - 1) / 2 + (j - i - 1) (image-local i, j; n boxes).
 * pair_slots: rel_inds i64[E,3] (image, subject, object: global box indices) + per-image first box / first slot / box count ->
 *   e2u i32[E] (slot of every edge) and u2e i32[U,2] (the at most two edges of every slot, -1 = none; filled in arrival order);
 *   ucount i32[U] is scratch; *flag: bit 0 = an edge outside its image's boxes, bit 1 = more than two edges on a pair.
 * gemm_addrows: sgg_gemm with an f32 row add_rows[add_idx[m]] (add_idx NULL: row m) added before bias and activation.
 * transpose_pairsum: out [C, ld_out] (x's element type) with out[c][u] = x[a][c] + x[b][c], (a, b) = u2e[u] in ascending order; columns >= U are 0.
 * pairsum: the same sums as rows, out [U, ld_out] with out[u][c] = x[a][c] + x[b][c] (C % 8 == 0): the operand of sgg_gemm_tn256.
 * group_bcast_add: y[m][j] += r[m][(j + col0) / group], j < ncol  (the rect term's share of fc6's weight gradient);
 * gemm_groupadd: the same addend in the epilogue of sgg_gemm (no bias, no activation). */
int sgg_pair_slots(const int64_t* rel_inds, const int* first, const int* ubase, const int* cnt, int E, int B, int U, int* e2u, int* u2e,
                   int* ucount, int* flag, void* stream);
int sgg_gemm_addrows(const void* A, int lda, const void* W, int ldw, const float* bias, const float* add_rows, int ld_add,
                     const int* add_idx, void* C, int ldc, int M, int N, int K, int act, int in_dtype, int out_dtype, void* stream);
int sgg_transpose_pairsum(const void* x, int64_t ldx, const int* u2e, void* out, int64_t ld_out, int U, int C, int dtype, void* stream);
int sgg_pairsum(const void* x, int64_t ldx, const int* u2e, void* out, int64_t ld_out, int U, int C, int dtype, void* stream);
int sgg_group_bcast_add(void* y, int64_t ldy, const float* r, int64_t ldr, int M, int ncol, int group, int col0, int dtype, void* stream);
int sgg_gemm_groupadd(const void* A, int lda, const void* W, int ldw, const float* gadd, int ld_gadd, int group, int col0, void* C, int ldc,
                      int M, int N, int K, int in_dtype, int out_dtype, void* stream);

/* ---- glue of the ResNet-50-FPN feature extractor (GQA configuration: sgg_models/rel_model_base.py:58-81; the convolutions themselves
 * run on sgg_gemm / sgg_conv3x3_relu).  NHWC, C a multiple of 8 except where noted.
 * sgg_im2col: patch matrix of a k x k / stride / pad convolution, row = output pixel, columns (ky, kx, c) zero-filled to Kp; src is a
 *   plane [B, H+2 src_pad, W+2 src_pad, Ca] of which the first C channels are used (any C; f32 -> bf16 conversion allowed).
 * sgg_col2im: the adjoint of sgg_im2col on a dense plane (src_pad 0, Ca = C): d_src[B,H,W,C] = gathered sums of d_cols[B*Ho*Wo, Kp] -- the
 *   input gradient of a convolution run as patch matrix x GEMM (the GAN's generator / discriminators, augment/gan.py:74-160, crn.py:64-142).
 * sgg_maxpool3x3s2: MaxPool2d(3, stride 2, padding 1) on [B,H,W,C] -> [B,(H-1)/2+1,(W-1)/2+1,C].
 * sgg_plane_copy: dst[b,y,x,:] = src[b, y*stride, x*stride, :] between planes with borders src_pad / dst_pad (interiors only).
 * sgg_add_relu: y = max(y + x, 0).
 * sgg_upsample_add: y[B,H,W,C] += nearest-neighbour upsampling of top[B,Ht,Wt,C] to (H,W) -- the top-down join of the feature pyramid
 *   ([3P] FeaturePyramidNetwork: F.interpolate(mode='nearest') + add; the detector of sgdet with backbone='resnet50'). */
int sgg_im2col(const void* src, int B, int H, int W, int Ca, int C, int src_pad, int k, int stride, int pad, int Ho, int Wo, void* dst,
               int Kp, int src_dtype, int dst_dtype, void* stream);
int sgg_col2im(const void* d_cols, int B, int H, int W, int C, int k, int stride, int pad, int Ho, int Wo, int Kp, void* d_src, int dtype,
               void* stream);
int sgg_maxpool3x3s2(const void* in, void* out, int B, int H, int W, int C, int dtype, void* stream);
int sgg_plane_copy(const void* src, int Hs, int Ws, int src_pad, void* dst, int Hd, int Wd, int dst_pad, int B, int C, int stride, int dtype,
                   void* stream);
int sgg_add_relu(void* y, const void* x, int64_t n, int dtype, void* stream);
int sgg_upsample_add(void* y, const void* top, int B, int H, int W, int Ht, int Wt, int C, int dtype, void* stream);
/* img_ptr i32[2*(B+1) + 66*B]: img_ptr[b] = first node of graph b (im_inds i64[N] ascending), img_ptr[B] = N; then
 * img_ptr[B+1+b] = out_ptr[first node of b] = first edge of graph b (out_ptr from sgg_edge_csr, same stream, edges sorted);
 * then per graph 66 graph-relative out-list offsets of its nodes (entries past the last node repeat the edge count). */
int sgg_graph_ptr(const int64_t* im_inds, int N, int B, const int* out_ptr, int* img_ptr, void* stream);

/* ---- a-9  GRU cell pointwise part: nn.GRUCell, rel_model_stanford.py:36-37,71-72,83,92 ----
 * gi = x W_ih^T + b_ih, gh = h W_hh^T + b_hh come from sgg_gemm ([M,3H], gate order r,z,n).
 * gh == NULL means h == 0: gh = b_hh (f32[3H]) and h_prev = 0 (first call, :68-72).
 * g_dtype = element type of gi/gh (f32 pre-activations may feed bf16 states), dtype = type of h_prev/h_out. */
int sgg_gru_gate_fwd(const void* gi, const void* gh, const float* b_hh, const void* h_prev, void* h_out, int M, int H,
                     const float* dot_w, int dot_ld, float* dots, int g_dtype, int dtype, void* stream);
/*   dots (optional, f32[M,4]): dots[m,k] = dot_w[k*dot_ld : k*dot_ld+H] . h_out[m,:] (on the stored, rounded values) -- the
 *   vertex / edge halves of the four gate pre-activations of the NEXT message-passing step (:78-89), made while the row is
 *   in registers so that no IMP kernel ever needs a whole row.  Needs H/8 a power of two <= 64. */
/* The edge GRU of a message-passing iteration, e_{i+1} = GRU_e(e_in, e_i) (:83), from the node projection instead of e_in rows:
 *   gi[e] = g_sub(e) P[s] + g_obj(e) P[o] + b_ih  with P = v_i W_ih^T f32[N,3H] (no bias), gates from node_dots / edge_dots / gate_b (a-8);
 *   gh [M,3H] (gh_dtype) = e_i W_hh^T + b_hh;  h_prev = e_i;  so i32[M,2] = (subject, object) node rows.  dots as above. */
int sgg_gru_gate_proj_fwd(const void* gh, const float* P, const float* b_ih, const int* so, const float* node_dots, const float* edge_dots,
                          const float* gate_b, const void* h_prev, void* h_out, int M, int H, const float* dot_w, int dot_ld, float* dots,
                          int dtype, int gh_dtype /* SGG_F32, or dtype: gh straight out of the GEMM's epilogue in the state's 16-bit type */,
                          void* stream);

/* ---- a-11  eval tail: rel_model_stanford.py:183-207 + filter_dets, lib/surgery.py:17-55 ----
 * obj: softmax over C classes, best class in 1..C-1 and its prob (sgcls/sgdet), or score 1 / given class (predcls
 *   when gt_classes != NULL).  rel: softmax over P predicates, triple score = max_{p>=1} * s_subj * s_obj.
 * sort: descending by score, ties by ascending edge index (torch.sort leaves ties unspecified).
 * Outputs: obj_scores f32[N], obj_preds i64[N], rels i64[E,2] (sorted), pred_scores f32[E,P] (sorted).
 * work: f32/i32 scratch of >= 2*pow2ceil(E) + E*P words. */
int sgg_eval_tail(const void* obj_dists, int N, int C, const void* rel_dists, int E, int P,
                  const int64_t* rel_inds /*[E,3]*/, const int64_t* gt_classes /*[N] or NULL*/, float* obj_scores,
                  int64_t* obj_preds, int64_t* rels, float* pred_scores, void* work, int dtype, void* stream);

/* ---- a-12  SGDet front end after the backbone: [3P] torchvision RPN + RoIHeads in eval mode, called at
 * sgg_models/rel_model_base.py:210-213.  Dense parts use sgg_conv3x3_relu / sgg_gemm / sgg_roi_align_fwd. ----
 * rpn_decode: head f32[B*Hf*Wf, ldh] (cols [0,A) objectness, [A,5A) deltas a*4+c) + base anchors f32[A,4] ->
 *   boxes f32[B,Hf*Wf*A,4] (BoxCoder weights 1,1,1,1, not clipped), scores = raw objectness, anchors in (y,x,a) order.
 * segmented_sort_desc: stable descending sort of each segment (rocPRIM); vals_out = position inside the segment.
 * gather_topk: first `take` sorted entries per segment -> clipped boxes, scores, labels, valid (finite score, w,h >= min_size).
 * nms: greedy NMS on score-ordered boxes [B,n,4], IoU > thresh suppresses (class-aware when labels != NULL).
 * compact_rois: kept boxes of all images -> rois f32[total,5], offsets i32[B+1].
 * det_candidates: pred f32[K,ldp] (cols [0,C) logits, [C,5C) regression c*4+k) -> per (roi, class>=1) softmax score
 *   (-inf unless > thresh and box >= min_size), decoded (weights 10,10,5,5) clipped box, label.
 * det_output: gather of the kept candidates -> boxes f32[B,max_keep,4], scores, labels i64. */
int sgg_rpn_decode(const float* head, int ldh, const float* base_anchors, int A, int Hf, int Wf, float stride_y, float stride_x,
                   int B, float* boxes, float* scores, void* stream);
int sgg_segmented_sort_desc(const float* keys_in, float* keys_out, int* vals_tmp, int* vals_out, int n, int nseg,
                            const int* seg_off, int seg_len_hint, void* temp, size_t* temp_bytes, void* stream);
int sgg_gather_topk(const float* keys_sorted, const int* vals_sorted, const int* seg_off, const float* boxes,
                    const int* labels_in, const float* img_hw, int B, int take, float min_size, float* out_boxes,
                    float* out_scores, int* out_labels, unsigned char* valid, void* stream);
/* The two calls above in one, for take <= 4096: the `take` best-scoring candidates of every segment, best first, equal scores by lower
 * index (the order of the stable sort), gathered / clipped / validated like sgg_gather_topk -- a 4-pass radix SELECT over the segment's
 * scores + an LDS bitonic sort of the selected ones, one workgroup per image; nothing that is not wanted is sorted. */
int sgg_topk_gather(const float* scores, const int* seg_off, const float* boxes, const int* labels_in, const float* img_hw, int B,
                    int take, float min_size, float* out_boxes, float* out_scores, int* out_labels, unsigned char* valid,
                    void* stream);
int sgg_nms(const float* boxes, const int* labels, const unsigned char* valid, int B, int n, float thresh, int max_keep,
            void* mask_ws, int* keep_idx, int* keep_cnt, void* stream);
int sgg_compact_rois(const float* boxes, const int* keep_idx, const int* keep_cnt, int B, int n, int max_keep, float* rois,
                     int* offsets, void* stream);
int sgg_det_candidates(const float* pred, int ldp, const float* rois, int K, int C, const float* img_hw, float score_thresh,
                       float min_size, float* cand_score, float* cand_box, int* cand_label, void* stream);
int sgg_det_output(const float* boxes, const float* scores, const int* labels, const int* keep_idx, const int* keep_cnt, int B,
                   int n, int max_keep, float* out_boxes, float* out_scores, int64_t* out_labels, void* stream);

/* ---- training side of the trainable relation head (main.py:100-120: forward in train mode, backward) ----
 * Dense gradient contractions reuse sgg_gemm on transposed operands.  All *_bwd calls are stream-ordered like the
 * forward ones; buffers documented "zeroed by the callee" are cleared with hipMemsetAsync on the same stream. */
/* nn.Dropout(p), in place, counter-based RNG (element i keeps iff hash(seed,i) >= p*2^32): rel_model_base.py:110-111 */
int sgg_dropout_fwd(void* x, int64_t n, float p, uint64_t seed, int dtype, void* stream);
/* the same mask as sgg_dropout_fwd(seed = seed_dev[0] * 4 + salt), the step's seed read from device memory (u64[1]) at run time: a launch
 * captured into a hipGraph (sgg_amd/graph_step.py) draws a new mask on every replay */
int sgg_dropout_fwd_dev(void* x, int64_t n, float p, const uint64_t* seed_dev, uint64_t salt, int dtype, void* stream);
/* dx = dy * (y > 0) * scale : backward of ReLU (scale 1) / ReLU->Dropout (y = saved post-dropout output, scale 1/(1-p)) */
int sgg_act_bwd(const void* dy, const void* y, void* dx, int64_t n, float scale, int g_dtype, int y_dtype, void* stream);
/* Reductions of this section are TWO-STAGE: row blocks write partial rows into a caller-provided f32 workspace `ws`, a second launch
 * adds them in a fixed order -- no float atomics, so a training step is bit-reproducible.  ws sizes are stated per entry.
 * Cross-entropy of logits f32[M,C] (row stride ld) against labels i64 (element stride label_stride), 'baseline' form of
 * lib/losses.py:41-43,74: loss[0] (+)= weight / norm[0] * sum_rows CE (accumulate = 0: overwritten -- the first head; `norm` on the device), and
 * grad[M,ldg] (g_dtype, columns >= C zero) = grad_scale * d loss / d logits (grad_scale: the loss scale of the f16 mode, 1 otherwise)
 * -- one launch for what F.cross_entropy + autograd do in ~12.  ws: f32[(M + 3) / 4].  flag (optional, i32[1], never cleared here):
 * bit 0 is raised when a label lies outside [0, C) (e.g. torch's ignore_index); such a row adds no loss and gets a zero gradient.
 * mode 0: 'baseline' as above.  mode 1 'dnorm' / mode 2 'dnorm-fgbg' (lib/losses.py:44-63, the density-normalised edge losses): `norm`
 * is f32[2] = (M_FG, M_BG) ON THE DEVICE (sgg_label_counts; summed over the ranks by the caller), a row's weight is
 * weight * (label > 0 ? (M_FG > 0 ? alpha / M_FG : 1) : mode 1 ? (M_BG > 0 && M_FG > 0 ? beta / M_FG : 1) : (M_BG > 0 ? beta / M_BG : 1))
 * -- the reference's edge_weights incl. its "stay 1" branches, chosen per row on the device: no host synchronisation. */
int sgg_ce_fwd_bwd(const float* logits, int ld, const int64_t* labels, int label_stride, int M, int C, const float* norm,
                   float weight, float grad_scale, float* loss, int accumulate, void* grad, int ldg, float* ws, int* flag, int g_dtype,
                   int mode, float alpha, float beta, void* stream);
/* counts[0] (+)= #{labels > 0}, counts[1] (+)= #{labels == 0} as f32 (M_FG, M_BG of lib/losses.py:29-34; M < 2^24). */
int sgg_label_counts(const int64_t* labels, int label_stride, int M, float* counts, int accumulate, void* stream);
/* out[N] = column sums of x[M,N] (row stride ld): bias gradients.  ws: f32[64 * N] (may be NULL when M <= 512). */
int sgg_colsum(const void* x, int M, int N, int ld, float* out, float* ws, int dtype, void* stream);
/* train-mode BatchNorm2d of the rect conv (lib/get_union_boxes.py:54,58) on row-major [rows, C] activations:
 * bn_stats: sums[2][C] = (sum x, sum x^2), ws f32[64 * 2C]; bn_finalize: batch mean / invstd, the affine
 * (scale, shift) and the running-stat update (momentum 0.01, unbiased var); bn_apply: y = x*scale+shift, optionally
 * followed by the max over 4 consecutive rows (MaxPool2d(3,2,1) on the 2x2 map, :55) with arg-max rows to `arg`;
 * bn_bwd: backward of ReLU -> BN [-> max4]: x = post-ReLU BN input, dy [rows(/4), C] -> dx [rows, C] (gradient at the
 * conv output), sums[2][C] = (dbeta, dgamma), ws f32[64 * 2C] (phases 0 and 1).
 * Synchronised statistics across data-parallel ranks (BatchNorm row): the caller all-reduces `sums` plus a
 * row count between the reduction and its use; count_dev (optional, device f32) then overrides `count` / `rows` as the
 * divisor.  bn_bwd phase: 0 = reduce + apply in one call, 1 = reduce only (local sums out), 2 = apply only (sums in). */
int sgg_bn_stats(const void* x, int M, int C, float* sums, float* ws, int dtype, void* stream);
int sgg_bn_finalize(const float* sums, int C, int count, const float* count_dev, const float* gamma, const float* beta,
                    float eps, float momentum, float* run_mean, float* run_var, float* mean, float* invstd, float* scale,
                    float* shift, void* stream);
int sgg_bn_apply(const void* x, const float* scale, const float* shift, void* out, unsigned char* arg, int rows_out, int C,
                 int max4, int dtype, void* stream);
int sgg_bn_bwd(const void* dy, const unsigned char* arg, const void* x, const float* mean, const float* invstd,
               const float* gamma, void* dx, float* sums, float* ws, int rows, int C, int max4, int phase, const float* count_dev,
               int dtype, void* stream);
/* nn.GRUCell backward, pointwise part: from dh[M,H] and the saved fp32 pre-activations gi/gh ([M,3H]; gh NULL = the
 * h=0 first call, b_hh given) -> d_gi, d_gh [M,3H] and dh_prev [M,H] (may be NULL). */
int sgg_gru_gate_bwd(const void* dh, const float* gi, const float* gh, const float* b_hh, const void* h_prev, void* d_gi,
                     void* d_gh, void* dh_prev, int M, int H, int dtype, void* stream);
/* backward of sgg_gru_gate_proj_fwd: from dh[M,H], the saved gh and P (recomputes gi and the cell) -> d_gi, d_gh [M,3H], dh_prev [M,H],
 * and dq f32[M,2] = (d_gi[e] . P[s], d_gi[e] . P[o]): the gradients of the scalar gates g_sub, g_obj before their sigmoids' derivative. */
int sgg_gru_gate_proj_bwd(const void* dh, const void* gh, const float* P, const float* b_ih, const int* so, const float* node_dots,
                          const float* edge_dots, const float* gate_b, const void* h_prev, void* d_gi, void* d_gh, void* dh_prev,
                          float* dq, int M, int H, int dtype, int gh_dtype, void* stream);
/* backward of one message-passing step's gates and context sums (rel_model_stanford.py:76-91), e = (s, o):
 * edge side: d_e[e] += g_out*d_ctx[s] + g_in*d_ctx[o] + sum_k da_k*w_k[H:];  da[E,4] = gate pre-activation gradients (order sub,obj,out,in):
 *   da_sub = dq[e,0] g_sub(1-g_sub), da_obj = dq[e,1] g_obj(1-g_obj), da_out = (d_ctx[s].e) g_out(1-g_out), da_in = (d_ctx[o].e) g_in(1-g_in).
 * node side: d_v[n] += sum_k S_k[n]*w_k[:H] with S_sub/S_out = sums of da[.,0]/da[.,2] over n's out-edges, S_obj/S_in = sums of da[.,1]/da[.,3]
 *   over its in-edges; nsum[N,4] = (S_sub, S_obj, S_out, S_in).  (The gate-weighted sums of d_gi -- the gradient of P -- are sgg_imp_ctx_fwd
 *   with pair 0.)
 * gate-weight gradients: d_w_k[H:] = rank4_reduce(da, e_i), d_w_k[:H] = rank4_reduce(nsum, v), d_b = colsum(da). */
int sgg_imp_edge_ctx_bwd(const void* e, const int* so, int E, int H, const float* node_dots, const float* edge_dots, const float* gate_w,
                         const float* gate_b, const float* dq, const void* d_ctx, void* d_e, float* da, int dtype, void* stream);
int sgg_imp_node_gates_bwd(const float* da, const int* out_ptr, const int* out_ids, const int* in_ptr, const int* in_ids,
                           const float* gate_w, int N, int H, void* d_v, float* nsum, int dtype, void* stream);
/* out[k,:H] (+)= sum_r a[r,k]*x[r,:], k<4 (out row stride out_ld; accumulate = 0: overwritten).  ws: f32[64 * 4 * H] */
int sgg_rank4_reduce(const float* a, const void* x, int R, int H, float* out, int out_ld, float* ws, int accumulate, int dtype, void* stream);

/* optimiser step of main.py:119-120: global-norm gradient clip (lib/pytorch_misc.py:625-656) + torch.optim.SGD
 * (momentum, weight decay, lib/pytorch_misc.py:144), fused and sync-free: sqnorm_acc accumulates sum(g^2) over all
 * parameters into one device float (accumulate = 0 on the first call of a step overwrites it: no clearing launch); sgd_step reads it: coef = min(1, max_norm/(sqrt(norm_sq)*grad_scale + 1e-6)),
 * g' = coef*grad_scale*g + wd*p, buf = first ? g' : momentum*buf + g', p -= lr*buf.  p, buf fp32; g fp32 or 16-bit.
 * sqnorm: ws f32[2048] (one partial per workgroup, added in a fixed order). */
int sgg_sqnorm_acc(const void* g, int64_t n, float* acc, float* ws, int accumulate, int dtype, void* stream);
int sgg_sgd_step(float* p, const void* g, float* momentum_buf, int64_t n, float lr, float weight_decay, float momentum,
                 int first_step, const float* norm_sq, float max_norm, float grad_scale, int g_dtype, void* stream);

/* Multi-tensor forms of the two calls above: host arrays (length count) of device pointers and sizes, one launch per
 * 32 tensors instead of one per parameter.  Pointers 16-byte aligned (tensors of fewer than 4 elements: any alignment).  lr per tensor (the reference's two
 * parameter groups, lib/pytorch_misc.py:135-144).  shadow: optional array (entries may be NULL) of 16-bit buffers (shadow_dtype:
 * SGG_BF16 / SGG_F16) that receive the updated parameter in the same pass -- the next forward's MFMA operand, so no separate cast pass.
 * A non-finite *norm_sq (overflowed scaled gradients of the f16 mode, a NaN batch) skips the update: no parameter, momentum or shadow is
 * touched and skipped[0] (optional i32 device counter; hand it to ONE call per step) is incremented. */
int sgg_sqnorm_multi(const void* const* g, const int64_t* n, int count, float* acc, float* ws /*[2048]*/, int accumulate, int dtype, void* stream);
int sgg_sgd_multi(float* const* p, const void* const* g, float* const* momentum_buf, void* const* shadow,
                  const int64_t* n, const float* lr, int count, float weight_decay, float momentum, int first_step,
                  const float* norm_sq, float max_norm, float grad_scale, int g_dtype, int shadow_dtype,
                  int max_blocks /* 0: default 512 */, int* skipped, void* stream);

/* x f32 [rows, K] (row stride ldx) -> the PAIR form (SGG_PAIR16): out f16 [rows, 2 * K_pad] = [hi (K_pad) | lo (K_pad)], hi = f16(x),
 * lo = f16(x - hi), columns K .. K_pad - 1 zero.  K_pad % 8 == 0 (the GEMMs want % 64), ldo >= 2 K_pad.  The operand format of the x3 mode
 * since round 6 (sgg_split3's [hi | hi | lo] rows are kept for the two-segment contractions).  Replaces: nothing in the reference (fp32
 * cuBLAS products, sgg_models/rel_model_stanford.py:97-107); it is how this library gets fp32-grade products out of the 16-bit matrix cores. */
int sgg_split2(const float* x, int64_t ldx, int64_t rows, int K, int K_pad, void* out, int64_t ldo, void* stream);

/* The x3 mode's operand form (a fast mode inside the 1e-3 parity clause): x f32 [rows, K] (row stride ldx) -> out f16
 * [rows, 3 K_pad] (row stride ldo), x = hi + lo with hi = f16(x), lo = f16(x - hi).  mode 0 (activations): [hi | hi | lo]; mode 1
 * (weights): [hi | lo | hi] -- one f16 MFMA contraction over the 3 K_pad columns then accumulates hi.hi + hi.lo + lo.hi in fp32 (22
 * significand bits per operand; the lo.lo term is dropped).  K_pad % 8 == 0 (the GEMMs want % 32), columns K .. K_pad - 1 are zero. */
int sgg_split3(const float* x, int64_t ldx, int64_t rows, int K, int K_pad, void* out, int64_t ldo, int mode, void* stream);

/* ---- utilities used by the host for weight preparation (load time, not on the step path) ---- */
int sgg_cast(const void* in, void* out, int64_t n, int in_dtype, int out_dtype, void* stream);
/* out[n][p][c] = in[n][c][p]  (fc6 K-order (c,ph,pw) -> (ph,pw,c); conv OIHW -> O(HW)I) */
int sgg_permute_ncp_to_npc(const void* in, void* out, int Nn, int C, int Pp, int in_dtype, int out_dtype, void* stream);

/* out[c][r] = in[r][c] (+ add[r][c / group], add f32 with row stride ld_add, or NULL); row strides in elements.
 * Feeds d W = dY^T X to sgg_gemm; the add form builds (edge_feat + conv(rects))^T for fc6's weight gradient.
 * colsum (optional, f32[C]): column sums of `in` (+add) -- the bias gradient of the same dY; colsum_ws: f32 scratch of >= ceil(R / 64) * C
 * floats (one partial row per row block, summed in ascending order: no atomics, bit-reproducible). */
int sgg_transpose(const void* in, int64_t ld_in, void* out, int64_t ld_out, int R, int C, const float* add, int64_t ld_add,
                  int group, float* colsum, float* colsum_ws, int in_dtype, int out_dtype, void* stream);
/* n <= 16 plain 16-bit transposes in ONE launch: out[i] [C_i rows, row stride ld_out_i] = in[i] [R_i, C_i]^T (row stride ld_in_i); columns past R_i of
 * an output row are not written.  Arrays of n host-side entries.  The W^T copies of the training backward's dX contractions (dX = dY W as an NT GEMM on
 * W^T), rebuilt after every optimiser update: no reference counterpart (cuBLAS takes a transpose flag), scheduling only. */
int sgg_transpose_multi(const void* const* in, const int64_t* ld_in, void* const* out, const int64_t* ld_out, const int* R, const int* C, int n,
                        int dtype, void* stream);

/* out[n][c] = sum_{p<group} in[n][c*group + p]  (fp32 in): fc6's folded columns W6sum[n,c] = sum_p W6[n,c,p] */
int sgg_group_sum(const float* in, int64_t ld_in, void* out, int64_t ld_out, int Nn, int C, int group, int out_dtype,
                  void* stream);
/* y += x (n multiple of 8) */
int sgg_add(void* y, const void* x, int64_t n, int y_dtype, int x_dtype, void* stream);
/* ---- f-1: scene-graph recall matching (lib/sgg_eval.py:280-417 -- evaluate_recall, _triplet, _compute_pred_matches), batched over
 * images.  gt_trip i32[G,3] / pred_trip i32[P,3] = (class_subj, predicate, class_obj); gt_box / pred_box f32[.,8] = (subject box,
 * object box) xyxy; gt_img i32[G] = image of each GT triplet; pred_ptr i32[B+1] = CSR offsets of each image's predictions, which
 * are in rank order (descending triple score).  A prediction matches a GT triplet when the three classes are equal and both
 * boxes have IoU >= iou_thresh (phrdet: the union boxes, :393-401).  first_rank[g] = rank inside its image of the first
 * matching prediction, INT32_MAX if none -- R@K = #{g : first_rank[g] < K} / G (the reference's |union(pred_to_gt[:K])| / G).
 * Optional per-triplet ranks (:236-272): gt_pair i32[G,2] / pred_pair i32[P,2] = box indices; pair_rank[g] = rank of the first
 * match among the predictions on the same box pair in either direction, -1 if none. */
int sgg_recall_first_match(const int32_t* gt_trip, const float* gt_box, const int32_t* gt_img, int G, const int32_t* pred_trip,
                           const float* pred_box, const int32_t* pred_ptr, int B, const int32_t* gt_pair,
                           const int32_t* pred_pair, float iou_thresh, int phrdet, int32_t* first_rank, int32_t* pair_rank,
                           void* stream);

/* ---- a-10 (optional flags -use_bias / -test_bias): FrequencyBias, lib/sparse_targets.py:26-31 (index_with_labels) as used at
 * sgg_models/rel_model_stanford.py:159-177.  obj_preds i64[N] (out) = best class in 1..C-1 of softmax(obj_dists [N,C], `dtype`),
 * or gt_classes i64[N] when that is not NULL (predcls, :166-167; obj_dists may then be NULL).  rel_out f32[E,P] =
 * table[obj_preds[subj]*C + obj_preds[obj], :] (+ rel_in f32[E,P] unless NULL = the test_bias form, :173-174); table f32[C*C,P] is
 * the embedding weight.  row_idx i32[E] (optional out) = the table row of each edge, the input of the backward:
 * d_table[row_idx[e], :] += d_out[e, :] (d_table zeroed by the caller; d rel_in = d_out). */
int sgg_freq_bias_fwd(const void* obj_dists, int N, int C, const int64_t* gt_classes, const int64_t* rel_inds /*[E,3]*/, int E,
                      const float* table, int P, const float* rel_in, float* rel_out, int64_t* obj_preds, int32_t* row_idx,
                      int dtype, void* stream);
int sgg_freq_bias_bwd(const float* d_out, const int32_t* row_idx, int E, int P, float* d_table, void* stream);

/* ---- f-4 (next row, started)  GAN generator data movement: augment/layout.py, augment/graphconv.py ----
 * boxes_to_layout (layout.py:33-71 with _boxes_to_grid :102-136 and _pool_samples :139-166), channels-last:
 *   vecs [O,S,S,D] object patches (S = 0: [O,D] vectors, expanded to a constant 8x8 patch as at :57-58), boxes f32[O,4] in [0,1]
 *   (x0,y0,x1,y1), obj_img i32[O] = image of each object, out [N,H,W,D] = per image the sum in ascending object order (avg = 1:
 *   mean over the image's objects) of the bilinear (zeros padding, align_corners=False) resampling of every patch into its box.
 *   bwd: d_vecs [O,S,S,D] (or [O,D]) from d_out [N,H,W,D]; counts i32[N] = objects per image (read only when avg = 1);
 *   H, W <= 256. */
int sgg_boxes_to_layout_fwd(const void* vecs, const float* boxes, const int* obj_img, int N, int O, int S, int D, int H, int W,
                            int avg, void* out, int dtype, void* stream);
int sgg_boxes_to_layout_bwd(const void* d_out, const float* boxes, const int* obj_img, const int* counts, int O, int S, int D,
                            int H, int W, int avg, void* d_vecs, int dtype, void* stream);
/* GraphTripleConv (graphconv.py:51-119): out[t] = [obj[s_t] | pred[t] | obj[o_t]] (:68-78), width 2*Din+De; edges i64[T,2]. */
int sgg_triple_gather(const void* obj, const void* pred, const int64_t* edges, int T, int Din, int De, void* out, int dtype,
                      void* stream);
/* pooled[n] = sum of rows[t, 0:Hd] over triples with s_t = n + sum of rows[t, o_off:o_off+Hd] over triples with o_t = n, divided
 * by the number of those triples when avg = 1 (clamped at 1) (:93-115); CSR lists as produced by sgg_edge_csr.
 * bwd writes only the two Hd-wide column blocks of d_rows [T, ld]. */
int sgg_triple_pool_fwd(const void* rows, int ld, int o_off, const int* out_ptr, const int* out_ids, const int* in_ptr,
                        const int* in_ids, int O, int Hd, int avg, void* pooled, int dtype, void* stream);
int sgg_triple_pool_bwd(const void* d_pooled, const int64_t* edges, const int* out_ptr, const int* in_ptr, int T, int Hd, int avg,
                        int ld, int o_off, void* d_rows, int dtype, void* stream);

/* ---------------------------------------------------------------- (e) the exchange step: gradient all-reduce over RCCL
 * The data-parallel train step's one collective (main.py:116-120 run on N GPUs) for a host that is not torch: thin entry points
 * over ncclGetUniqueId / ncclCommInitRank / ncclAllReduce(sum) / ncclCommDestroy of the RCCL the process carries (looked up with dlopen at
 * the first call: SGG_RCCL_LIB, then librccl.so; no link-time dependency).  This repo's Python host uses torch.distributed('nccl') -- the
 * same library -- through sgg_amd/dist.py.  One communicator per process = per GPU (the current HIP device). */
int sgg_allreduce_unique_id(void* id128);                                             /* rank 0: 128 bytes to hand to every rank */
int sgg_allreduce_init(const void* id128, int world, int rank, void** comm);
int sgg_allreduce_sum(void* comm, void* buf, int64_t n, int dtype, void* stream);     /* in place, SGG_F32 / SGG_BF16 / SGG_F16 */
int sgg_allreduce_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif /* SGG_HIP_H_ */
