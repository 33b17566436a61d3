/* C ABI of libfacenet_hip.so -- the MI355X (gfx950) drop-in boundary for the
 * sMedX/FaceNet training/inference hot path.
 *
 * The reference has no FFI of its own: its hot path is the set of TensorFlow /
 * Keras ops called from facenet/models/inception_resnet_v1.py, facenet/facenet.py,
 * facenet/statistics.py and apps/train_softmax.py.  Every entry point below
 * cites the reference call site(s) whose arithmetic it replaces.
 *
 * Conventions: plain pointers and sizes only; every pointer is DEVICE memory
 * owned by the caller (activations NHWC, low precision = bf16 or f16 selected
 * by `dtype`; parameters/gradients fp32); kernels never allocate; every call
 * is asynchronous on `stream` (a hipStream_t) and is HIP-graph capturable;
 * return 0 on success, negative on error with text in fn_last_error()
 * (thread-local).  No global mutable state: calls are re-entrant.
 */
#ifndef FACENET_HIP_H
#define FACENET_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define FN_BF16 0
#define FN_F16 1

/* Order-independent accumulators.  Sums that many workgroups contribute to -- BatchNorm batch statistics, the BatchNorm-backward
 * sums, bias gradients, the losses -- are 64-bit FIXED-POINT integers: every workgroup converts its fp32 partial sum once (round
 * to nearest) and adds it with an integer atomic, so the total has the same bits whatever order the workgroups arrive in and a
 * training step is reproducible bit for bit (fp32 atomics were not).  value = integer * 2^-bits.  Forward statistics use
 * FN_ACC_STAT_BITS (|sum| < 8.8e12), gradient sums FN_ACC_GRAD_BITS (|sum| < 8.4e6, resolution 9e-13).  Callers zero them
 * (all-zero bits) and read them back with fn_acc_to_float or by scaling on the host. */
typedef int64_t fn_acc_t;
#define FN_ACC_STAT_BITS 20
#define FN_ACC_GRAD_BITS 40

const char* fn_last_error(void);
int fn_abi_version(void);

/* ---- convolution as implicit GEMM on MFMA ------------------------------------
 * Replaces tf.keras.layers.Conv2D (+ the BatchNormalization statistics pass, ReLU,
 * bias, tf.concat and the residual "net += scale * up" that surround it):
 * inception_resnet_v1.py:90-138,160-193,215-248,269-299,316-367,387-430 (Conv2D),
 * :141-148,196-202,251-257 (concat + residual), :304-306,372-375 (concat),
 * and the Dense layers at :462 and apps/train_softmax.py:57-63 (a 1x1 conv on [N,1,1,C]).
 * The descriptor always states the FORWARD geometry of the layer. */
typedef struct fn_conv_desc {
    int32_t N, H, W, Cin;      /* forward input  [N,H,W,Cin]  */
    int32_t OH, OW, Cout;      /* forward output [N,OH,OW,Cout] */
    int32_t KH, KW, stride, pad_h, pad_w;
    int32_t dtype;             /* FN_BF16 | FN_F16 : activation + packed-weight storage */
    int32_t ld_x, ld_y;        /* channel stride (elements) of the x-side and y-side buffers (concat-free slices) */
    int32_t relu;              /* fwd epilogue: max(.,0) */
    int32_t accumulate;        /* fwd/dgrad epilogue: add to what `y`/`dx` already holds */
    int32_t out_f32;           /* fwd: y (dgrad: dx) is fp32 instead of low precision */
    int32_t ld_res;            /* channel stride of resid */
    float scale;               /* fwd: y = resid + scale*(conv + bias) when resid != NULL, else conv + bias */
    int32_t splits;            /* wgrad: split-K factor over pixels (0 = library picks) */
    int32_t stats_sq_off;      /* fwd: element offset from the sum array to the sum-of-squares array in `stats` */
    int32_t stats_replicas;    /* fwd: number of accumulator replicas (row tile t adds into replica t % replicas); 0/1 = one */
    int32_t stats_rep_stride;  /* fwd: element stride between replicas */
    const void* x;             /* fwd/wgrad: input activations; dgrad: unused */
    const void* w;             /* fwd: packed [Cout][KH*KW*Cin]; dgrad: transposed pack [Cin][KH*KW*Cout] */
    void* y;                   /* fwd: output; dgrad/wgrad: dY (read) */
    void* dx;                  /* dgrad: output dX [N,H,W,ld_x] */
    float* dw;                 /* wgrad: fp32 [Cout][KH*KW*Cin], atomically accumulated (caller zeroes) */
    const float* bias;         /* fwd: per-Cout fp32 or NULL (BN-folded shift / `up` bias) */
    fn_acc_t* stats;           /* fwd: NULL or fixed point (FN_ACC_STAT_BITS): stats[c] += sum, stats[stats_sq_off + c] += sum of squares (BN batch statistics) */
    const void* resid;         /* fwd: residual trunk [N,OH,OW,ld_res] or NULL */
    /* dgrad: optional fused reduction of the BatchNorm backward of the layer that produced x (see fn_bn_relu_train_bwd):
     * bn_y = that layer's raw output (same slice as dx), bn_acc[rep*stride + c] += sum dyh, [.. + bn_sq_off + c] += sum dyh*xhat */
    const void* bn_y;
    const float* bn_scale;
    const float* bn_shift;
    const float* bn_beta;
    fn_acc_t* bn_acc;          /* fixed point, FN_ACC_GRAD_BITS */
    int32_t ld_bn_y, bn_sq_off, bn_replicas, bn_rep_stride, bn_relu;
    /* fwd / wgrad: "normalise on load".  When nrm_stats is set, x is the RAW output of a BatchNormalization(center only)+ReLU
     * layer whose batch statistics (sum | sum of squares, replicated like `stats`) have been accumulated by the producing
     * fn_conv2d_fwd; the operand the convolution sees is relu((x - mean) * rstd + beta), computed per element while the tile
     * is staged (zero padding stays zero).  The activated tensor is then never written.  Index 0 of nrm_stats / nrm_beta
     * is channel 0 of x; nrm_count = N*H*W of x.  Cin <= 512.  fn_bn_finalize publishes the same scale / shift for the
     * backward pass and updates the moving statistics. */
    const fn_acc_t* nrm_stats; /* as `stats` */
    const float* nrm_beta;
    int32_t nrm_sq_off, nrm_replicas, nrm_rep_stride, nrm_count;
    float nrm_eps;
    /* fwd, optional with nrm_stats: the convolution also MATERIALISES the activated tensor it normalises -- nrm_z has the
     * geometry of x (same ld_x); every element is written exactly once, by the workgroups of the first Cout tile while they
     * stage the centre tap.  Needs stride 1 and OH x OW == H x W (1x1, or 'same' padding).  This replaces the
     * fn_bn_relu_train_fwd launch of the producing layer; the weight gradient then reads nrm_z like any activation. */
    void* nrm_z;
    /* fwd / dgrad tile variant: 0 = library heuristic, BM*1000+BN with BM, BN in {128, 64, 32} = caller's choice (the host side
     * times the candidates once per plan: facenet_amd/engine.py autotune), 9000000 = the halo-tile kernel (3x3, stride 1,
     * channels a multiple of 8; FN_EUNSUPPORTED otherwise) also where the heuristic would not pick it.  Results do not depend
     * on the tile beyond the summation order. */
    int32_t tile_fwd, tile_dgrad;
    /* dgrad of SIBLING 1x1 stride-1 layers that read the same x (inception towers branching off one trunk): dX = y-gradient of
     * this layer times its transposed pack PLUS the same for up to two more layers (dy2/w2, dy3/w3; their Cout and dy row
     * stride), computed as one GEMM whose K runs through the sources -- one launch and one pass over dX instead of a chain of
     * read-modify-write accumulations.  dy2 == NULL: ordinary single layer. */
    const void* dy2;
    const void* w2;
    const void* dy3;
    const void* w3;
    int32_t Cout2, ld_y2, Cout3, ld_y3;
    /* dgrad, optional: the RESIDUAL BACKWARD of the block whose output is x, fused into this launch's epilogue (it is the last
     * producer of that output's gradient; inception_resnet_v1.py:145-148,199-202,254-257: out = act(trunk + scale*(up + bias))).
     *   total = this data gradient + rb_prev        rb_prev : gradient x has received so far (geometry of dx), or NULL
     *   g     = rb_out ? total * (rb_out > 0) : total rb_out  : the block's forward output (ReLU mask), NULL = no activation
     *   rb_dtrunk (+)= g                             gradient of the block's trunk input; rb_accumulate: add to what it holds
     *   rb_dup = rb_scale * g                        gradient of the block's `up` convolution output
     *   rb_dbias[c] += sum over pixels of rb_dup     (fp32, the `up` bias gradient)
     * All four tensors have dx's geometry (ld_x).  dx itself is not written.  rb_dup == NULL: ordinary data gradient. */
    const void* rb_prev;
    const void* rb_out;
    void* rb_dtrunk;
    void* rb_dup;
    fn_acc_t* rb_dbias;        /* fixed point, FN_ACC_GRAD_BITS (fn_acc_to_float moves the bias gradients into the fp32 gradient buffer) */
    float rb_scale;
    int32_t rb_accumulate;
    /* fwd, optional: PReLU with one slope per output channel (Keras PReLU(shared_axes=[1,2]) after Conv2D, PReLU() after Dense:
     * the P/R/O-Net layers of the MTCNN detector behind detectors/face_detector.py:63-78): y = v > 0 ? v : prelu[c] * v,
     * applied to conv + bias.  Not combined with resid / relu / accumulate. */
    const float* prelu;
} fn_conv_desc;

/* One Inception-ResNet-B block ("Block17", inception_resnet_v1.py:153-204) of the BN-FOLDED inference network in ONE launch:
 * 1x1 | 1x1 -> 1x7 -> 7x1 towers, concat, `up` 1x1 + bias, scaled residual add, ReLU.  One workgroup per image; the tower
 * activations stay in LDS, only weights stream (csrc/block_fused.hip).  x, y: [N,8,8,896] low precision (x != y); weights: the
 * inference packs [Cout][taps][Cin] of the five layers (fn_fold_bn); b_t*: the folded BatchNorm shifts, b_up: the `up` bias.
 * Equals the five fn_conv2d_fwd launches it replaces up to the summation order. */
int fn_block17_infer(const void* x, void* y, int N, const void* w_t0, const void* w_t1a, const void* w_t1b, const void* w_t1c, const void* w_up,
                     const float* b_t0, const float* b_t1a, const float* b_t1b, const float* b_t1c, const float* b_up, float scale, int relu,
                     int dtype, void* stream);
/* fn_block17_infer plus 64 extra workgroups (on CUs the one-image-per-workgroup launch leaves idle) that read
 * [warm, warm + warm_bytes) into every XCD's L2: the bytes the NEXT launch streams, normally the next block's weight packs
 * (16-byte aligned, < 2 GiB).  Results are those of fn_block17_infer; warm == NULL (with warm_bytes 0) is fn_block17_infer. */
int fn_block17_infer_warm(const void* x, void* y, int N, const void* w_t0, const void* w_t1a, const void* w_t1b, const void* w_t1c,
                          const void* w_up, const float* b_t0, const float* b_t1a, const float* b_t1b, const float* b_t1c, const float* b_up,
                          float scale, int relu, const void* warm, int64_t warm_bytes, int dtype, void* stream);
/* One Inception-ResNet-A block ("Block35", inception_resnet_v1.py:83-150) of the BN-folded inference network in ONE launch (seven
 * convolution launches otherwise): x, y [N,17,17,256]; w_1x1 / b_1x1: the three tower-entry 1x1 layers (tower_conv0/Conv2d_1x1,
 * tower_conv1/Conv2d_0a_1x1, tower_conv2/Conv2d_0a_1x1); w_3x3 / b_3x3: tower_conv1/Conv2d_0b_3x3, tower_conv2/Conv2d_0b_3x3,
 * tower_conv2/Conv2d_0c_3x3; w_up / b_up: the `up` layer [256][96] and its bias.  Arrays of 3 device pointers (host arrays). */
int fn_block35_infer(const void* x, void* y, int N, const void* const* w_1x1, const void* const* w_3x3, const void* w_up,
                     const float* const* b_1x1, const float* const* b_3x3, const float* b_up, float scale, int relu, int dtype, void* stream);
/* fn_block35_infer plus the warm-ahead workgroups of fn_block17_infer_warm (same contract for warm / warm_bytes). */
int fn_block35_infer_warm(const void* x, void* y, int N, const void* const* w_1x1, const void* const* w_3x3, const void* w_up,
                          const float* const* b_1x1, const float* const* b_3x3, const float* b_up, float scale, int relu,
                          const void* warm, int64_t warm_bytes, int dtype, void* stream);
int fn_conv2d_fwd(const fn_conv_desc* d, void* stream);
int fn_conv2d_dgrad(const fn_conv_desc* d, void* stream);
int fn_conv2d_wgrad(const fn_conv_desc* d, void* stream);
/* Grouped forward / data-gradient convolutions: one launch for n INDEPENDENT layers that share a tile variant
 * (= fn_conv2d_variant(d, op)) and 1x1-ness (plain).  Planned on the host like the grouped weight gradients below. */
int fn_conv2d_arg_bytes(void);
int fn_conv2d_group_build(const fn_conv_desc* descs, int n, int op, int variant, void* host_args, int32_t* host_prefix, int32_t* smem_bytes);
int fn_conv2d_grouped(const void* dev_args, const int32_t* dev_prefix, int n, int total_blocks, int variant, int plain, int smem_bytes,
                      int dtype, void* stream);
/* Grouped weight gradients: one launch for many layers that share a tile variant (= fn_conv2d_variant(d, 2)).
 * fn_conv2d_wgrad_group_build plans on the HOST: it fills host_args (n * fn_conv2d_wgrad_arg_bytes() bytes, opaque) and
 * host_prefix (n+1 workgroup offsets) and returns the total workgroup count; the caller copies both to device memory once and
 * replays fn_conv2d_wgrad_grouped every step (pointers inside the descriptors must stay valid).  Members that normalise x
 * on load (nrm_stats) form groups of their own: pass variant + 1000000 to both calls.  Likewise fn_conv2d_grouped takes
 * plain | 2 for a group whose members all normalise on load. */
int fn_conv2d_wgrad_arg_bytes(void);
int fn_conv2d_wgrad_group_build(const fn_conv_desc* descs, int n, int variant, void* host_args, int32_t* host_prefix, float* ws,
                                int64_t* ws_elems);
int fn_conv2d_wgrad_grouped(const void* dev_args, const int32_t* dev_prefix, int n, int total_blocks, int variant, int dtype, void* stream);
/* The grouped path is deterministic and free of atomics: a layer that is not split over pixels stores dw directly; a split layer
 * stores split z into slab z of the workspace `ws` (fp32 [splits][Cout*K] per layer, laid out by group_build) and
 * fn_conv2d_wgrad_reduce, launched after fn_conv2d_wgrad_grouped with the same dev_args, writes dw = slab 0 + slab 1 + ... in that
 * order.  Call group_build once with ws = NULL to learn *ws_elems (floats), allocate, and call it again with the pointer.  dw
 * needs no zeroing on this path.  (fn_conv2d_wgrad, the single-layer launch, still accumulates atomically into a zeroed dw.) */
int fn_conv2d_wgrad_reduce(const void* dev_args, int n, void* stream);
/* tile variant the descriptor dispatches to (op 0 fwd, 1 dgrad, 2 wgrad): BM*1000+BN; measurement aid only */
int fn_conv2d_variant(const fn_conv_desc* d, int op);

/* ---- input normalisation: facenet/facenet.py:67-86 (ImageProcessing.call) -------
 * u8 NHWC [N,H,W,3] -> low precision [N,H,W,8] (channels 3..7 zero), mode 0 = per-image
 * min/max to [-1,1], mode 1 = per_image_standardization.  `work` = 8*N 32-bit words of scratch (8-byte aligned). */
int fn_image_normalize(const uint8_t* img, void* out, float* work, int N, int HW, int mode, int dtype, void* stream);
int fn_image_normalize_f32(const float* img, void* out, float* work, int N, int HW, int mode, int dtype, void* stream);
/* tf.image.resize(images, [size,size]) of facenet.py:70 (bilinear, half-pixel centres, no antialias): u8 or fp32 NHWC
 * [N,H,W,3] -> fp32 [N,OH,OW,3]; the identity at the configured size, so plans skip it then. */
int fn_image_resize_bilinear(const void* img, int src_is_f32, float* out, int N, int H, int W, int OH, int OW, void* stream);
/* tf.image.resize_with_crop_or_pad(image, size, size) of ImageLoader.__call__ (facenet.py:45-54) for a ragged batch of decoded
 * HWC u8 images packed back to back: image n starts at byte offsets[n] of src and is hw[2n] x hw[2n+1] x 3; centre crop
 * (offset max((h-S)//2, 0)) and centre zero-pad (offset max((S-h)//2, 0)) to dst u8 [N,S,S,3].  Sizes are not checked against
 * the src allocation: the caller owns the packing. */
int fn_crop_or_pad_u8(const uint8_t* src, const long long* offsets, const int32_t* hw, uint8_t* dst, int N, int S, void* stream);
/* gather rows of a u8 image pool by index (triplet batch assembly): out[i] = pool[idx[i]] */
int fn_gather_images(const uint8_t* pool, const int32_t* idx, uint8_t* out, int n_out, int bytes_per_image, void* stream);

/* Batched finalisation for the layers consumed through nrm_* (no fn_bn_relu_train_fwd launch): for every channel c < CB with
 * reps[c] > 0: mean/var from the replicated sums (count[c] elements), save_scale = rstd, save_shift = beta - mean*rstd,
 * moving statistics updated as in fn_bn_relu_train_fwd.  One launch for the whole network. */
int fn_bn_finalize(const fn_acc_t* stats, int sq_off, int rep_stride, const int32_t* reps, const int32_t* count, const float* beta,
                   float* save_scale, float* save_shift, float* moving_mean, float* moving_var, float momentum, float eps, int CB,
                   void* stream);

/* ---- BatchNormalization (center only, no scale; eps 1e-3, momentum 0.99) --------
 * inception_resnet_v1.py:56-63 and every BatchNormalization(**...) line; ReLU() fused.
 * Training: y (raw conv output, channel slice [0,C) of a [M,ld_y] buffer) -> z = relu((y-mean)*rstd+beta)
 * with batch statistics from `stats` (sum,sumsq as written by fn_conv2d_fwd); scale=rstd and
 * shift=beta-mean*rstd are saved for backward and the moving statistics are updated (biased variance). */
int fn_bn_relu_train_fwd(const void* y, int ld_y, void* z, int ld_z, int M, int C, const fn_acc_t* stats, int stats_sq_off, int stats_replicas,
                         int stats_rep_stride, const float* beta,
                         float* save_scale, float* save_shift, float* moving_mean, float* moving_var, float momentum, float eps,
                         int relu, int dtype, void* stream);
/* backward: dz (grad wrt the BN+ReLU output) -> dy (grad wrt the raw conv output y), in place; y is the raw forward
 * conv output (xhat and the ReLU mask are recomputed from it: masked elements still receive the batch-statistic
 * terms).  acc (zeroed by the caller) holds sum dyh at acc[rep*stride + c] and sum dyh*xhat at acc[rep*stride + acc_sq_off + c];
 * with reduced = 0 this call fills it (reduce kernel, replica 0), with reduced = 1 a fn_conv2d_dgrad epilogue already did.
 * dbeta[C] += sum dyh. */
int fn_bn_relu_train_bwd(void* dz, int ld_d, const void* y, int ld_y, int M, int C, const float* beta, const float* save_scale,
                         const float* save_shift, float* dbeta, fn_acc_t* acc, int acc_sq_off, int acc_replicas, int acc_rep_stride,
                         int reduced, int relu, int dtype, void* stream);

/* ---- pooling: MaxPool2D(3, strides=2, 'valid') :301,369,409 ; AvgPool2D([3,3]) + Flatten :460-461 */
/* argmax (optional, u8 [N,OH,OW,C]): scan position 0..8 of the FIRST maximum of every window; when given to the backward it
 * replaces the recomputation from x (x may then be NULL). */
int fn_maxpool3x3s2_fwd(const void* x, int ld_x, void* y, int ld_y, int N, int H, int W, int C, uint8_t* argmax, int dtype, void* stream);
int fn_maxpool3x3s2_bwd(const void* x, int ld_x, const void* dy, int ld_dy, void* dx, int ld_dx, int N, int H, int W, int C,
                        const uint8_t* argmax, int accumulate, int dtype, void* stream);
/* ---- MTCNN face detector (detectors/face_detector.py:63-78 wraps PyPI `mtcnn`; apps/extract_faces.py:30,56) -------------
 * The P/R/O-Net convolutions and dense layers are fn_conv2d_fwd launches (bias + fn_conv_desc.prelu); these three entry points
 * are the rest of the device work.  Arithmetic restated in oracle/mtcnn_oracle.py (parity unpinned: the package and cv2 are
 * not installed).
 * fn_area_resize_crop: for k < n, boxes[4k..] = (ox, oy, cw, ch) is a crop of the uint8 HWC frame (zero outside the frame),
 *   resized to OH x OW like cv2.resize(crop, (OW, OH), interpolation=cv2.INTER_AREA) -- true area resampling when both axes
 *   shrink, cv2's area-mode bilinear otherwise -- then (v - 127.5) * 0.0078125 and stored TRANSPOSED as out[k][x][y][8] low
 *   precision (channels 3..7 zero): the networks were trained on transposed images.  source_is_u8 = 1: the crop is a uint8
 *   image (float accumulators, result rounded half-to-even to uint8: the stage-1 pyramid); 0: the crop is float64 (stages 2, 3:
 *   double accumulators, no rounding; the enlarging path is only built for this case). */
int fn_area_resize_crop(const uint8_t* frame, int H, int W, const int32_t* boxes, int n, int OH, int OW, int source_is_u8, void* out,
                        int dtype, void* stream);
/* MaxPooling2D(pool_size=k, strides=stride, 'valid' or 'same'): pad_h / pad_w = Keras' padding BEFORE (0 for every MTCNN
 * layer), windows are clipped at the border, OH / OW as Keras computes them. */
int fn_maxpool2d_fwd(const void* x, int ld_x, void* y, int ld_y, int N, int H, int W, int C, int k, int stride, int pad_h, int pad_w,
                     int OH, int OW, int dtype, void* stream);
/* Whole-frame pyramid level (source_is_u8 case of fn_area_resize_crop with the window = the frame, same results bit for bit) in
 * two passes, horizontal then vertical -- the order cv2 itself works in; rows: fp32 workspace [H][OW][3].  OH <= H, OW <= W. */
int fn_area_resize_frame(const uint8_t* frame, int H, int W, int OH, int OW, float* rows, void* out, int dtype, void* stream);
/* P-Net map [ncell][ld] fp32 = (logit0, logit1, reg0..3, ...) per cell: p1 = softmax(logits)[1]; cells with p1 >= threshold are
 * appended to cand (16-byte aligned) as 8-float records (cell index as int bits, tag as int bits, p1, reg0..3, 0) in any order;
 * `tag` lets the levels of a pyramid share one buffer.  *counter += number of hits (may exceed max_cand: only records below
 * max_cand are stored); reset_counter: zero it first. */
int fn_mtcnn_candidates(const float* map, long ncell, int ld, float threshold, float* cand, int32_t* counter, int max_cand, int tag,
                        int reset_counter, void* stream);
/* Greedy NMS of the package (__nms): boxes [n][ld] float64 rows (x1, y1, x2, y2, ...), order = np.argsort(scores) (ascending; the
 * best box is order[n-1]), ratio = inter / (area_i + area_j - inter) or, by_min, inter / min(area_i, area_j), all in float64 with
 * the package's operation order; a box survives while every better kept box has ratio <= threshold.  keep[0..*n_keep) = indices
 * of the kept boxes, best first.  workspace: n * ceil(n / 64) * 8 bytes (pairwise bit matrix).
 * _batch: njobs independent jobs in one pair of launches (one workgroup scans each job); boxes / order / keep are concatenated in
 * job order, sizes / thresholds / by_min are HOST arrays of njobs entries, n_keep[j] per job, workspace = sum of the jobs'. */
int fn_nms_greedy(const double* boxes, int ld, const int32_t* order, int n, double threshold, int by_min, void* workspace,
                  long workspace_bytes, int32_t* keep, int32_t* n_keep, void* stream);
int fn_nms_greedy_batch(const double* boxes, int ld, const int32_t* order, const int32_t* sizes, const double* thresholds, const int32_t* by_min,
                        int njobs, void* workspace, long workspace_bytes, int32_t* keep, int32_t* n_keep, void* stream);
int fn_avgpool_fwd(const void* x, void* y, int N, int HW, int C, int dtype, void* stream);
int fn_avgpool_bwd(const void* dy, void* dx, int N, int HW, int C, int dtype, void* stream);

/* ---- residual backward for "net = act(net + scale*up)" (:145-148,199-202,254-257) ------------
 * dpre = dout * (out>0 if relu); dtrunk = dpre (or += when accumulate); dup = scale*dpre; dbias[C] += sum(dup). */
int fn_residual_bwd(const void* dout, const void* out, void* dtrunk, void* dup, fn_acc_t* dbias, int M, int C, float scale, int relu,
                    int accumulate, int dtype, void* stream);

/* dst[i] = src[i] * 2^-bits (the bias gradients leave their fixed-point accumulators for the fp32 gradient buffer) */
int fn_acc_to_float(const fn_acc_t* src, float* dst, long n, int bits, void* stream);

/* ---- embedding head on fp32 [N,E]: BN without ReLU (:467) and tf.nn.l2_normalize (:491-492) ---- */
int fn_head_bn_fwd(const float* y, float* out, int N, int E, const float* beta, float* moving_mean, float* moving_var,
                   float* save_mean, float* save_rstd, int training, float momentum, float eps, void* stream);
int fn_head_bn_bwd(const float* dout, const float* y, const float* save_mean, const float* save_rstd, float* dbeta, void* dy_lp,
                   int N, int E, int dtype, void* stream);
int fn_l2norm_fwd(const float* x, float* out, int N, int E, float eps, void* stream);
int fn_l2norm_bwd(const float* x, const float* dout, float* dx, int N, int E, float eps, void* stream);
int fn_cast_f32_to_lp(const float* x, void* y, long n, int dtype, void* stream);

/* ---- distances / triplets ---------------------------------------------------------------------
 * fn_pairwise_sqdist: facenet/statistics.py:22-57 (pairwise_similarities): metric 0 -> 2(1-a.b) with the
 * dot clipped to [-1,1], metric 1 -> arccos, metric 2 -> |a|^2+|b|^2-2a.b (un-normalised inputs).
 * range[2] receives min/max of the raw dot products (the reference's +-(1+atol) check is done by the caller).
 * Triplet selection / loss are build-defined (arXiv 1503.03832 sec. 3). */
int fn_pairwise_sqdist(const float* xa, const float* xb, float* out, float* range, int n, int m, int E, int metric, void* stream);
int fn_select_triplets(const float* dist, const int32_t* labels, int n, float alpha, int nrof_triplets, uint32_t seed,
                       int semi_hard, int32_t* triplets, int32_t* info, void* stream);
/* loss: fp32[4] -- word 0 receives the loss; words 2-3 are the launch's own fixed-point accumulator (FN_ACC_GRAD_BITS) */
int fn_triplet_loss_fwd_bwd(const float* emb, float* demb, float* loss, int T, int E, float alpha, void* stream);

/* ---- face-to-face validation statistics: facenet/statistics.py:111-138 (ConfidenceMatrix) with the class-balanced
 * weights of SimilarityCalculator.evaluate (:92-103).  emb fp32 [n,E], unit-norm rows grouped by class (class c = rows
 * cls_start[c] .. cls_start[c+1]); thresholds ascending, T <= 256; out fp64 [4*T] = tp | tn | fp | fn;
 * range[2] = ordered-int min/max of the raw dot products (for the reference's +-(1+atol) check, :40-42). */
int fn_confidence_counts(const float* emb, const int32_t* cls_start, int C, int E, const float* thresholds, int T, int metric,
                         double* out, int32_t* range, void* stream);

/* ---- softmax classifier loss: apps/train_softmax.py:91 (SparseCategoricalCrossentropy(from_logits)); loss: fp32[4] as above;
 * dbias (optional): fixed point, FN_ACC_GRAD_BITS, += column sums of dlogits */
int fn_softmax_xent_fwd_bwd(const float* logits, int ld, const int32_t* labels, float* loss, void* dlogits_lp, int ld_d, fn_acc_t* dbias, int N,
                            int C, float grad_scale, int dtype, void* stream);

/* ---- optimiser: tf.keras.optimizers.Adam(epsilon=0.1) apps/train_softmax.py:92 + Keras L2(5e-4) (:65) ----
 * hyper = device word[8] {lr, beta1^t, beta2^t, grad_scale, t (int32: Keras' `iterations`), 3 spare}; fn_adam_tick advances t and
 * re-derives the beta powers from it on the device (graph replay safe; t survives past the fp32 underflow of beta1^t).
 * Elements [0, n_decay) get the coupled L2 term g += 2*l2*w.  w_lp receives the low-precision copy of w[0, n_lp). */
int fn_adam_keras(float* w, const float* g, float* m, float* v, void* w_lp, long n_lp, long n, long n_decay, float* hyper, float beta1,
                  float beta2, float eps, float l2, int dtype, void* stream);
int fn_adam_tick(float* hyper, float beta1, float beta2, void* stream);

/* ---- weight packs ----------------------------------------------------------------------------------
 * table = device int32 [n_layers][8] {w_off, cout, ktot, taps, cin, bn_off(-1 none), fold_bias_off, 0}.
 * fn_pack_transpose: wt[cin][tap][cout] = w[cout][tap][cin] (dgrad operand) for every layer.
 * fn_fold_bn: wf = lp(w * rsqrt(var+eps)[cout]), bias = beta - mean*rsqrt(var+eps)   (facenet/tfutils.py:244-250) */
int fn_pack_transpose(const void* w_lp, void* wt_lp, const int32_t* table, int n_layers, int max_layer_elems, int dtype, void* stream);
int fn_fold_bn(const float* w, void* wf_lp, float* fold_bias, const float* beta, const float* moving_mean, const float* moving_var,
               const int32_t* table, int n_layers, int max_layer_elems, float eps, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif
