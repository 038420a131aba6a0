/*
 * casapose_hip.h -- C ABI of libcasapose_hip.so (gfx950 / MI355X).
 *
 * The reference (fraunhoferhhi/casapose) has no FFI: its hot path bottoms out in
 * TensorFlow / tensorflow-addons ops.  Each entry point below replaces the native
 * work behind one group of reference call sites (cited as file:line relative to the
 * reference tree).  The Python host (casapose_amd/) binds these with ctypes; a
 * maintainer of the reference would bind them the same way.
 *
 * Conventions
 *   - every function returns 0 on success, a negative cp_status otherwise; nothing
 *     throws across the boundary; cp_last_error() returns a thread-local message.
 *   - all pointers are DEVICE pointers unless the name ends in _host; the caller owns
 *     every buffer (inputs, outputs, workspaces); no hidden allocation.
 *   - `stream` is a hipStream_t passed as void* (0 = null stream); calls are
 *     asynchronous with respect to the host and re-entrant across streams.
 *   - tensors are NHWC, fp32 unless stated; label maps are uint8 (0 = background).
 */
#ifndef CASAPOSE_HIP_H
#define CASAPOSE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum cp_status {
    CP_OK = 0,
    CP_ERR_INVALID = -1,   /* bad argument / unsupported shape */
    CP_ERR_LAUNCH = -2,    /* hipLaunch / runtime failure */
    CP_ERR_NO_DEVICE = -3
} cp_status;

const char* cp_last_error(void);
/* ABI version of the library: CP_ABI_VERSION of the header it was built from.  300 (round 3): cp_conv_desc starts with `struct_size`
 * and every entry point that takes a descriptor refuses one whose struct_size differs from the library's sizeof(cp_conv_desc)
 * (CP_ERR_INVALID, message in cp_last_error()) instead of reading fields a shorter or longer caller-side struct does not have.
 * 301 (round 6): + the f16x2 range-guard entry points (cp_f16x2_monitor_set / _get, cp_f16x2_range_check, cp_amax_f32); no struct changed.
 * 302 (round 6): cp_conv_desc ends with head_prefix / head_prefix_n / head_prefix_ld (whole output records from the last fused head). */
#define CP_ABI_VERSION 302
int cp_version(void);
/* sizeof(cp_conv_desc) / sizeof(cp_conv_source) as this library was compiled: a binder checks them against its own declaration at load time */
size_t cp_conv_desc_size(void);
size_t cp_conv_source_size(void);
/* Blocks of the PERSISTENT convolution / GEMM launches (csrc/conv_hsplit.hip, csrc/wino_gemm_split.hip): 256 = one per CU (default; the
 * environment variable CASAPOSE_PERSIST_BLOCKS sets the initial value).  A smaller multiple of 8 leaves whole CUs free, so that an HBM-bound
 * kernel launched on ANOTHER stream runs beside the matrix-pipe kernel instead of behind it -- the two-stream forward of the host layer sets 224
 * around its launches.  Process-wide, read at launch time on the calling thread. */
int cp_set_persistent_blocks(int blocks);
int cp_get_persistent_blocks(void);
/* number of visible gfx950 devices (0 on a CPU-only host); never initialises a context */
int cp_device_count(void);
/* Matrix-pipe probe (measurement aid, bench.py): one launch of a bare MFMA stream on every SIMD -- which = 0: v_mfma_f32_32x32x2_f32,
 * 1: v_mfma_f32_32x32x16_bf16 -- so that the rate the part SUSTAINS under its power limit can be printed beside the datasheet peak the
 * roofline entries use.  ws: cp_mfma_probe_workspace_bytes() of device memory; *flops receives the FLOPs of the launch (host memory). */
size_t cp_mfma_probe_workspace_bytes(void);
int cp_mfma_probe(int which, int iters, void* ws, double* flops, void* stream);

/* ------------------------------------------------------------------------------------
 * Fused implicit-GEMM convolution, forward, fp32 on v_mfma_f32_32x32x2_f32.
 *
 * Replaces: layers.Conv2D in the encoder (casapose/pose_models/models/resnet.py:85-87,
 * 97-99,103,249) and decoder 1 (models/casapose.py:71-74, pose_models.py:546,616);
 * PartialConvolution.calc (models/_normalization_layers.py:325-373);
 * inference-mode SyncBatchNormalization + ReLU / leaky pair (resnet.py:78-79,100-101,
 * 250-251,303-304; casapose.py:77,98-107); ClassAdaptiveWeightedNormalization.calc
 * (_normalization_layers.py:119-139); layers.Add (resnet.py:110); layers.concatenate
 * (pose_models.py:542-545,571,583,595,607); GuidedUpsampling gather
 * (_normalization_layers.py:554-558) and UpSampling2D(bilinear) (casapose.py:135-140)
 * when fused into the consumer's operand load.
 *
 *  out[n,oy,ox,co] = EPI( sum_{s in sources} sum_{ky,kx} sum_c
 *                         A_s(n, oy*stride+ky*dil-pad, ox*stride+kx*dil-pad, c) * Wp[co, k(s,ky,kx,c)] )
 *
 *  A_s(n,y,x,c) is 0 outside [0,in_h)x[0,in_w), else the source value fetched through
 *  the source's spatial mode, optionally (source 0/1 individually) passed through a
 *  per-channel affine `v*pre_scale[c]+pre_shift[c]`, and, when `tap_label` is set,
 *  multiplied by [tap_label[n,y,x] == tap_label[n,oy,ox]] (partial convolution; needs
 *  stride 1 so both live on the same grid).
 *
 *  EPI(v): v *= row_scale[n,oy,ox] (partial-conv 9/count)      if row_scale
 *          v += residual[n,oy,ox,co]                           if residual
 *          out_raw = v                                         if out_raw
 *          t = v*sc + sh ; (sc,sh) = (scale[co],shift[co]) or, if epi_label,
 *                          (scale[epi_label[n,oy,ox]*cout+co], shift[...])   (CLADE table)
 *          t = act(t): 0 none, 1 relu, 2 leaky 0.1 (relu(t)-relu(-0.1t))
 *          out_act = t                                         if out_act
 *
 *  Packed weights Wp: row-major [cout][ktot] fp32, produced by cp_conv_pack_weights
 *  (K order: source 0 chunks, then source 1 chunks; see cp_conv_ktot).
 * ---------------------------------------------------------------------------------- */

enum { CP_SRC_DIRECT = 0, CP_SRC_NEAREST_SEL = 1, CP_SRC_BILINEAR_X2 = 2,
       CP_SRC_ZERO_INSERT_X2 = 3 /* the source holds the EVEN positions of the in_h x in_w grid, the rest is zero:
                                    the data-gradient of a stride-2 convolution (transposed convolution) */ };
enum { CP_ACT_NONE = 0, CP_ACT_RELU = 1, CP_ACT_LEAKY01 = 2 };

typedef struct cp_conv_source {
    const float* data;      /* NHWC, pixel stride `ld` floats                                 */
    int channels;           /* multiple of 32, or exactly 4 ("C4" mode: 8 taps per K chunk)  */
    int ld;                 /* floats between consecutive pixels (>= channels)               */
    int mode;               /* CP_SRC_*: DIRECT reads an in_h x in_w grid; the X2 modes read  */
                            /* an (in_h/2) x (in_w/2) grid                                   */
    const uint8_t* sel;     /* CP_SRC_NEAREST_SEL: [n,in_h,in_w] neighbour index 0..3         */
    const float* pre_scale; /* optional per-channel affine applied to in-bounds values       */
    const float* pre_shift;
} cp_conv_source;

typedef struct cp_conv_desc {
    uint32_t struct_size;       /* = sizeof(cp_conv_desc) of the header the CALLER was built against; checked by every entry point */
    int batch, in_h, in_w;      /* conv-input grid (after any fused x2 upsampling)           */
    int out_h, out_w, cout;
    int kh, kw, stride, dilation, pad;
    int num_sources;            /* 1 or 2 (channel concatenation, source 0 first)            */
    cp_conv_source src[2];
    const float* weights;       /* packed [cout][ktot]                                        */
    const float* weights_halo;  /* optional second packing (cp_conv_pack_weights_halo_host): lets 3x3/s1/p1   */
                                /* layers with cout <= 64 (a multiple of 4; outputs / residual 16-byte aligned  */
                                /* with ld % 4 == 0) run on the LDS-resident halo-tile kernel                 */
    const uint8_t* tap_label;   /* optional [n,in_h,in_w] -> partial-conv tap mask            */
    /* epilogue */
    const float* row_scale;     /* optional [n,out_h,out_w]                                   */
    const float* residual;      /* optional NHWC, pixel stride residual_ld                    */
    int residual_ld;
    const float* scale;         /* optional [cout] or [classes][cout] with epi_label          */
    const float* shift;
    const uint8_t* epi_label;   /* optional [n,out_h,out_w]                                   */
    int act;                    /* CP_ACT_* applied to out_act only                          */
    float* out_raw;             /* optional, pixel stride out_raw_ld                          */
    int out_raw_ld;
    float* out_act;             /* optional, pixel stride out_act_ld                          */
    int out_act_ld;
    int tile_hint;              /* 0 = auto; otherwise a CP_TILE_* value (benchmark / tests) */
    /* optional fused 1x1 head (halo-tile kernel only, cout == 32): head_out[n,y,x,0..head_cout) = sum_c out_act[..,c] *
     * Wh[c][q]; replaces pv_final_conv_segmentation / pv_final_conv_vertex (pose_models.py:546,616).  head_weights from
     * cp_conv_pack_head_weights_host.  With a head, out_act/out_raw may both be NULL (the 32-channel tensor is not stored). */
    const float* head_weights;
    float* head_out;
    int head_cout, head_out_ld;
    /* optional grouped GEMM (the 36 Winograd planes in one launch): output pixels [g*group_rows, (g+1)*group_rows) use the
     * packed weights at weights + g*group_weight_stride floats.  group_rows must be a multiple of 128; 0 = ungrouped. */
    int group_rows, group_weight_stride;
    /* optional, with a fused head: head_label_out[n,y,x] = arg-max over the first head_label_classes head channels of the pixel (uint8; first
     * maximum wins, as cp_argmax_labels) -- the hard label map of pose_models.py:547-554 straight from the head's registers */
    uint8_t* head_label_out;
    int head_label_classes;
    /* optional, with a fused head (round 6, ABI 302; cp_conv2d_fwd_split* with a head-only layer): whole output RECORDS.  head_out then addresses the
     * record of a pixel (stride head_out_ld, 16-byte aligned); its floats [0, head_prefix_n) are copied from head_prefix[pixel * head_prefix_ld + j]
     * (dense rows another head wrote), the head's channel q goes to float head_prefix_n + q.  Two heads that each write a slice of the records
     * leave every 128-byte line partly written, which costs the memory system a read-modify-write; with this the last head writes whole lines.
     * 8 <= head_prefix_n <= 12.  NULL: head_out addresses the head's first column, as before. */
    const float* head_prefix;
    int head_prefix_n, head_prefix_ld;
} cp_conv_desc;

enum { CP_TILE_AUTO = 0, CP_TILE_128x128 = 1, CP_TILE_64x128 = 2, CP_TILE_128x64 = 3, CP_TILE_128x32 = 4,
       CP_TILE_64x64 = 5, CP_TILE_256x32 = 6, CP_TILE_HALO = 7 /* halo-tile kernel (needs weights_halo) */,
       CP_TILE_STEM = 8 /* 7x7/s2 4->64 stem kernel (weights_halo = cp_conv_pack_weights_stem_host packing) */ };

/* K extent (multiple of 32) of the packed weight rows for a conv with the given sources */
int cp_conv_ktot(int kh, int kw, int num_sources, const int* channels);
/* HOST helper: pack a Keras-layout kernel into Wp.  `layout` 0 = HWIO [kh][kw][cin][cout]
 * (layers.Conv2D), 1 = IHWO [cin][kh][kw][cout] (PartialConvolution.conv_w,
 * _normalization_layers.py:314-319).  cin = sum of real source channels; a source with
 * channels==4 in the descriptor may carry `real_channels[s]` < 4 real input channels.
 * dst has cout*ktot floats.  All pointers are host memory. */
int cp_conv_pack_weights_host(const float* w_host, int layout, int kh, int kw, int cout, int num_sources,
                              const int* channels, const int* real_channels, float* dst_host);
/* halo-kernel weight layout [chunk][tap][32 or 64][kc]: float count and HOST packing (3x3 kernels only) */
int cp_conv_halo_weight_floats(int cout, int num_sources, const int* channels);
int cp_conv_pack_weights_halo_host(const float* w_host, int layout, int cout, int num_sources, const int* channels,
                                   const int* real_channels, float* dst_host);
/* packing for the stem kernel (csrc/conv_stem.hip: 7x7 / stride 2 / pad 3, one 4-channel source, cout = 64), passed in
 * cp_conv_desc.weights_halo: 12800 floats [25 two-tap steps][2][64][4]; layout as for cp_conv_pack_weights_host */
int cp_conv_pack_weights_stem_host(const float* w_host, int layout, int real_channels, float* dst);
/* HOST: pack a [1][1][32][head_cout] (HWIO) 1x1 kernel for the fused head: 1024 floats */
int cp_conv_pack_head_weights_host(const float* w_host, int head_cout, float* dst_host);
int cp_conv2d_fwd_f32(const cp_conv_desc* desc, void* stream);
/* which CP_TILE_* instantiation cp_conv2d_fwd_f32 will launch for this descriptor (profiling aid) */
int cp_conv_selected_tile(const cp_conv_desc* desc);

/* ------------------------------------------------------------------------------------
 * The same convolution on the bf16 matrix pipe (csrc/conv_hsplit.hip), for the shallow high-resolution layers: 3x3 / stride 1 /
 * pad 1, cout <= 512 (a multiple of 4; channels beyond 64 in passes of 64), sources with 16-multiple channels plus an optional trailing 4-channel source (the image); source 0 may be
 * CP_SRC_BILINEAR_X2 or (with tap_label) CP_SRC_NEAREST_SEL, the fused upsamplings of the decoders; a fused 1x1 head as in cp_conv2d_fwd_f32.
 * Replaces the same reference call sites as cp_conv2d_fwd_f32 for those layers (models/casapose.py:61-82, resnet.py:97-103 stage 1).
 *   planes = 3: fp32-EQUIVALENT -- every fp32 operand is split exactly into three bf16 terms and six bf16 products are accumulated in fp32
 *               (error <= that of the fp32 MFMA); the default of the training plan, opt-in for inference
 *   planes = 1: operands rounded to bf16 (nearest even), fp32 accumulation -- "bf16 convolutions" (BASELINE.json configs[2]); 3e-2 gates
 * The descriptor is the one of cp_conv2d_fwd_f32 (tensors stay fp32 in HBM; desc->weights / weights_halo are ignored); epilogue:
 * row_scale (partial-conv 9/count, computed from tap_label), residual, out_raw, affine / CLADE table + activation -> out_act.
 * Weights: cp_conv_pack_weights_split_host lays a Keras kernel out as the fp32 image of the kernel's fragment stream
 * (cp_conv_split_weight_floats floats); cp_conv_split_weights_f32 turns that image into the bf16 planes
 * (cp_conv_split_weight_bytes bytes) on the device -- one gather + one launch re-packs after an optimizer step.
 * ---------------------------------------------------------------------------------- */
enum { CP_PLANES_F16X2 = 0x12 };   /* `planes` code of the fp16 two-way split (two planes; see cp_wino_gemm_split_scaled_f32) */
float cp_f16x2_weight_scale(float max_abs);   /* the power of two that brings max |w| into [2^11, 2^12) */
/* ---- the f16x2 RANGE GUARD behind the C ABI (round 6; the load path it protects: test_casapose.py:225-228 followed by model(img, training=False)) ----
 * The fp16 two-way split reproduces an fp32 operand to one ulp only while max |operand| of the converted tensor stays inside a band
 * ([0.5, 65504 / 4]).  WEIGHTS are brought there by cp_f16x2_weight_scale; ACTIVATIONS are measured on the device:
 *   - a MONITOR SLOT is four 32-bit words of device memory, 16-byte aligned, zeroed by the caller:
 *       [0] bits of max |x| over every fp32 value the armed launches converted to an fp16 pair (atomic max of the bit pattern), [1] number of
 *       launches that reported, [2] the same maximum for the operand of a fused 1x1 head (exists in registers only), [3] reserved;
 *   - cp_f16x2_monitor_set(slot) arms `slot` for the CALLING THREAD's following launches (NULL disarms).  Reporting launches:
 *       cp_conv2d_fwd_split_scaled / cp_conv2d_fwd_stem_split_scaled with CP_PLANES_F16X2 (what their loaders convert: the sources through the
 *       stem's input affine; the low-resolution source of a bilinear x2 input, which bounds its interpolation; word [2]: the fused head's operand),
 *       cp_wino_gemm_split_scaled_f32 with CP_PLANES_F16X2 (every row of V: arm it for a plain 1x1 GEMM only -- the padding rows of Winograd
 *       planes hold stale data), cp_wino_input_transform_f32 / _pre_f32 and cp_wino_output_input_transform_f32 (the planes V they WRITE: arm
 *       these, not the GEMM, for a Winograd layer).  A slot is sticky: maxima accumulate over launches and forwards until the caller zeroes it.
 *       An un-armed launch pays one uniform branch per staged slice;
 *   - cp_amax_f32 folds max |x| of a caller's own strided tensor into slot[0] (same atomic), for operands no kernel above converts;
 *   - cp_f16x2_range_check(amax, lo, hi, &rescale): 0 = inside [lo, hi] (or amax == 0), 1 = multiply the operand by the power of two *rescale
 *       (amax * rescale in [2^10, 2^11)) and the consumer's accumulator factor by its inverse -- exact; for V through
 *       cp_wino_input_transform_pre_f32's per-channel scale, for a fused head through the normalisation table feeding its (positively homogeneous)
 *       activation and head_descale --, 2 = no power of two in [2^-24, 2^24] helps or amax is not finite: run the layer with planes = 3.
 * Protocol of the host layer (casapose_amd/engine.py, ForwardPlan): a plan's first forward runs armed, reads the slots back (one synchronisation),
 * applies the remedies and repeats until every layer is in the band; afterwards every CASAPOSE_F16X2_MONITOR_EVERY-th forward runs armed, the
 * slots are copied to pinned host memory asynchronously and judged at the start of a later forward -- no synchronisation on the hot path; a slot
 * outside the band re-arms the calibration and raises one warning. */
int cp_f16x2_monitor_set(uint32_t* slot);
uint32_t* cp_f16x2_monitor_get(void);
int cp_f16x2_range_check(float amax, float lo, float hi, float* rescale);
int cp_amax_f32(const float* x, long long groups, long long group_stride, long long count, uint32_t* slot, void* stream);
int cp_conv_split_applicable(const cp_conv_desc* desc);   /* 1 if cp_conv2d_fwd_split covers this descriptor */
int cp_conv_split_weight_floats(int cout, int num_sources, const int* channels);
size_t cp_conv_split_weight_bytes(int cout, int num_sources, const int* channels, int planes);
int cp_conv_pack_weights_split_host(const float* w_host, int layout, int cout, int num_sources, const int* channels,
                                    const int* real_channels, float* dst_host);
int cp_conv_split_weights_f32(const float* packed, long long floats, int planes, void* out, void* stream);
/* HOST: a [1][1][32][head_cout] 1x1 kernel as the fp32 image (1024 floats) of the fused head's fragments; cp_conv_split_weights_f32
 * makes its planes.  head_weights_split may be NULL when desc->head_out is NULL. */
int cp_conv_pack_head_split_host(const float* w_host, int head_cout, float* dst_host);
int cp_conv2d_fwd_split(const cp_conv_desc* desc, const void* weights_split, const void* head_weights_split, int planes, void* stream);
/* planes = CP_PLANES_F16X2 (see cp_wino_gemm_split_scaled_f32 below for the arithmetic): the weight image is multiplied by `scale` (a power of two,
 * cp_f16x2_weight_scale(max |w|)) before its fp16 split, the convolution multiplies its accumulators by w_descale = 1 / scale (the fused head's by
 * head_descale, the inverse of the head image's scale).  With planes = 1 / 3 the factors must be 1 and the calls equal the two above. */
int cp_conv_split_weights_scaled_f32(const float* packed, long long floats, int planes, float scale, void* out, void* stream);
int cp_conv2d_fwd_split_scaled(const cp_conv_desc* desc, const void* weights_split, const void* head_weights_split, int planes, float w_descale,
                               float head_descale, void* stream);
/* Direct convolution with bf16 OPERANDS for the deep 3x3 layers (round 3; BASELINE.json configs[2] "bf16 convs"; csrc/conv_bf16d.hip): 3x3 / stride 1,
 * pad = dilation in {1, 2, 4}, one or two direct sources of 16-multiple channels, cout a multiple of 128 (<= 512), fp32 tensors in HBM, operands
 * rounded to bf16 (nearest even) while staged, fp32 accumulation on v_mfma_f32_32x32x16_bf16.  Epilogue: + residual, out_raw and / or
 * out_act = act(raw * scale[c] + shift[c]) with per-channel tables (no labels / tap masks / heads).  weights_bf16 = the ONE-plane
 * fragment stream of cp_conv2d_fwd_split (cp_conv_pack_weights_split_host + cp_conv_split_weights_f32(planes = 1)).  NOT fp32-equivalent:
 * gates 3e-2 of the output range against the fp32 convolution, 2e-5 against the convolution of the bf16-rounded operands. */
int cp_conv_bf16_deep_applicable(const cp_conv_desc* desc);
int cp_conv2d_fwd_bf16_deep(const cp_conv_desc* desc, const void* weights_bf16, void* stream);

/* ------------------------------------------------------------------------------------
 * Small streaming kernels around the convolutions
 * ---------------------------------------------------------------------------------- */

/* [n,h,w,3] -> [n,h,w,4] (4th channel 0): operand layout for the C4 source mode.
 * Replaces nothing arithmetic; feeds conv0 (resnet.py:247-249) and blocks 5/10
 * (pose_models.py:545,607). */
int cp_pad_channels_3to4(const float* src, float* dst, long long pixels, void* stream);

/* ZeroPadding2D(1) + MaxPooling2D(3x3, stride 2) (resnet.py:253-254) with an optional fused
 * per-channel affine + ReLU of the consumer's BatchNorm (resnet.py:78-79).  channels % 4 == 0. */
int cp_maxpool3x3s2_f32(const float* src, int batch, int h, int w, int channels, const float* scale,
                        const float* shift, int relu, float* dst, void* stream);

/* UpSampling2D(size 2, bilinear) == tf.image.resize half-pixel centres (casapose.py:135-140).
 * Stand-alone form; channels % 4 == 0. */
int cp_upsample_bilinear_x2_f32(const float* src, int batch, int h, int w, int channels, float* dst,
                                void* stream);

/* GuidedUpsampling gather (_normalization_layers.py:554-565), stand-alone form, given the
 * neighbour-selection map from cp_label_pyramid.  src [n,h,w,c] -> dst [n,2h,2w,c]. */
int cp_guided_upsample_x2_f32(const float* src, const uint8_t* sel, int batch, int h, int w, int channels,
                              float* dst, void* stream);

/* Front end of ransac_voting_layer_all_masks (ransac_voting.py:276-301): object masks [b,h,w,objects] (float, > 0.5 = inside, no background
 * channel) -> uint8 label map (highest such object index + 1; 0 = none) and counts[b][objects] = pixels inside each mask. */
int cp_mask_to_labels_f32(const float* mask, int batch, int h, int w, int objects, uint8_t* labels, int32_t* counts, void* stream);

/* arg-max over `classes` contiguous values per pixel (pixel stride ld) -> uint8 label.
 * Replaces softmax(1e6*x) as a hard one-hot (pose_models.py:547-554; voting_layers_2d.py:38-41).
 * First maximum wins (ties are undefined behaviour in the reference). */
int cp_argmax_labels(const float* logits, int ld, int classes, long long pixels, uint8_t* labels,
                     void* stream);

/* Everything decoder 2 derives from the label map (pose_models.py:556-559;
 * _normalization_layers.py:294-299,333-352,512-551):
 *   levels 0..3 = full, 1/2, 1/4, 1/8 resolution (HalfSize == [::2,::2]);
 *   labels[l]    uint8 [n,h_l,w_l]                 (labels[0] is the INPUT)
 *   pnorm[l]     float [n,h_l,w_l]  9 / #{3x3 taps in bounds with the centre's label}
 *   sel[l]       uint8 [n,h_l,w_l]  for l = 0..2: which neighbour of level l+1 the guided
 *                                   upsampling picks for each pixel of level l (0..3)
 * Any output pointer may be NULL.  h, w are the level-0 size; h_l = h_{l-1}/2 (floor). */
int cp_label_pyramid(const uint8_t* labels0, int batch, int h, int w, uint8_t* const* labels /*[4]*/,
                     float* const* pnorm /*[4]*/, uint8_t* const* sel /*[3]*/, void* stream);

/* ------------------------------------------------------------------------------------
 * Keypoint voting
 * ---------------------------------------------------------------------------------- */

/* CoordLSVotingWeighted.calc (casapose/pose_estimation/voting_layers_2d.py:83-122):
 * per (image, object, keypoint) fp64 sums of w(I-nn^T) and w(I-nn^T)c over the object's
 * pixels, then the 2x2 pseudo-inverse solve.  `field` is the network output
 * [n,h,w,ld] with seg logits at channel seg_off (classes = objects+1 values), directions
 * (dy,dx)*kp at dir_off and confidence logits at conf_off.  If `labels` is non-NULL it is
 * used as the hard object map instead of the arg-max of the logits (this is how the
 * connected-component filter of :43-79 is applied).  sums_ws: fp64 [n][objects][kp][5]
 * workspace (zeroed by the call).  keypoints: fp32 [n][objects][kp][2] in (y,x) pixels. */
int cp_ls_vote_f32(const float* field, int ld, int seg_off, int dir_off, int conf_off, const uint8_t* labels,
                   int batch, int h, int w, int objects, int kp, double* sums_ws, float* keypoints,
                   void* stream);
/* the same with the pixel weight selectable: sigmoid_weights = 0 -> softplus(conf) (cp_ls_vote_f32), 1 -> sigmoid(conf)
 * (CoordLSVotingWeighted(sigmoid_weights=True), voting_layers_2d.py:32-33; sigmoid_scale is 1 in the reference) */
int cp_ls_vote_w_f32(const float* field, int ld, int seg_off, int dir_off, int conf_off, const uint8_t* labels,
                     int batch, int h, int w, int objects, int kp, int sigmoid_weights, double* sums_ws, float* keypoints,
                     void* stream);
size_t cp_ls_vote_workspace_bytes(int batch, int objects, int kp);

/* Largest-connected-component filter of voting_layers_2d.py:43-79 (tfa.image.connected_components,
 * 4-connectivity, keep the largest component of each object if it has >= min_size pixels):
 * labels_in uint8 [n,h,w] -> labels_out (pixels outside the kept component become 0).
 * rank 1 = the reference's default: the histogram entry that ranks SECOND in top_k order (the first is assumed to be bin 0, "everything
 * else"); rank 2 = output_second_largest_component (:58-59,71-73: three bins, the THIRD entry).  ws: int32 workspace of cp_ccl_workspace_bytes. */
int cp_ccl_filter_labels(const uint8_t* labels_in, int batch, int h, int w, int objects, int min_size, int rank, void* ws,
                         uint8_t* labels_out, void* stream);
size_t cp_ccl_workspace_bytes(int batch, int h, int w, int objects);

/* ransac_voting_layer_all_masks (casapose/pose_estimation/ransac_voting.py:276-368,447-484).
 * One call votes all (image, object) pairs.  labels: uint8 [n,h,w] hard object map
 * (arg-max one-hot of pose_evaluation.py:37-38); vertex: [n,h,w,ld] with (dy,dx)*kp at
 * dir_off.  idx: int32 [max_iter][n][objects][hyp][kp][2] uniform random draws in
 * [0, 2^31) supplied by the caller (the reference draws tf.random.uniform per round,
 * :319-321; tests inject them); a draw d selects the (d % tn)-th pixel of the object in
 * raster order (the order of tf.where, :303-305).  Objects with fewer than min_num pixels
 * give zeros (:290-292).  The random sub-sampling of objects above max_num pixels
 * (:295-301) is the caller's job (apply it to `labels`); max_num is accepted for signature
 * parity only.  Rounds stop per object when 1-(1-r_min^2)^hyps > confidence or after
 * max_iter rounds (:340-347), decided on the device; after rounds 1, 2, 4, 8, 16 the call synchronises `stream` once to learn whether any object
 * is still voting and stops launching rounds when none is.  hyp must be a multiple of 16.
 * out: fp32 [n][objects][kp][2] in (x,y); rounds_out (optional) int32 [n][objects].
 * ws: workspace of cp_ransac_workspace_bytes. */
int cp_ransac_vote_f32(const uint8_t* labels, const float* vertex, int ld, int dir_off, int batch, int h,
                       int w, int objects, int kp, const int32_t* idx, int hyp, float inlier_thresh,
                       float confidence, int max_iter, int min_num, int max_num, void* ws, float* out,
                       int32_t* rounds_out, void* stream);
/* The same voter with the random numbers made INSIDE the library from a 64-bit seed (counter-based: equal seeds give equal results, no state):
 * the pixel-pair draws of every round, and the random thinning of objects above max_num pixels (:295-301: a pixel is kept with probability
 * max_num / count) as part of the compaction -- no draw tensor (94 MB per call at the reference's settings), no pass over the mask outside. */
int cp_ransac_vote_seeded_f32(const uint8_t* labels, const float* vertex, int ld, int dir_off, int batch, int h,
                              int w, int objects, int kp, unsigned long long seed, int hyp, float inlier_thresh,
                              float confidence, int max_iter, int min_num, int max_num, void* ws, float* out,
                              int32_t* rounds_out, void* stream);
size_t cp_ransac_workspace_bytes(int batch, int h, int w, int objects, int kp, int hyp);

/* GuidedBilinearUpsampling (_normalization_layers.py:569-664; casapose_c_gcu4_bilat): mask[n,Y,X] bit j = the low-resolution label of
 * tap j in {(y,x),(y,x+1),(y+1,x),(y+1,x+1)} (zero padded bottom/right) equals labels_hi[n,Y,X]; the upsampling blends the four taps
 * with the fixed sub-pixel weights after replacing non-matching taps by the mean of the matching ones (0 if none), which is the
 * 4-tap gather out = sum_j coef_j(mask, sub-pixel) * tap_j.  h, w = LOW-resolution size for the two upsampling calls. */
int cp_guided_match_mask(const uint8_t* labels_hi, const uint8_t* labels_lo, int batch, int h_hi, int w_hi, uint8_t* mask, void* stream);
int cp_guided_bilinear_upsample_x2_f32(const float* src, const uint8_t* mask, int batch, int h, int w, int channels, float* dst, void* stream);
int cp_guided_bilinear_upsample_x2_bwd_f32(const float* dy, int ld_dy, const uint8_t* mask, int batch, int h, int w, int channels, float* dx,
                                           void* stream);

/* ------------------------------------------------------------------------------------
 * Winograd F(4x4,3x3) path for deep 3x3 / stride-1 / pad == dilation convolutions (same layers.Conv2D call sites as
 * cp_conv2d_fwd_f32; the MFMA work drops 4x).  Three launches:
 *   cp_wino_input_transform_f32 : V[p][t][c_off + c] = (B^T d B)[p] for every 6x6 patch of `src` (once per source of a
 *                                 concatenated input); V is [36][tiles_padded][ldv]
 *   cp_conv2d_fwd_f32           : 1x1 over 36*tiles_padded "pixels" with group_rows = tiles_padded,
 *                                 group_weight_stride = cout*ldk and the weights of cp_wino_pack_weights_host -> M
 *   cp_wino_output_transform_f32: Y = A^T M A + the epilogue of cp_conv2d_fwd_f32 (residual, affine / per-label table,
 *                                 activation, raw and activated stores)
 * Dilation d is exact by sub-grid decomposition (d*d independent dilation-1 problems).  cp_wino_tiles returns the tile
 * count and its padding (multiple of 128).  cp_wino_pack_weights_host writes dst[p][co][k_off + c] = (G g G^T)[p] for the
 * input channels [c_begin, c_begin+real_channels) of a HWIO kernel; dst is [36][cout][ldk] and must be zeroed first.
 * ---------------------------------------------------------------------------------- */
int cp_wino_tiles(int batch, int h, int w, int dilation, int* tiles, int* tiles_padded);
/* The grouped GEMM as a dedicated persistent kernel: M[r][n] = sum_k V[r][k] * U[r / group_rows][n][k], rows = 36*tiles_padded,
 * k % 32 == 0.  Same arithmetic as the grouped mode of cp_conv2d_fwd_f32 (which remains available); the operand stream is
 * pipelined ACROSS tiles because a tile only lives for k/32 chunks. */
int cp_wino_gemm_f32(const float* V, const float* U, float* M, int rows, int group_rows, int k, int n, void* stream);
/* OPT-IN alternative with fp32-equivalent results on the bf16 matrix pipe: every operand is split exactly into three bf16 terms and six
 * of the nine products (all but the three below 2^-24) are accumulated in fp32 (csrc/wino_gemm_split.hip).  V is the same
 * fp32 tensor (split while it is staged); the weights are pre-split ONCE into fragment-major bf16 planes by cp_wino_split_weights_f32
 * (U as for cp_wino_gemm_f32: [groups][n][k] fp32; `out` of cp_wino_split_weights_bytes(groups, n, k) bytes, caller-owned).
 * group_rows must be a multiple of 128.  Selected by CASAPOSE_WINO_GEMM=split; never the default of bench.py. */
size_t cp_wino_split_weights_bytes(int groups, int n, int k);
int cp_wino_split_weights_f32(const float* U, int groups, int n, int k, void* out, void* stream);
int cp_wino_gemm_split_f32(const float* V, const void* Usplit, float* M, int rows, int group_rows, int k, int n, void* stream);
/* planes = 3: the above.  planes = 2: hi + mid planes only (16 significand bits per operand; products hi*hi, hi*mid, mid*hi): half the MFMAs, NOT
 * fp32-equivalent -- for the bf16 conv modes (BASELINE.json configs[2]; gates 3e-2).  Same pre-split weights. */
int cp_wino_gemm_split_planes_f32(const float* V, const void* Usplit, float* M, int rows, int group_rows, int k, int n, int planes, void* stream);
/* planes = CP_PLANES_F16X2: the fp16 TWO-way split (csrc/split_f16.h): hi = rn_f16(x), lo = rn_f16(x - hi) reproduce an fp32 operand to within
 * one fp32 ulp (2^-23 worst case, 0.75 * 2^-24 rms, three operands in four exactly) and the three products hi*hi, hi*lo, lo*hi are exact in the fp32 accumulator of v_mfma_f32_32x32x16_f16 -- fp32-level
 * accuracy (measured against fp64 beside the fp32 MFMA and the exact bf16 split: tests/test_gpu_f16x2.py) with HALF the MFMAs of
 * the exact bf16 split.  fp16's range is handled by scaling: the weights are multiplied by a power of two before their split
 * (cp_f16x2_weight_scale(max |w|): max -> [2^11, 2^12), so that the low parts are normal numbers) and the kernel multiplies its accumulators by
 * c_scale = 1 / scale (exact); activations are used as they are: fp32-level for max |v| in [1, 65504]; below, the low parts
 * become subnormal (absolute 2^-25: the error degrades as 2^-25 / max |v|); above, conversions clamp at +-65504 and the error grows to 2^-12 of the
 * operand -- never inf / NaN.  The split-weight buffer has the size and layout of the three-plane one (planes 0 / 1 used). */
int cp_wino_split_weights_scaled_f32(const float* U, int groups, int n, int k, int planes, float scale, void* out, void* stream);
int cp_wino_gemm_split_scaled_f32(const float* V, const void* Usplit, float* M, int rows, int group_rows, int k, int n, int planes, float c_scale,
                                  void* stream);
int cp_wino_pack_weights_host(const float* w_hwio, int cin_total, int cout, int c_begin, int channels, int real_channels, int ldk,
                              int k_off, float* dst);
/* device version of the weight transform (training: after every optimizer step): g(ky,kx,c,o) is read at
 * w[ky'*stride_ky + kx'*stride_kx + c*stride_in + o*stride_out] with (ky',kx') = flip ? (2-ky,2-kx) : (ky,kx), so the same kernel
 * serves HWIO / IHWO masters and the flipped, transposed data-gradient kernel.  U[p][o][k_off + c], rows of ldk floats. */
int cp_wino_transform_weights_f32(const float* w, long long stride_ky, long long stride_kx, long long stride_in, long long stride_out,
                                  int flip, int channels, int cout, int ldk, int k_off, float* U, void* stream);
/* weight gradient through the Winograd planes (training): dM[p][t][c] = (A dY A^T)[p] for every 4x4 tile of dy; then
 * dU[p][o][k] = sum_t dM[p][t][o] * V[p][t][k] is cp_conv2d_wgrad_f32 in its grouped mode (group_rows = tiles_padded) and
 * cp_wino_weight_grad_f32 folds the 36 planes back: dw(ky,kx,c,o) = sum_{a,b} G[a][ky] G[b][kx] dU[a*6+b][o][k_off+c], written at
 * dw[ky*stride_ky + kx*stride_kx + c*stride_in + o*stride_out] (the master layout). */
int cp_wino_dy_transform_f32(const float* dy, int ld, int channels, int batch, int h, int w, int dilation, float* dM, void* stream);
int cp_wino_weight_grad_f32(const float* dU, int channels, int cout, int ldk, int k_off, long long stride_ky, long long stride_kx,
                            long long stride_in, long long stride_out, float* dw, int accumulate, void* stream);
/* The grouped weight-gradient GEMM of the Winograd layers on the bf16 matrix pipe (csrc/wino_wgrad_split.hip):
 *   du[g][n][k] = sum_t dm[g][t][n] * v[g][t][k]      (g < groups = 36 planes, t < rows = padded tiles, n = cout, k = cin)
 * what cp_conv2d_wgrad_f32 computes in its grouped mode with fp32 MFMAs, here as exact three-way bf16 splits (planes = 3, fp32-equivalent:
 * six exact products per fp32 product, fp32 accumulation) or with operands rounded to bf16 (planes = 1).  dm = the output of
 * cp_wino_dy_transform_f32, v = the forward's transformed input, du feeds cp_wino_weight_grad_f32.  n and k multiples of 128
 * (cp_wino_wgrad_split_applicable); du is overwritten. */
int cp_wino_wgrad_split_applicable(int groups, int rows, int n, int k);
int cp_wino_wgrad_split_f32(const float* dm, const float* v, float* du, int groups, int rows, int n, int k, int planes, void* stream);
/* The same GEMM with planes = CP_PLANES_F16X2 (round 6): both operands as fp16 pairs, three exact products per fp32 product.  dm is multiplied
 * by dm_scale and v by v_scale before the split (powers of two the caller picks so that each operand's maximum sits in [0.5, 65504 / 4]:
 * cp_wino_dy_transform_f32 and the input transforms report those maxima into the armed monitor slot), du takes 1 / (dm_scale v_scale).
 * planes 1 / 3 with factors of 1 are cp_wino_wgrad_split_f32. */
int cp_wino_wgrad_split_scaled_f32(const float* dm, const float* v, float* du, int groups, int rows, int n, int k, int planes, float dm_scale,
                                   float v_scale, void* stream);
int cp_wino_input_transform_f32(const float* src, int ld, int channels, int batch, int h, int w, int dilation, float* V, int ldv,
                                int c_off, void* stream);
int cp_wino_output_transform_f32(const float* M, int cout, int batch, int h, int w, int dilation, const float* residual, int residual_ld,
                                 const float* scale, const float* shift, const uint8_t* epi_label, int act, float* out_raw,
                                 int out_raw_ld, float* out_act, int out_act_ld, void* stream);
/* The ResNet stem (conv0: 7x7 / stride 2 / pad 3, one 4-channel source, cout 64 -- the range of the CP_TILE_STEM kernel) on the bf16 matrix
 * pipe (csrc/conv_stem_split.hip, round 4): planes = 3 exact three-way bf16 splits (fp32-equivalent), planes = 1 bf16 operands.  Same descriptor
 * as cp_conv2d_fwd_f32 (input affine on real pixels, per-channel affine + activation epilogue, raw / activated outputs).  Weights:
 * cp_conv_pack_weights_stem_split_host (HOST: HWIO / IHWO kernel -> fp32 image of the fragment stream, cp_conv_stem_split_weight_floats()
 * floats), then cp_conv_split_weights_f32(image, floats, planes, out) on the device (floats / 512 * planes * 1024 bytes). */
int cp_conv_stem_split_weight_floats(void);
int cp_conv_pack_weights_stem_split_host(const float* w_host, int layout, int real_channels, float* dst_host);
int cp_conv2d_fwd_stem_split(const cp_conv_desc* d, const void* weights_split, int planes, void* stream);
/* planes = CP_PLANES_F16X2: as cp_conv2d_fwd_split_scaled (weights from cp_conv_split_weights_scaled_f32 with the same power-of-two scale) */
int cp_conv2d_fwd_stem_split_scaled(const cp_conv_desc* desc, const void* weights_split, int planes, float w_descale, void* stream);
/* Output transform of one Winograd layer FUSED with the input transform of the next (round 4): Y = A^T M A, + residual, raw store (optional),
 * t = act(Y * scale[c] + shift[c]) (per channel; optional activated store), then V[p][t][c_off + c] = (B^T t B)[p] of the consumer -- for two
 * Winograd convolutions of the same (batch, h, w, dilation) where the consumer's only source is this activated output (the residual-unit chains
 * of resnet.py:57-113 in inference).  The activated map never goes to HBM unless out_act is given.  One block owns a whole sub-grid (the pixels
 * with equal (y mod d, x mod d)) x 32 channels (60x80 at dilation 4: 4 x 5 tiles) or x 16 channels (the 30x40 sub-grids of dilation 2);
 * cp_wino_output_input_applicable says whether a geometry fits one of the two block shapes, otherwise the two separate transforms are used. */
int cp_wino_output_input_applicable(int batch, int h, int w, int dilation, int cout);
int cp_wino_output_input_transform_f32(const float* M, int cout, int batch, int h, int w, int dilation, const float* residual, int residual_ld,
                                       const float* scale, const float* shift, int act, float* out_raw, int out_raw_ld, float* out_act,
                                       int out_act_ld, float* V, int ldv, int c_off, void* stream);
/* Training-step forms of the two transforms (round 3, fused normalisation: casapose.py:76-105 / resnet.py:78-103 are convolution ->
 * normalisation -> activation -> convolution).  _stats: also accumulates stats[c] = sum, stats[cout + c] = sum of squares (fp64; zeroed by
 * the call) of the RAW output over the real pixels = the batch statistics cp_bn_stats_f32 would compute from out_raw.  _pre: applies
 * y = act(fma(x, pre_scale[c], pre_shift[c])) (cp_affine_act_f32's expression, per channel, both NULL = identity) to every real pixel as it
 * is loaded, so the activated tensor between a normalisation layer and a Winograd layer need not be stored.  pre_scale alone (pre_shift
 * NULL, pre_act CP_ACT_NONE; round 6) = a per-channel factor only: what brings a gradient into fp16's band in front of an f16x2 GEMM. */
int cp_wino_output_transform_stats_f32(const float* M, int cout, int batch, int h, int w, int dilation, const float* residual, int residual_ld,
                                       const float* scale, const float* shift, const uint8_t* epi_label, int act, float* out_raw, int out_raw_ld,
                                       float* out_act, int out_act_ld, double* stats, void* stream);
int cp_wino_input_transform_pre_f32(const float* src, int ld, int channels, int batch, int h, int w, int dilation, float* V, int ldv, int c_off,
                                    const float* pre_scale, const float* pre_shift, int pre_act, void* stream);

/* ====================================================================================
 * TRAINING PATH (train_casapose.py:494-611: forward with training=True, compute_loss,
 * tf.GradientTape gradients, Adam).  The reference gets every gradient below from
 * TensorFlow's autodiff of the layers cited at the forward entry points; the kernels here
 * are the adjoints of those layers, written out.
 * ==================================================================================== */

/* Weight gradient of cp_conv2d_fwd_f32 (Conv2DBackpropFilter of resnet.py:85-103,249; casapose.py:71-74;
 * PartialConvolution, _normalization_layers.py:325-373):
 *   dw_packed[co][k] (+)= sum_m dy[m][co] * A[m][k]
 * with A the forward's implicit im2col matrix (same descriptor: geometry, sources, tap_label; weights, outputs
 * and epilogue fields are ignored -- for a partial convolution dy is the gradient of the un-normalised sum,
 * i.e. already multiplied by row_scale, see cp_bn_act_bwd_apply_f32) and k in the packed K order, so dw_packed has the layout
 * of cp_conv_pack_weights_host.  Sources must be CP_SRC_DIRECT without pre-affine.  Split over pixels
 * with fp32 atomics: the summation order is not fixed (results reproducible to fp32 rounding only). */
int cp_conv2d_wgrad_f32(const cp_conv_desc* desc, const float* dy, int dy_ld, float* dw_packed, int accumulate, void* stream);
/* The same product on the bf16 matrix pipe (csrc/conv_wgrad_split.hip), the backward partner of cp_conv2d_fwd_split: 3x3 / stride 1 / pad 1,
 * direct sources whose channel counts are multiples of 32 plus an optional trailing 4-channel source (the image: its columns are computed by
 * the fp32 kernel), cout a multiple of 32, tap_label as above.  planes = 3: both operands split exactly into three bf16 terms, six products per
 * fp32 product, fp32 accumulation (fp32-equivalent); planes = 1: operands rounded to bf16 (BASELINE.json configs[2]); planes = CP_PLANES_F16X2
 * (round 6): both operands as fp16 pairs, three exact products -- the caller keeps max |x| and max |dy| inside [0.5, 65504 / 4] (cp_amax_f32).  Same dw_packed layout,
 * same accumulate flag, same unfixed summation order as cp_conv2d_wgrad_f32. */
int cp_conv_wgrad_split_applicable(const cp_conv_desc* desc);
int cp_conv2d_wgrad_split(const cp_conv_desc* desc, const float* dy, int dy_ld, float* dw_packed, int accumulate, int planes, void* stream);

/* The two full-resolution 1x1 heads of the training step (pv_final_conv_segmentation / pv_final_conv_vertex, pose_models.py:546,616: 32 input
 * channels, cout <= 32) as streaming kernels (csrc/head1x1.hip; exact fp32 MFMA like cp_conv2d_fwd_f32 / cp_conv2d_wgrad_f32): w and dw are the
 * Keras kernel [32][cout] itself (no packing).  x rows: 32 channels, 16-byte aligned, ld_x % 4 == 0.  dy_row_floats = floats that may be READ in
 * each dy row starting at dy (>= cout; >= 32 enables 16-byte loads, columns >= cout are ignored).  accumulate: add to dx / dw. */
int cp_head1x1_fwd_f32(const float* x, int ld_x, long long pixels, const float* w, int cout, float* out, int ld_out, void* stream);
int cp_head1x1_dgrad_f32(const float* dy, int ld_dy, int dy_row_floats, long long pixels, const float* w, int cout, float* dx, int ld_dx,
                         int accumulate, void* stream);
int cp_head1x1_wgrad_f32(const float* x, int ld_x, const float* dy, int ld_dy, long long pixels, int cout, float* dw, int accumulate, void* stream);

/* Fused normalisation of decoder blocks 5 / 10 in the training step (round 3; casapose.py:76-105 is one Keras block -- convolution,
 * (class-adaptive) normalisation, activation -- and the heads follow it, pose_models.py:546,616).  The raw convolution output x [pixels][32]
 * stays the only stored tensor: the head's forward and weight gradient recompute y = act(fma(x, scale[l], shift[l])) (tables [classes][32],
 * labels = NULL for classes == 1, act = CP_ACT_*), and the two passes of the normalisation backward recompute the head's data gradient
 * g_y = dout W^T on the matrix pipe instead of reading a stored one: same red / chan / dx as cp_bn_act_bwd_reduce_f32 / _apply_f32 fed with
 * cp_head1x1_dgrad_f32's output.  pixels % 32 == 0; every dout row has >= 32 readable floats from `dout` on (columns >= cout ignored). */
int cp_head1x1_fwd_affine_f32(const float* x, int ld_x, long long pixels, const float* scale, const float* shift, const uint8_t* labels, int classes,
                              int act, const float* w, int cout, float* out, int ld_out, void* stream);
/* The same forward writing COMPLETE output records (round 6): out[p][0 .. prefix_n) = prefix[p][0 .. prefix_n) (dense rows another head
 * wrote: the segmentation logits), out[p][prefix_n + q] = this head's column q.  Two heads that each fill a slice of [pixels][K + V] records
 * leave every 128-byte line partly written, which costs a read-modify-write in the memory system (2.2-2.7x the time of the same bytes
 * written as whole lines); with this entry point the last head is the only writer of the records.  1 <= prefix_n <= 16,
 * ld_out >= prefix_n + cout (== for whole lines). */
int cp_head1x1_fwd_affine_record_f32(const float* x, int ld_x, long long pixels, const float* scale, const float* shift, const uint8_t* labels,
                                     int classes, int act, const float* w, int cout, const float* prefix, int ld_prefix, int prefix_n, float* out,
                                     int ld_out, void* stream);
int cp_head1x1_wgrad_affine_f32(const float* x, int ld_x, const float* scale, const float* shift, const uint8_t* labels, int classes, int act,
                                const float* dy, int ld_dy, long long pixels, int cout, float* dw, int accumulate, void* stream);
int cp_head1x1_bn_bwd_reduce_f32(const float* x, int ld_x, const float* dout, int ld_dout, int dout_row_floats, long long pixels, const float* w, int cout,
                                 const float* mean, const float* rstd, const float* gamma, const float* fwd_scale, const float* fwd_shift,
                                 const uint8_t* labels, int classes, int act, double* red, double* chan, void* stream);
int cp_head1x1_bn_bwd_apply_f32(const float* x, int ld_x, const float* dout, int ld_dout, int dout_row_floats, long long pixels, const float* w, int cout,
                                const float* mean, const float* rstd, const float* gamma, const float* fwd_scale, const float* fwd_shift,
                                const uint8_t* labels, int classes, int act, const double* chan, double global_pixels, const float* row_scale,
                                float* dx, int ld_dx, void* stream);

/* The DATA gradient needs no entry point of its own: it is cp_conv2d_fwd_f32 over dy with the kernel
 * flipped and transposed (host-side repack), pad' = dilation*(k-1) - pad, the same dilation, and, for a
 * stride-2 forward, source mode CP_SRC_ZERO_INSERT_X2.  A partial convolution keeps its tap_label (the mask
 * is symmetric) and takes row_scale-multiplied dy (see cp_bn_act_bwd_apply_f32). */

/* Batch statistics of (Sync)BatchNormalization in training mode (resnet.py:78,100,247,250,303;
 * casapose.py:77; _normalization_layers.py:108): sums[c] = sum_p x[p][c], sums[C+c] = sum_p x[p][c]^2 in
 * fp64 (zeroed by the call).  The caller all-reduces `sums` across replicas (SyncBN) and derives
 * mean / biased variance.  channels % 4 == 0, <= 1024. */
int cp_bn_stats_f32(const float* x, long long pixels, int channels, int ld, double* sums, void* stream);

/* Turn the (all-reduced) statistic sums into everything the normalisation needs, in one launch:
 * mean[c], rstd[c] = 1/sqrt(biased var + eps); gamma_full/beta_full [classes][channels] (gamma [classes][real_channels]
 * or null = 1, beta likewise = 0; padding channels get 1/0); scale = gamma*rstd, shift = beta - mean*scale (pad_one: padding
 * channels get scale 0, shift 1); moving = moving*momentum + batch*(1-momentum) when moving_mean/moving_var are given
 * (Keras BatchNormalization update with the biased batch variance).  pixels = GLOBAL sample count. */
int cp_bn_finalize_f32(const double* sums, double pixels, int channels, int real_channels, int classes, const float* gamma,
                       const float* beta, float eps, int pad_one, float momentum, float* moving_mean, float* moving_var, float* mean,
                       float* rstd, float* gamma_full, float* beta_full, float* scale, float* shift, void* stream);
/* d beta[l][c] = red[(l*C+c)*2], d gamma[l][c] = red[(l*C+c)*2+1] as fp32 over the real channels (either output may be null) */
int cp_bn_param_grads_f32(const double* red, int channels, int real_channels, int classes, float* dgamma, float* dbeta, void* stream);

/* y[p][c] = act(x[p][c]*scale[l][c] + shift[l][c]), l = labels ? labels[p] : 0 -- the normalise(+CLADE
 * modulation, _normalization_layers.py:119-139)+activation (casapose.py:98-107) step with the batch
 * statistics folded into scale/shift by the caller. */
int cp_affine_act_f32(const float* x, long long pixels, int channels, int ld_x, const float* scale, const float* shift,
                      const uint8_t* labels, int act, float* y, int ld_y, void* stream);

/* Backward of y = act(gamma[l][c]*xhat + beta[l][c]), xhat = (x-mean[c])*rstd[c] (batch statistics), pass 1:
 *   red[(l*C+c)*2 + {0,1}] = sum_p {g, g*xhat}           (g = dy*act')  -> d beta[l][c], d gamma[l][c]
 *   chan[c*2 + {0,1}]      = sum_p {g*gamma, g*gamma*xhat}                -> the two batch means of pass 2
 * (fp64, zeroed by the call; gamma/beta may be null = 1/0; classes = 1 without labels).  The caller
 * all-reduces `chan` across replicas for SyncBN.
 * fwd_scale / fwd_shift (optional, [classes][channels], both or neither): the folded tables the forward's cp_affine_act_f32 was called
 * with.  When given, act' is evaluated at the forward's own pre-activation fma(x, scale, shift) -- the branch of every ReLU / leaky pair is
 * then bit-for-bit the one the forward took, so the result is the exact gradient of the piecewise-linear function the forward evaluated
 * (recomputing gamma*xhat + beta rounds differently and flips elements that sit within an ulp of zero). */
int cp_bn_act_bwd_reduce_f32(const float* x, int ld_x, const float* dy, int ld_dy, long long pixels, int channels, int classes,
                             const float* mean, const float* rstd, const float* gamma, const float* beta, const uint8_t* labels,
                             int act, const float* fwd_scale, const float* fwd_shift, double* red, double* chan, void* stream);
/* pass 2: dx = rstd*(g*gamma - chan[c][0]/N - xhat*chan[c][1]/N) * (row_scale ? row_scale[p] : 1),
 * N = global_pixels (all replicas); accumulate != 0 adds to dx (a tensor with several consumers). */
int cp_bn_act_bwd_apply_f32(const float* x, int ld_x, const float* dy, int ld_dy, long long pixels, int channels, const float* mean,
                            const float* rstd, const float* gamma, const float* beta, const uint8_t* labels, int act,
                            const float* fwd_scale, const float* fwd_shift, const double* chan, double global_pixels, const float* row_scale,
                            float* dx, int ld_dx, int accumulate, void* stream);

/* Adjoints of MaxPooling2D(3, strides 2) after ZeroPadding2D(1) (resnet.py:252-253), UpSampling2D(bilinear)
 * (casapose.py:135-140) and the GuidedUpsampling gather (_normalization_layers.py:554-558).  Gather form
 * (every input pixel sums the output gradients that reference it): deterministic, no atomics.
 * h, w are the LOW-resolution (input) sizes for the two upsampling adjoints. */
int cp_maxpool3x3s2_bwd_f32(const float* x, const float* dy, int batch, int h, int w, int channels, float* dx, int accumulate,
                            void* stream);
/* The training step's pair (round 3): the forward also records the arg-max TAP (0..8, raster order, first maximum wins, zero padding takes
 * part) of every output element in idx [batch][ho][wo][channels] (one byte each, 4-byte aligned), and the adjoint routes dy by those bytes
 * without reading x: same dx as cp_maxpool3x3s2_bwd_f32. */
int cp_maxpool3x3s2_idx_f32(const float* src, int batch, int h, int w, int channels, float* dst, uint8_t* idx, void* stream);
int cp_maxpool3x3s2_bwd_idx_f32(const uint8_t* idx, const float* dy, int batch, int h, int w, int channels, float* dx, int accumulate, void* stream);
int cp_upsample_bilinear_x2_bwd_f32(const float* dy, int ld_dy, int batch, int h, int w, int channels, float* dx, void* stream);
int cp_guided_upsample_x2_bwd_f32(const float* dy, int ld_dy, const uint8_t* sel, int batch, int h, int w, int channels, float* dx,
                                  void* stream);

/* dst[i] = idx[i] >= 0 ? src[idx[i]] : 0 -- re-pack the master (Keras-layout) weights into a kernel layout after
 * an optimizer step; cp_scatter_f32 is the inverse (packed weight gradient -> master layout; idx injective on
 * its non-negative entries). */
int cp_gather_f32(const float* src, const int32_t* idx, long long n, float* dst, void* stream);
int cp_scatter_f32(const float* src, const int32_t* idx, long long n, float* dst, int accumulate, void* stream);
/* out = alpha*a + beta*b (b may be null) */
int cp_axpby_f32(const float* a, float alpha, const float* b, float beta, long long n, float* out, void* stream);

/* tf.keras.optimizers.Adam.apply_gradients (train_casapose.py:334-347,611; epsilon 1e-7):
 *   g = grads*grad_scale; m = b1*m+(1-b1)*g; v = b2*v+(1-b2)*g^2;
 *   params -= lr*sqrt(1-b2^step)/(1-b1^step) * m/(sqrt(v)+eps)          (step counts from 1) */
int cp_adam_step_f32(float* params, const float* grads, float* m, float* v, long long n, float lr, float beta1, float beta2,
                     float eps, int step, float grad_scale, void* stream);

/* compute_loss for the merged-output models (train_casapose.py:40-145), value AND gradient:
 *   loss_sums[0] mask   = mean softmax cross-entropy(labels_ce)                         (:59-60)
 *   loss_sums[1] vertex = smooth_l1_loss(dirs, unit vectors to the keypoints, fg)        (loss_functions.py:14-44)
 *   loss_sums[2] proxy  = proxy_voting_loss_v2(loss_per_object=False)                    (loss_functions.py:132-203)
 * out: network output [batch,h,w,ld] = [seg_dim logits | 2*kp directions (dy,dx) | ...]; labels_ce / labels_fg:
 * uint8 [batch,h,w] (target_seg / filtered_seg as class indices); keypoints_yx: fp32 [batch][objects][kp][2].
 * With filter_with_segmentation the foreground keeps only pixels whose arg-max prediction equals labels_fg
 * (:64-69); with filter_high_proxy_errors objects whose mean smooth-L1 proxy distance (proxy_voting_dist, loss_functions.py:47-129;
 * objects with < 20 pixels count as 0) is >= 5 are dropped from it as well (:71-93).  object_loss_values (optional, [batch][objects])
 * receives those per-object values.  dout [batch*h*w][dld] receives d(mask_w*mask + vertex_w*vertex + proxy_w*proxy)/d out with the
 * logit gradients in columns [0,seg_dim) and the direction gradients in [vert_off, vert_off+2kp); every other
 * column is zeroed.  ws: cp_pose_loss_workspace_bytes. */
size_t cp_pose_loss_workspace_bytes(int batch, int h, int w);
int cp_pose_loss_f32(const float* out, int ld, int seg_dim, int kp, const uint8_t* labels_ce, const uint8_t* labels_fg,
                     const float* keypoints_yx, int objects, int batch, int h, int w, int filter_with_segmentation,
                     int filter_high_proxy_errors, float mask_w, float vertex_w, float proxy_w, void* ws, float* dout, int dld,
                     int vert_off, double* loss_sums, float* object_loss_values, void* stream);

/* compute_loss for the `pvnet` model with SEPARATED vector fields (train_casapose.py:57,97-125): the output holds seg_dim logits followed by
 * objects*2*kp direction channels, object o in [seg_dim + o*2kp, seg_dim + (o+1)*2kp).  vertex = sum over objects of smooth_l1_loss(slice,
 * target slice, one-hot of the object), proxy = sum over objects of proxy_voting_loss_v2(slice, the object's keypoints, one-hot of the object):
 * a pixel contributes to the slice of its own object only, normalised by 2kp * (pixels of that object in the image) + 1e-3, mean over the
 * batch.  Everything else as cp_pose_loss_f32 (labels, filter_with_segmentation, the gradient row: logits in [0,seg_dim), the directions of
 * object o in [vert_off + o*2kp, ...)).  ws: cp_pose_loss_sep_workspace_bytes. */
size_t cp_pose_loss_sep_workspace_bytes(int batch, int h, int w, int seg_dim);
int cp_pose_loss_sep_f32(const float* out, int ld, int seg_dim, int kp, const uint8_t* labels_ce, const uint8_t* labels_fg,
                         const float* keypoints_yx, int objects, int batch, int h, int w, int filter_with_segmentation, float mask_w,
                         float vertex_w, float proxy_w, void* ws, float* dout, int dld, int vert_off, double* loss_sums, void* stream);

/* The reference's loss / target functions as stand-alone passes (values only; the training step differentiates the fused kernels above).
 * They back the importable casapose.utils.loss_functions / casapose.utils.image_utils modules.
 *
 * cp_vector_field_f32 -- compute_vertex_hcoords_batch_v3 (utils/image_utils.py:17-63) and, with separated != 0, the per-object concatenation of
 * get_all_vectorfields (:66-79): out[p][slot*2kp + 2j..] = (keypoint_j - pixel centre) of the pixel's object (label l in 1..objects, slot =
 * separated ? l-1 : 0), l2-normalised when normalize != 0 (tf.math.l2_normalize, epsilon 1e-12), zero elsewhere.  keypoints_yx:
 * [batch][objects][instances][kp][2]; with several instances the one whose first keypoint (the centre) is nearest to the pixel is used. */
int cp_vector_field_f32(const uint8_t* labels, const float* keypoints_yx, int batch, int h, int w, int objects, int instances, int kp,
                        int separated, int normalize, float* out, int ld, void* stream);
/* smooth_l1_loss (utils/loss_functions.py:14-44) up to its final normalisation: sums[2b] = sum over the image's pixels and channels of
 * smoothL1(|w (pred - target)|), sums[2b+1] = sum of w; w = weights[p*wld] (wmode 0), |1 - weights[p*wld]| (wmode 1, `invert_weights`) or 1
 * (wmode 2, `ignore_weights`).  elem (optional, [batch*pixels][channels]) receives the per-element values (`normalize=False, reduce=False`). */
int cp_smooth_l1_f32(const float* pred, int pld, const float* target, int tld, const float* weights, int wld, int wmode, int channels,
                     int batch, long long pixels_per_image, double* sums, float* elem, void* stream);
/* proxy_voting_dist / proxy_voting_loss_v2 (utils/loss_functions.py:47-203) up to their final normalisations: per pixel and keypoint the
 * perpendicular distance between the keypoint of the pixel's object (labels: 0 = no object -> object 0, as the arg-max of an all-zero one-hot
 * row; l -> object l-1; minimum over the object's instances) and the line through the pixel centre along the predicted direction, times the
 * pixel weight (as cp_smooth_l1_f32).  img_sums[2b], [2b+1] = sum of smoothL1(dist), sum of weights; obj_sums / obj_counts (optional,
 * [batch][objects]) = the same sum per object (unsorted_segment_sum) and the object's pixel count; dist_out / elem (optional, [pixels][kp]) =
 * the distances / their smooth-L1 values. */
int cp_proxy_voting_f32(const float* pred, int pld, int kp, const uint8_t* labels, const float* weights, int wld, int wmode,
                        const float* keypoints_yx, int objects, int instances, int batch, int h, int w, double* img_sums, double* obj_sums,
                        int32_t* obj_counts, float* dist_out, float* elem, void* stream);

/* Backward of cp_ls_vote_f32 (the reference differentiates CoordLSVotingWeighted with tf.GradientTape,
 * train_casapose.py:555-579): dkeypoints = d loss / d keypoints [batch][objects][kp][2] (y,x px); sums_ws = the workspace
 * the forward left behind; pu_ws: float [batch*objects*kp*4] scratch.  Writes (accumulate: adds) d loss / d directions
 * into dfield[pix][ddir_off + 2j..] and d loss / d confidence logits into dfield[pix][dconf_off + j] for the pixels of
 * each object (zero elsewhere).  Optionally adds conf_coef[b][j]*sigmoid(conf) on the pixels with reg_labels != 0 (the
 * confidence regulariser of keypoint_reprojection_loss, loss_functions.py:254-262).  `labels` is required. */
int cp_ls_vote_bwd_f32(const float* field, int ld, int dir_off, int conf_off, const uint8_t* labels, int batch, int h, int w,
                       int objects, int kp, const double* sums_ws, const float* dkeypoints, float* pu_ws,
                       const uint8_t* reg_labels, const float* conf_coef, float* dfield, int dld, int ddir_off, int dconf_off,
                       int accumulate, void* stream);
/* counts[0][b][c] / counts[1][b][c] = pixels of class c in labels / labels_est (may be null) of image b; conf_sums[b][j] =
 * sum over labels != 0 of softplus(conf_j) (fp64) -- objects_available and the confidence regulariser
 * (loss_functions.py:236-262).  Zeroed by the call. */
int cp_kp_stats_f32(const float* field, int ld, int conf_off, const uint8_t* labels, const uint8_t* labels_est, int batch, int h, int w,
                    int classes, int kp, int32_t* counts, double* conf_sums, void* stream);
/* keypoint_reprojection_loss without BPnP (loss_functions.py:207-344): coords_yx [batch*objects][kp][2] voted keypoints
 * (crop pixels), affine [batch][6] row-major 2x3 crop->image map of transform_points_back_tf_batch, gt_xy the projected
 * ground-truth keypoints (image pixels), avail [batch*objects] in {0,1}.  loss = sum_n avail*mean_j cap(smoothL1(|gt - T(p)|))
 * / sum(avail); g_yx = weight * d loss / d coords_yx. */
int cp_kp_reproj_loss_f32(const float* coords_yx, const float* gt_xy, const float* affine, const float* avail, int batch, int objects,
                          int kp, float max_pixel_error, float weight, float* g_yx, double* loss_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CASAPOSE_HIP_H */
