/*
 * nicv2_hip.h - C ABI of the MI355X (gfx950) implementation of the reference's per-sample hot path:
 *               feature-grid gather + positional encoding + tiny GELU MLP, forward and backward.
 *
 * The reference (21K1113/Neural_Image_Compression_V2) is pure Python and has no FFI of its own; the
 * boundary is the Python call surface of Projects/fp_def.py, Projects/utils.py, Projects/models.py and
 * the create_decoder_input / ColorDecoder / decode_image part of Projects/image_compression.py.
 * Every entry point below names the reference lines it replaces.  The host mirror of those Python
 * signatures lives in neural_image_compression_v2_amd/ and binds this library with ctypes
 * (INTEGRATION.md shows the stub a reference maintainer would add).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer (HBM) unless the name ends in _host; tensors are dense fp32.
 *  - a grid is [C, (Z,) Y, X] like the reference (fp_def.py:54,76); the FIRST sample axis ("x",
 *    coord[0]) walks the LAST grid axis (fp_def.py:81-112).  All per-axis arrays in this header are
 *    given in sample-axis order (x, y, z).
 *  - samples of a crop are numbered x-outer ... z-inner, crops back to back (fp_def.py:124-129,
 *    image_compression.py:97); N = num_crops * extent[0] * extent[1] * extent[2].
 *  - decoder weights are nn.Linear layout [out, in] row-major (image_compression.py:57-64).
 *  - all functions are asynchronous on `stream` (a hipStream_t passed as void*), allocate nothing and
 *    never synchronise, so a caller may capture them into a hipGraph.
 *  - return value: 0 on success, a positive hipError_t from the runtime, or a negative NIC_E_* code.
 */
#ifndef NICV2_HIP_H
#define NICV2_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NIC_ABI_VERSION 9

enum {
    NIC_OK = 0,
    NIC_E_NULL = -1,        /* required pointer is null */
    NIC_E_UNSUPPORTED = -2, /* dim / method / channels / hidden combination has no kernel */
    NIC_E_SHAPE = -3,       /* extents, node counts or origins inconsistent (index would leave the grid) */
    NIC_E_WORKSPACE = -4,   /* workspace smaller than nic_workspace_bytes() */
    NIC_E_ARG = -5          /* other invalid argument */
};

/* pe_mode */
enum { NIC_PE_TRIANGULAR = 0 /* utils.py:211-227 */, NIC_PE_SINUSOIDAL = 1 /* utils.py:198-208 */ };
/* g1_weight_mode */
enum {
    NIC_G1_REFERENCE = 0,  /* bilinear in 2D; the reference's permuted trilinear products in 3D (fp_def.py:176-183) */
    NIC_G1_TEXTBOOK = 1,   /* true trilinear */
    NIC_G1_UNWEIGHTED = 2  /* plain corner sum: what the reference does when step_number == 2 (fp_def.py:136) */
};
/* noise_mode */
enum {
    NIC_NOISE_NONE = 0,
    NIC_NOISE_TENSOR = 1,  /* caller supplies the [N, Cin] noise (parity with torch.rand_like, image_compression.py:250) */
    NIC_NOISE_KERNEL = 2   /* generated in-kernel: Threefry-4x32-12 keyed by (seed, offset, global sample id, channel block) */
};

/* Geometry of one launch: which grid pair, which samples.  Mirrors the arguments of
 * create_g0_g1 / create_g0_g1_3d / create_g0_g1_3d_v2 (fp_def.py:115,148,187) and of
 * create_decoder_input_2d/_3d/_3d_v2 (image_compression.py:71,103,137). */
struct nic_step_tail;
typedef struct nic_path_desc {
    int32_t dim;             /* 2 or 3 */
    int32_t method;          /* 1: 2D.  3: 3D, 8 raw G0 corners.  4: 3D, 4 tetrahedral G0 corners (fp_def.py:107-112) */
    int32_t channels;        /* FEATURE_PYRAMID_CHANNELS (var2.py:68) */
    int32_t pe_channels;     /* PE_CHANNELS (var2.py:69) */
    int32_t hidden;          /* HIDDEN_LAYER_CHANNELS (var2.py:72) */
    int32_t pe_mode;         /* NIC_PE_* */
    int32_t g1_weight_mode;  /* NIC_G1_* */
    int32_t log2_step;       /* step_number = 2^log2_step = 2^(mip - 2(fl+1))  (image_compression.py:79) */
    float lod_value;         /* value of the LOD channel = mip_level (image_compression.py:95) */
    int32_t num_crops;
    int32_t extent[3];       /* samples per axis of one crop (sample_number); extent[2] = 1 in 2D */
    int32_t g0_nodes[3];     /* nodes per axis of G0 = fp[2*fl]  (x, y, z order) */
    int32_t g1_nodes[3];     /* nodes per axis of G1 = fp[2*fl+1] */
    float pe_div[8];         /* sinusoidal div_term, fp32, pe_channels/2 entries (utils.py:202) */
    int32_t noise_mode;      /* NIC_NOISE_* */
    int32_t num_bits;        /* FP_BITS: noise amplitude 2^-num_bits (image_compression.py:250) */
    uint64_t noise_seed;
    uint64_t noise_offset;  /* e.g. the training step */
    int64_t sample_base;     /* global id of this launch's sample 0 (data-parallel shards: keeps the noise world-size invariant) */
    float loss_scale;        /* 1 / (3 * N_global): the MSELoss mean (image_compression.py:259) */
    int32_t flags;           /* NIC_FLAG_* (0 is always valid) */
    int32_t passes;          /* training entry points: every crop is sampled `passes` times in one launch (0 and 1 = once); pass k of
                              * crop c has the sample ids (c * passes + k) * n_per_crop + ..., i.e. its own noise and its own target /
                              * dy rows (N = num_crops * passes * n_per_crop) - the same result as listing the crop `passes` times, but
                              * a cell's gradients are summed over all passes before they go to memory (stripe-sharded multi-GPU steps).
                              * Every other entry point: 0 or 1. */
    int32_t dz_scale_log2;   /* NIC_FLAG_FP16 only: dZ is carried as 2^dz_scale_log2 x dZ.  0: nic_fused_forward_backward / _img / _img_dev choose it from
                              * loss_scale (2 loss_scale 2^k in [4, 8)); nic_fused_backward_dy takes 2^0 - pass the exponent that brings the incoming
                              * dY to O(1) (a mean-squared-error dY is 2 (y - t) / (3 N): k = round(log2(3 N)) + 1) */
    int32_t max_workgroups;  /* 0: the launch may fill the chip (one persistent workgroup per CU, two for inference).  n > 0: at most n
                              * workgroups (rounded down to a multiple of 8, at least 8): independent fits launched on separate
                              * streams (BASELINE config 5) then share the CUs side by side instead of queueing behind each other's
                              * persistent grids. */
    const struct nic_step_tail *tail; /* training entry points (nic_fused_forward_backward, _img, _img_dev, nic_fused_backward_dy,
                              * nic_fused_ml_forward_backward): null, or the optimiser step that follows the launch (image_compression.py:266-269) -
                              * the reduction of the decoder-gradient records then also runs Adam over every listed tensor (below: nic_step_tail);
                              * every other entry point: must be null */
} nic_path_desc;
/* Every crop origin is a multiple of the cell size 1 / step_number (1 when step_number >= 1), e.g. whole-image passes from
 * origin 0: the launch then covers extent / cell blocks per axis instead of the unaligned upper bound extent / cell + 1
 * (origins live on the device, the library cannot look). Setting it for unaligned origins drops samples. */
#define NIC_FLAG_ORIGINS_ALIGNED 1
/* Matrix products on the bf16 matrix pipe with each fp32 operand carried as a hi + lo bf16 pair (16 significant bits, three
 * bf16 MFMAs per product, fp32 accumulation); activations, noise, loss and all accumulators stay fp32.
 *   2D training (nic_fused_forward_backward, _img, nic_fused_backward_dy): every product of the step, 1.8x faster;
 *   3D training: the four chained products (layer 1, layer 2 and their input-gradient transposes), 1.2x faster - the
 *     [sample][feature] bf16 images of the weight-gradient products do not fit the LDS next to the 3D weights;
 *   inference (nic_fused_forward, nic_fused_forward_u8), every layout: layers 1 and 2, 1.6-2x faster.
 * Agreement with the fp32 kernels: outputs ~3e-7, gradients <= 7e-6 relative (the fp32 kernels themselves sit ~1e-6 from the
 * CPU oracle).  The fp32-input MFMA blocks the wave's vector issue, the bf16 one does not. */
#define NIC_FLAG_SPLIT_BF16 2
/* With NIC_FLAG_SPLIT_BF16, 2D training: keep the step on the 4-wave x 32-sample kernel (one wave per SIMD) instead of the default
 * 8-wave x 16-sample kernel (two waves per SIMD; same arithmetic mode, different tiling and summation order).  Kept for
 * comparison runs and as the second implementation the parity tests hold the default against. */
#define NIC_FLAG_SPLIT_TILE32 4
/* With NIC_FLAG_SPLIT_BF16, 2D, n_linear = 3: run the step (or decode) on the depth-generic kernel that serves n_linear = 5 (4 waves per
 * workgroup, the layer loop): the cross-check of that kernel against the dedicated 3-layer kernels. */
#define NIC_FLAG_MLPN 8
/* 16-bit grid STORAGE (the reference's FP_NUM_DTYPE = 16 maps its grids to torch.float16, utils.py:301-313; its own 16-bit run does not
 * train, readme.md:9): g0 / g1 point at bfloat16 or IEEE half arrays of the usual [C, Y, X] shape; every value is widened to fp32 in the
 * gather and all arithmetic, the gradients (dense fp32 tensors of the grids' shapes) and the optimiser state stay fp32 - nic_adam_multi
 * keeps an fp32 master and writes the 16-bit mirror (nic_adam_tensor.param16).  2D with NIC_FLAG_SPLIT_BF16 (nic_fused_forward,
 * nic_fused_forward_backward, _img, nic_fused_backward_dy), or any layout with NIC_FLAG_BF16 (the training entry points). */
#define NIC_FLAG_GRID_BF16 16
#define NIC_FLAG_GRID_FP16 32
/* PLAIN bf16 matrix products (BASELINE.json's "bf16"): every product operand - weights, noisy inputs, GELU outputs,
 * the dZ of every layer, the stored GELU derivatives - is ONE bf16 value, accumulation / biases / activations / loss / grid-gradient
 * sums are fp32 (csrc/fused_q16.hpp lists the rounding points; oracle/nic_oracle.py::mlp_forward_backward_bf16 restates them).
 * Training entry points (nic_fused_forward_backward, _img, nic_fused_backward_dy), every layout (2D, 3D methods 3 and 4),
 * n_linear 3 or 5, fp32 or 16-bit grid storage; 8 waves x 16 samples, two waves per SIMD.  Takes precedence over NIC_FLAG_SPLIT_BF16.
 * Results are checked against the precision-emulating oracle at 1e-3 (outputs) and stay within ~1e-2 of the fp32 arithmetic. */
#define NIC_FLAG_BF16 64
/* PLAIN fp16 matrix products: NIC_FLAG_BF16's kernels, rounding points and entry points with IEEE half operands instead of bfloat16 (the reference's own
 * 16-bit type: FP_NUM_DTYPE / MLP_NUM_DTYPE = 16 map to torch.float16, utils.py:301-313; BASELINE config 3's "fp16"): v_mfma_f32_16x16x32_f16 runs at the
 * bf16 rate with 11 significant bits instead of 8 - outputs ~2e-5 and gradients ~1e-3 from the fp32 arithmetic, checked against the fp32 oracle directly.
 * dZ of a mean over millions of samples is far below the half range (2 loss_scale ~ 1e-7), so the kernel carries 2^k dZ and multiplies every sum of dZ
 * products by 2^-k where it leaves the kernel (exact): k = nic_path_desc.dz_scale_log2, or - 0 - chosen from loss_scale by the MSE entry points.  Default
 * channel counts; takes precedence over NIC_FLAG_BF16 and NIC_FLAG_SPLIT_BF16. */
#define NIC_FLAG_FP16 128
/* The `origins` argument of the FUSED entry points (nic_fused_*) points to HOST memory: num_crops x dim int32 values, num_crops <= NIC_ORIGINS_INLINE_MAX,
 * read during the call and handed to the kernel by value.  The host loop of the reference draws its crop origins on the host
 * (image_compression.py:26-50): a step then needs no upload (a 5 us copy in front of a 0.1 ms kernel) and the kernel no dependent global load in front
 * of its gathers.  Not for the captured loop (nic_fused_forward_backward_img_dev reads the origins nic_sampler_step_begin wrote) nor for the
 * layer-wise entry points (nic_encode*, nic_gather_corners, ..: NIC_E_ARG). */
#define NIC_FLAG_ORIGINS_HOST 256
#define NIC_ORIGINS_INLINE_MAX 16

/* ColorDecoder parameters (image_compression.py:54-68): state_dict keys decoder.{0,2,4}.{weight,bias}.  The reference hard-codes 3
 * Linear layers (n_linear = 3, or 0); n_linear = 5 is the "4 x 64" decoder of BASELINE.json's north star - Linear(Cin,H), three
 * Linear(H,H), Linear(H,3), GELU between, Sigmoid at the end (keys decoder.{0,2,4,6,8}) - supported by the fused 2D entry points with
 * NIC_FLAG_SPLIT_BF16 (nic_fused_forward, nic_fused_forward_backward, _img, nic_fused_backward_dy); everything else returns
 * NIC_E_UNSUPPORTED for it.  Layer i lives in w[i] / b[i]; the output layer is the last one. */
#define NIC_MAX_LINEAR 5
typedef struct nic_mlp {
    const float *w[NIC_MAX_LINEAR]; /* [H,Cin], [H,H] x (n_linear - 2), [3,H] */
    const float *b[NIC_MAX_LINEAR]; /* [H], [H] x (n_linear - 2), [3] */
    int32_t n_linear;               /* 3 (0 means 3) or 5; nic_decoder_general_*: 2 .. NIC_MAX_LINEAR */
    int32_t reserved;
} nic_mlp;

typedef struct nic_mlp_grads {
    float *w[NIC_MAX_LINEAR];
    float *b[NIC_MAX_LINEAR];
} nic_mlp_grads;

int nic_abi_version(void);
const char *nic_error_string(int code);

/* Number of decoder input channels Cin (var2.py:114-118). */
int nic_decoder_input_channels(int dim, int method, int channels, int pe_channels);

/* Bytes of scratch the *_backward entry points need (per-wave partial sums of the decoder gradients,
 * reduced in a fixed order by a second kernel so decoder gradients and loss are run-to-run bit-stable). */
size_t nic_workspace_bytes(const nic_path_desc *desc);

/* ---- encode only: replaces create_decoder_input_2d/_3d/_3d_v2 and finally_decode_input_*
 *      (image_compression.py:71-211).  out = [N, Cin].  origins = int32 [num_crops, dim]. */
int nic_encode(const nic_path_desc *desc, const float *g0, const float *g1, const int32_t *origins,
               float *out, void *stream);

/* ---- autograd backward of nic_encode: dx = [N, Cin] -> scatter-add into g0_grad / g1_grad (shapes of g0 / g1, NOT
 *      zeroed here: fp32 atomic adds, the reference's index_put_(accumulate=True) from loss.backward(),
 *      image_compression.py:265).  The PE and LOD columns of dx have no parameters behind them and are ignored. */
int nic_encode_backward(const nic_path_desc *desc, const int32_t *origins, const float *dx, float *g0_grad,
                        float *g1_grad, void *stream);

/* ---- encode, split outputs: replaces create_g0_g1 / _3d / _3d_v2 (fp_def.py:115-223) for ONE crop.
 *      out = [K0*C + K1*C + P*dim, n] rows: raw G0 corners, weighted G1 corners (not summed), PE. */
int nic_encode_split(const nic_path_desc *desc, const float *g0, const float *g1, const int32_t *origins,
                     float *out, void *stream);

/* ---- corner gathers on explicit index vectors: replaces create_g / create_g_3d / create_g_3d_v2 (fp_def.py:81-112).
 *      grid = [C, (nz,) ny, nx]; xi / yi / zi = int32 [n] (zi null in 2D); corner_set 0: the 4 corners of 2D,
 *      1: the 8 corners of 3D, 2: the 4 tetrahedral corners; out = [K, C, n] in the reference's corner order.
 *      Indices are clamped into the grid (the reference would raise IndexError). */
int nic_gather_corners(const float *grid, int channels, int nx, int ny, int nz, const int32_t *xi, const int32_t *yi,
                       const int32_t *zi, int64_t n, int corner_set, float *out, void *stream);

/* ---- positional encodings on explicit coordinates (utils.py:198-227).  coord = [dim, n], out = [P*dim, n]. */
int nic_positional_encoding(const float *coord, int64_t n, int dim, int pe_channels, int pe_mode,
                            const float *pe_div_host, float *out, void *stream);

/* ---- LUT positional encoding gather: replaces TriangularPositionalEncoding1D.forward
 *      (positional_encoding.py:36-42).  lut = [rows, seq_len], coord = int64 [b, L], out = [b, rows, L]. */
int nic_lut_gather(const float *lut, int rows, int seq_len, const int64_t *coord, int64_t b, int64_t L,
                   float *out, void *stream);

/* ---- decoder on explicit inputs: replaces ColorDecoder.forward (image_compression.py:66-68).
 *      x = [n, cin], y = [n, 3]. */
int nic_decoder_forward(const nic_mlp *mlp, const float *x, int64_t n, int cin, int hidden, float *y, void *stream);

/* ---- its autograd backward: dy = [n,3] -> dx = [n,cin] (may be null) and the six parameter gradients
 *      (overwritten, not accumulated). */
int nic_decoder_backward(const nic_mlp *mlp, const float *x, const float *dy, int64_t n, int cin, int hidden,
                         float *dx, const nic_mlp_grads *grads, void *workspace, size_t workspace_bytes, void *stream);

/* ---- the same decoder for ANY Cin >= 1, HIDDEN_LAYER_CHANNELS >= 1 (var2.py:72) and 2 <= n_linear <= NIC_MAX_LINEAR, on explicit
 *      inputs: layer-wise LDS-tiled fp32 products (csrc/decoder_general.hip), the route of every flag combination the fused kernels do
 *      not specialise (ColorDecoder.forward, image_compression.py:54-68, with HIDDEN_LAYER_CHANNELS / FEATURE_PYRAMID_CHANNELS /
 *      PE_CHANNELS of any value).  The sample axis is walked in chunks; the workspace (nic_decoder_general_workspace_bytes for the same
 *      n, cin, hidden, n_linear; training = 1 for the backward call) holds one chunk's activations, their GELU derivatives and the
 *      per-slice weight-gradient sums.  _backward recomputes the forward pass of a chunk itself: x and dy in, dx (may be null) and the
 *      2 n_linear parameter gradients out (overwritten; null entries are skipped; fixed summation order, bit-stable run to run). */
size_t nic_decoder_general_workspace_bytes(int64_t n, int cin, int hidden, int n_linear, int training);
int nic_decoder_general_forward(const nic_mlp *mlp, const float *x, int64_t n, int cin, int hidden, float *y, void *workspace,
                                size_t workspace_bytes, void *stream);
int nic_decoder_general_backward(const nic_mlp *mlp, const float *x, const float *dy, int64_t n, int cin, int hidden, float *dx,
                                 const nic_mlp_grads *grads, void *workspace, size_t workspace_bytes, void *stream);

/* ---- HIDDEN_LAYER_CHANNELS below 64 on the FUSED kernels (which are built for 64): a decoder with H hidden units IS the decoder with 64 units
 *      whose extra units have zero weights and biases (pre-activation 0, GELU(0) = 0, zero outgoing weights: exact zero gradients).
 *      nic_decoder_pad writes the zero-padded copies ([H, K] -> [hidden_padded, K'], the output layer [3, H] -> [3, hidden_padded]) into the
 *      caller's `dst` tensors, the fused entry points run on them with desc->hidden = hidden_padded, nic_decoder_unpad copies the real block of
 *      every padded gradient into the caller's gradient tensors (null entries of `dst` are skipped).  One launch each; fused.PaddedMlp does this
 *      for every fused entry point of the Python side. */
int nic_decoder_pad(const nic_mlp *src, int cin, int hidden, int hidden_padded, const nic_mlp_grads *dst, void *stream);
int nic_decoder_unpad(const nic_mlp_grads *src_padded, int n_linear, int cin, int hidden, int hidden_padded, const nic_mlp_grads *dst,
                      void *stream);

/* ---- fused inference: encode + (optional noise) + decoder.  Replaces finally_decode_input_* + arc_decoder(x)
 *      inside decode_image (image_compression.py:313-345).  y = [N, 3]. */
int nic_fused_forward(const nic_path_desc *desc, const float *g0, const float *g1, const int32_t *origins,
                      const nic_mlp *mlp, const float *noise, float *y, void *stream);

/* ---- the same training step with the target read straight from a resident image instead of an [N,3] tensor (SURVEY 8f
 *      rank 3; the sampler's slicing / reshape / stack of image_compression.py:37-47 disappears): sample i of crop k takes
 *      image[c][origin_k + i].  `data` is the reference's dataset tensor [3, S0, S1(, S2)] (first spatial axis = first sample
 *      axis), contiguous, fp32 - or uint8 codes with target = u / den, correctly rounded like torch's true division (den 255:
 *      ToTensor, image_compression.py:436-440; den 256: the 3D loader, :474).  Every origin + extent must lie inside `size`. */
typedef struct nic_target_image {
    const void *data;
    int32_t is_u8;       /* 0: fp32 planar [3, S0, S1(, S2)].  1: uint8 planar, target = u / den.  2: uint8 RGBX interleaved
                          *    [S0, S1(, S2)] dwords R | G << 8 | B << 16 (nic_rgbx_interleave / nic_rgbx_downsample2): the three
                          *    targets of a sample are ONE load, target = byte / den */
    float den;           /* uint8 only */
    int32_t size[3];     /* S0, S1, S2 (1 for 2D) */
    int32_t reserved;
} nic_target_image;
int nic_fused_forward_backward_img(const nic_path_desc *desc, const float *g0, const float *g1, const int32_t *origins,
                                   const nic_mlp *mlp, const float *noise, const nic_target_image *image, float *y, float *loss,
                                   float *g0_grad, float *g1_grad, const nic_mlp_grads *grads, void *workspace,
                                   size_t workspace_bytes, void *stream);

/* ---- device-side sampler (image_compression.py:26-50, 221-226).  The reference draws the LOD with Python's
 *      `random` and the crop origins with torch.randint on the host and uploads them every step; this counter-based generator
 *      (Threefry-4x32-12 keyed by seed, step and crop; csrc/nic_device.hpp::sampler_block) gives the same LAWS - LOD uniform on
 *      0..max_mip, or floor(-log2 U / 2) = clz(word) >> 1; origins uniform on [0, range) per axis - without host RNG state:
 *      the LOD decides the launch geometry, so it is evaluated on the host (pure function, no GPU); the origins are written by a
 *      tiny kernel straight into the int32 [num_crops, dim] device buffer the fused entry points read - no upload per step.
 *      nic_sampler_origins_host is the same function on the host (tests, debugging). */
int nic_sampler_lod_host(uint64_t seed, uint64_t step, int uniform_distribution, int max_mip_level);
int nic_sampler_origins_host(uint64_t seed, uint64_t step, int num_crops, int dim, int32_t range, int32_t *origins_host);
int nic_sampler_draw_origins(uint64_t seed, uint64_t step, int num_crops, int dim, int32_t range, int32_t *origins, void *stream);

/* ---- resident RGBX target images: planar uint8 [3][n] (the codes ToTensor / the 3D loader divide, image_compression.py:436-440, 474)
 *      -> n dwords; and the next level of a 2D mip chain by a 2 x 2 box filter with round-to-nearest (the reference resizes with
 *      torchvision's Resize, :429-477: its filter is not reproduced - targets at mip > 0 are this library's own definition;
 *      mip 0, the only level of the no-mip default, is exact). */
int nic_rgbx_interleave(const uint8_t *planar, int64_t n, uint32_t *rgbx, void *stream);
int nic_rgbx_downsample2(const uint32_t *src_rgbx, int s0, int s1, uint32_t *dst_rgbx, void *stream);
/* ---- the reference's OWN mip filter: transforms.Resize on the PIL image (image_compression.py:434-440) is Pillow's two-pass fixed-point resize
 *      (src/libImaging/Resample.c; torchvision is a thin wrapper: functional.resize -> Image.resize(size, BILINEAR)).  One pass along `axis` of an RGBX
 *      image [s0][s1] (axis 1 = the contiguous one; Pillow runs it first, then axis 0): out = clip8((2^21 + sum_t pixel[lo + t] * kk[o][t]) >> 22) per
 *      channel, taps [lo, lo + count) = bounds[o][0..1], kk int32 [out_size][ksize] in 2^-22 units - precompute_coeffs + normalize_coeffs_8bpc, restated
 *      on the host by sampler.resize_coeffs.  Every level is resized from the ORIGINAL image, like the reference.  Bit-exact against Pillow
 *      (tests/test_host_cpu.py pins the coefficients, tests/test_gpu_general.py the images). */
int nic_rgbx_resample_axis(const uint32_t *src_rgbx, int s0, int s1, int axis, int out_size, const int32_t *bounds, const int32_t *kk, int ksize,
                           uint32_t *dst_rgbx, void *stream);

/* ---- decode straight from the stored codec (image_compression.py:307-346 after fp_load, fp_def.py:258-263):
 *      the grids are the uint8 tensors fp_savable wrote (models.py:61-64), dequantised in-kernel exactly like load4fp
 *      (models.py:68-71), so the result is bit-identical to nic_load4fp_u8 + nic_fused_forward at a quarter of the grid bytes.
 *      desc->num_bits is the codec's bit depth (1..8); desc->noise_mode must be NIC_NOISE_NONE.  n_linear 3: every layout; n_linear 5: 2D
 *      (split-bf16 products, like nic_fused_forward for that depth).
 *      y: fp32 [N,3] and/or y_u8: quantize_to_bit(y) as bytes (models.py:39-40) - at least one. */
int nic_fused_forward_u8(const nic_path_desc *desc, const uint8_t *g0_u8, const uint8_t *g1_u8, const int32_t *origins,
                         const nic_mlp *mlp, float *y, uint8_t *y_u8, void *stream);

/* ---- fused training step core: encode + noise + decoder + MSE loss + full backward.  Replaces
 *      image_compression.py:239-265 (create_decoder_input_*, rand_like noise, decoder(...), MSELoss, backward).
 *      target = [N,3].  y may be null.  loss = 1 float (the mean, desc->loss_scale * sum of squared error).
 *      g0_grad / g1_grad have the shapes of g0 / g1 and must be ZEROED by the caller (the kernel adds into
 *      them with fp32 atomics, like index_put_(accumulate=True)); decoder gradients and loss are overwritten. */
int nic_fused_forward_backward(const nic_path_desc *desc, const float *g0, const float *g1, const int32_t *origins,
                               const nic_mlp *mlp, const float *noise, const float *target, float *y, float *loss,
                               float *g0_grad, float *g1_grad, const nic_mlp_grads *grads,
                               void *workspace, size_t workspace_bytes, void *stream);

/* ---- same, for an arbitrary upstream gradient dy = [N,3] instead of the built-in MSE (autograd backward of
 *      nic_fused_forward; the forward is recomputed). */
int nic_fused_backward_dy(const nic_path_desc *desc, const float *g0, const float *g1, const int32_t *origins,
                          const nic_mlp *mlp, const float *noise, const float *dy,
                          float *g0_grad, float *g1_grad, const nic_mlp_grads *grads,
                          void *workspace, size_t workspace_bytes, void *stream);

/* ---- quantisers / codec (models.py:55-71, fp_def.py:227-263).  In-place allowed (dst == src). */
int nic_quantize(const float *src, float *dst, int64_t n, int num_bits, void *stream);              /* floor(x*(2^b-1)+.5)/(2^b-1) */
int nic_quantize_to_bit(const float *src, float *dst, int64_t n, int num_bits, void *stream);       /* models.py:39-40 */
int nic_clamp(float *x, int64_t n, float lo, float hi, void *stream);                                /* fp_quantize_clamp */
int nic_save4fp_u8(const float *src, uint8_t *dst, int64_t n, int num_bits, void *stream);          /* models.py:61-64 */
int nic_load4fp_u8(const uint8_t *src, float *dst, int64_t n, int num_bits, void *stream);          /* models.py:68-71 */

/* ---- PSNR with peak = 2^num_bits (utils.py:117-130): out[0] = mse, out[1] = psnr dB (inf when mse == 0). */
int nic_psnr(const float *a, const float *b, int64_t n, int num_bits, float *out2, void *workspace, size_t workspace_bytes, void *stream);

/* ---- fused Adam step + in-place clamp for one parameter tensor (image_compression.py:266-269, 361-365):
 *      torch.optim.Adam semantics (no weight decay, no amsgrad), bias-corrected with step count `step` (1-based).
 *      clamp_lo > clamp_hi disables the clamp.  The hyper-parameters are doubles because torch's are Python floats: the library
 *      forms (float)(1 - beta), (float)(lr / bias_correction1) exactly as torch does (1.0f - 0.999f != 0.001f).  The clamp keeps
 *      NaN (torch.clamp_ propagates it; a diverged grid must not be silently reset to clamp_lo). */
int nic_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, double lr,
                  double beta1, double beta2, double eps, int64_t step, float clamp_lo, float clamp_hi, void *stream);

/* ---- the whole optimiser step in ONE launch: Adam for every listed tensor - grids at lr 0.01, decoder
 *      at lr 0.005, each already scaled by the caller's CosineAnnealingLR factor - with the fp_quantize_clamp of the grids
 *      folded in (image_compression.py:266-269, 361-365; fp_def.py:227-232).  torch keeps one step counter per parameter
 *      (a parameter whose .grad is None is skipped and does not advance), hence `step` per tensor.  The tensor table is
 *      read on the host and passed to the kernel by value: at most NIC_ADAM_MAX_TENSORS per call. */
#define NIC_ADAM_MAX_TENSORS 32
typedef struct nic_adam_tensor {
    float *param;
    const float *grad;
    float *exp_avg;
    float *exp_avg_sq;
    int64_t n;
    int64_t step;            /* 1-based step count of THIS tensor */
    double lr;
    float clamp_lo, clamp_hi; /* clamp_lo > clamp_hi: no clamp */
    void *param16;            /* null, or a 16-bit mirror of `param` (same element count) rewritten with the rounded new values */
    int32_t param16_kind;     /* 1: bfloat16, 2: IEEE half (round to nearest even) */
    int32_t flags;            /* NIC_ADAM_ZERO_GRAD: the launch also zeroes `grad` (written through the const pointer) once it has been read - the
                                 gradient bucket of an atomically accumulating step is clean for the next step without a fill kernel; 0 otherwise */
    int32_t reps;             /* 0 / 1: one contiguous run of n elements.  r > 1: r runs of n elements each - run k starts k * rep_stride elements into
                                 param / grad / param16 and k * state_rep_stride into exp_avg / exp_avg_sq: a block of node rows of EVERY channel of a
                                 grid [C, rows, ..] in one entry (rep_stride = the channel plane) with moments allocated for those rows only
                                 (state_rep_stride = n) - the stripe-owned optimiser of the multi-GPU step */
    int32_t reserved;
    int64_t rep_stride, state_rep_stride;
} nic_adam_tensor;
#define NIC_ADAM_ZERO_GRAD 1
#define NIC_ADAM_SCHED_COL1 2 /* nic_adam_multi_dev: this tensor takes its step size from column 1 of the schedule (the decoder group), else column 0 */
int nic_adam_multi(const nic_adam_tensor *tensors, int count, double beta1, double beta2, double eps, void *stream);

/* ---- the optimiser step as the TAIL of the fused training step (nic_path_desc.tail): image_compression.py:263-269 (backward, optimizer.step,
 *      fp_quantize_clamp) in TWO launches instead of three or four.  A training entry point ends with the reduction of the per-workgroup
 *      decoder-gradient records; with a tail that launch gets more blocks: the first `n_stream` tensors (the grids: their gradients are complete
 *      when the fused kernel ends) are streamed by the extra blocks exactly as nic_adam_multi would - NIC_ADAM_ZERO_GRAD, clamp and 16-bit
 *      mirror included - WHILE the reduction walks the records (it is latency-bound, the streaming HBM-bound); the remaining tensors are the
 *      decoder's: their .grad must be the nic_mlp_grads buffers of the same call, and the thread that finishes element i of such a gradient
 *      updates parameter i on the spot.  Same arithmetic as nic_adam_multi after the plain entry point, bit for bit.  count <= NIC_ADAM_MAX_TENSORS;
 *      sched / sched_rows: nic_adam_multi_dev's device schedule, read with the `step_dev` of nic_fused_forward_backward_img_dev (null: the
 *      per-tensor .step / .lr are used). */
/* measurement hook (bench.py's roofline leg): the NEXT training entry point called on this thread records `hip_event` (a hipEvent_t) on its
 * stream right after its fused kernel - before the reduction / tail launch - so that HIP events can bracket the dominant kernel alone inside
 * a timed region of whole steps.  One-shot; null clears it. */
int nic_mark_kernel_end(void *hip_event);

typedef struct nic_step_tail {
    const nic_adam_tensor *tensors;
    int32_t count;
    int32_t n_stream;
    double beta1, beta2, eps;
    const float *sched;
    int64_t sched_rows;
} nic_step_tail;

/* ---- hipGraph-captured training loops (no reference counterpart: the reference's loop is host Python, image_compression.py:215-303).
 *      The reference's own launchers run 320 000 steps of 8 x 32^3 samples: the GPU needs ~0.1 ms per step, the host loop twice that.
 *      With the step number in DEVICE memory one captured sequence [nic_sampler_step_begin -> nic_fused_forward_backward_img_dev ->
 *      nic_adam_multi_dev] serves every step and is replayed without the host touching a single argument:
 *        counters   int64 [2] device: [0] = next step to run, [1] = the step in flight (what the other two entry points read through
 *                   `step_dev` = counters + 1)
 *        nic_sampler_step_begin  t = counters[0]; counters[1] = t; counters[0] = t + 1; loss_hist[t - 1] = *loss_slot (the loss the previous
 *                   step's reduction wrote; either pointer may be null); origins of step t as nic_sampler_draw_origins(seed, t, ..)
 *        nic_fused_forward_backward_img_dev  nic_fused_forward_backward_img with noise offset desc->noise_offset + *step_dev (in-kernel noise
 *                   or none; kernels: 2D NIC_FLAG_SPLIT_BF16 with 3 layers, or NIC_FLAG_BF16 - otherwise NIC_E_UNSUPPORTED)
 *        nic_adam_multi_dev  nic_adam_multi with the per-step scalars read from row min(*step_dev, sched_rows - 1) of `sched`, a device
 *                   table [sched_rows][4] of floats the caller fills once: {lr_0 / bias_correction1, lr_1 / bias_correction1,
 *                   sqrt(bias_correction2), 0} for Adam step row + 1 (formed in double and cast once, like nic_adam_multi does per call);
 *                   nic_adam_tensor.step / .lr are ignored, NIC_ADAM_SCHED_COL1 selects the column. */
int nic_sampler_step_begin(uint64_t seed, int64_t *counters, int num_crops, int dim, int32_t range, int32_t *origins,
                           const float *loss_slot, float *loss_hist, int64_t hist_len, void *stream);
int nic_fused_forward_backward_img_dev(const nic_path_desc *desc, const float *g0, const float *g1, const int32_t *origins,
                                       const nic_mlp *mlp, const nic_target_image *image, float *loss, float *g0_grad, float *g1_grad,
                                       const nic_mlp_grads *grads, const int64_t *step_dev, void *workspace, size_t workspace_bytes,
                                       void *stream);
int nic_adam_multi_dev(const nic_adam_tensor *tensors, int count, double beta1, double beta2, double eps, const float *sched,
                       int64_t sched_rows, const int64_t *step_dev, void *stream);

/* ---- MULTI-LEVEL fused step (no reference semantics: the reference reads ONE level pair per sample, fp_def.py:24-34, image_compression.py:76-79;
 *      BASELINE.json config 2 names a "16-level grid", SURVEY 0 asks for the level count as a kernel parameter).  A sample gathers from the first
 *      `levels` level pairs AT ONCE and the decoder sees their encodings concatenated:
 *          x = [enc_0 | .. | enc_{levels-1} | lod],  enc_l = [G0_l corners (4 C) | blended G1_l (C) | PE_l (2 P)]
 *      enc_l being exactly what create_g0_g1 (fp_def.py:115-145) computes for pair l at step_number 2^(desc->log2_step - 2 l) (image_compression.py:79:
 *      4^-(l+1) at mip 0), the positional encoding on THAT pair's G1-cell coordinate; Cin = levels (5 C + 2 P) + 1 (decoder weights mlp->w[0] = [H, Cin]).
 *      2D (desc->dim = 2, method = 1), fp32 grids, plain-bf16 products (the arithmetic of NIC_FLAG_BF16: csrc/fused_q16.hpp), in-kernel / tensor / no
 *      noise, ONE launch: gathers, noise, decoder forward, loss, backward, one atomic flush per touched cell and pair.  Fused kernels exist for
 *      (levels, C, n_linear) in {(2,4,3), (3,4,3), (5,4,3), (2,4,5), (3,4,5), (2,12,3), (3,12,3)} with P = 6, H = 64 (what fits the 160 KB of LDS beside the
 *      [64, Cin] weight image); everything else returns NIC_E_UNSUPPORTED (the caller composes nic_encode x levels + the general decoder instead).
 *      desc->g0_nodes / g1_nodes are ignored (pairs carry their own); desc->extent, num_crops, origins, noise, loss_scale, sample_base as in
 *      nic_fused_forward_backward.  Gradients are ADDED into pairs->g0_grad / g1_grad (dense fp32 tensors of the grids' shapes, zeroed by the caller). */
#define NIC_ML_MAX_LEVELS 5
typedef struct nic_ml_pairs {
    int32_t levels;                               /* 2 .. NIC_ML_MAX_LEVELS */
    int32_t reserved;
    const float *g0[NIC_ML_MAX_LEVELS];           /* [C, ny, nx] fp32 */
    const float *g1[NIC_ML_MAX_LEVELS];
    float *g0_grad[NIC_ML_MAX_LEVELS];            /* training entry point only */
    float *g1_grad[NIC_ML_MAX_LEVELS];
    int32_t g0_nodes[NIC_ML_MAX_LEVELS][2];       /* nodes per axis (x, y) */
    int32_t g1_nodes[NIC_ML_MAX_LEVELS][2];
} nic_ml_pairs;
int nic_fused_ml_forward_backward(const nic_path_desc *desc, const nic_ml_pairs *pairs, const int32_t *origins, const nic_mlp *mlp,
                                  const float *noise, const float *target, float *y, float *loss, const nic_mlp_grads *grads,
                                  void *workspace, size_t workspace_bytes, void *stream);
/* forward only (decode): y = [N, 3] */
int nic_fused_ml_forward(const nic_path_desc *desc, const nic_ml_pairs *pairs, const int32_t *origins, const nic_mlp *mlp, float *y,
                         void *stream);

/* ---- multi-GPU, stripe-sharded grids (no reference counterpart - the reference is single-device): the per-step exchange buffer
 *      [small | boundary rows of G0 | boundary rows of G1].  `small` = the head of the flat gradient bucket (loss + decoder gradients,
 *      n_small floats); a row set = `nrows` node rows (hyper-planes of the slowest spatial axis: `row_elems` contiguous floats) of every one of
 *      `channels` channel planes (`plane` floats apart) of one grid-gradient tensor.  nic_stripe_pack gathers everything into `buf`
 *      (layout: small, then per set [channel][row][row_elems]); the caller all-reduces `buf` over RCCL; nic_stripe_unpack writes the sums
 *      back.  One launch each (the torch formulation was index_select x 2 + cat + copy + index_copy x 2 per step). */
#define NIC_STRIPE_MAX_ROWS 15
typedef struct nic_row_set {
    float *base;             /* the gradient tensor [C, rows, row_elems...] */
    int64_t plane;           /* floats per channel */
    int32_t row_elems;       /* floats per node row */
    int32_t channels;
    int32_t nrows;           /* <= NIC_STRIPE_MAX_ROWS */
    int32_t rows[NIC_STRIPE_MAX_ROWS];
} nic_row_set;
int nic_stripe_pack(const float *small_buf, int64_t n_small, const nic_row_set *sets, int nsets, float *buf, void *stream);
int nic_stripe_unpack(float *small_buf, int64_t n_small, const nic_row_set *sets, int nsets, const float *buf, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* NICV2_HIP_H */
