/* smirk_hip.h — C ABI of libsmirk_hip.so: the MI355X (gfx950) implementation of SMIRK's per-frame hot path
 *   SmirkEncoder -> FLAME -> Renderer -> SmirkGenerator   (reference: demo.py:107-112,167-169; smirk_trainer.py:37-48,94)
 *
 * The reference has no FFI layer: its boundary is four Python nn.Modules.  Each entry point below is
 * what a binding for that module's forward() would call; the file:line it replaces is cited on every declaration.
 *
 * Conventions (all entry points)
 *   - every pointer is a DEVICE pointer to contiguous memory, fp32 unless the name says otherwise; the caller owns all
 *     buffers (inputs, outputs, workspace) — the library never allocates, frees or synchronises;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the null stream) and the call returns at once;
 *   - return value: 0 = SMIRK_OK, negative = error (smirk_strerror); nothing throws, nothing prints;
 *   - image tensors at the boundary are NCHW like the reference; internal activations are NHWC fp32;
 *   - results: raster indices bit-exact vs the reference algorithm, everything else within the fp32 tolerances
 *     stated in tests/ (DESIGN.md §Numerics).
 */
#ifndef SMIRK_HIP_H
#define SMIRK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMIRK_OK 0
#define SMIRK_ERR_BAD_ARG (-1)
#define SMIRK_ERR_WORKSPACE (-2)
#define SMIRK_ERR_LAUNCH (-3)
#define SMIRK_ERR_UNSUPPORTED (-4)

const char* smirk_strerror(int code);
/* Split-fp16 range flag.  Activations are stored as fp16 (hi, lo) pairs: a value of magnitude >= 65520 cannot be carried (hi becomes inf) and every
 * later layer would silently turn it into inf / NaN pixels — with the reference's fp32 tensors (smirk_generator.py:51-86, smirk_encoder.py:14-133) the same
 * checkpoint simply works.  Every kernel that writes the storage format audits what it stores and sets ONE sticky host-visible word; no entry of this library
 * synchronises to look at it.  peek: non-zero once any kernel THAT HAS FINISHED saw such a value since the last clear (the Python modules call it at the start of
 * each forward and raise SmirkHipError: "raise on the next call"; after a synchronisation of the caller's own it is exact).  clear: re-arm. */
unsigned smirk_range_flag_peek(void);
void smirk_range_flag_clear(void);

/* ABI version of this header (bumped on any signature change). */
int smirk_abi_version(void);

/* ------------------------------------------------------------------------------------------------------------------
 * FLAME  — replaces FLAME.forward (src/FLAME/FLAME.py:232-315) incl. lbs() (src/FLAME/lbs.py:140-227),
 *          batch_rodrigues (:274-305), batch_rigid_transform (:321-378), the dynamic-landmark LUT
 *          (FLAME.py:117-159) and vertices2landmarks (lbs.py:101-137).
 * ------------------------------------------------------------------------------------------------------------------ */
#define SMIRK_FLAME_KCHUNK 32 /* K padding granule of the packed blendshape basis */

typedef struct SmirkFlameModel {
    int32_t V;        /* 5023 vertices                                   (FLAME.py:61 v_template)            */
    int32_t VP;       /* V rounded up to 32 (row count of `dirs` planes)                                         */
    int32_t F;        /* 9976 faces                                      (FLAME.py:58 faces_tensor)          */
    int32_t n_shape;  /* 300                                             (FLAME.py:50)                       */
    int32_t n_exp;    /* 50                                                                                  */
    int32_t KP;       /* (n_shape+n_exp+36) rounded up to SMIRK_FLAME_KCHUNK                                 */
    int32_t n_static; /* 51 static FAN landmarks                         (FLAME.py:97-98)                    */
    int32_t n_dyn;    /* 17 dynamic contour landmarks per LUT row        (FLAME.py:99-100)                   */
    int32_t n_lut;    /* 79 LUT rows                                                                         */
    int32_t n_full;   /* 68                                              (FLAME.py:101-102)                  */
    int32_t n_mp;     /* 105                                             (FLAME.py:112-113)                  */
    int32_t _pad;
    const float* dirs;        /* [3][VP][KP]  packed basis: k<n_shape+n_exp -> shapedirs[v][c][k] (FLAME.py:67-69),
                                 next 36 -> posedirs[k][3v+c] (FLAME.py:71-73), rest 0; k contiguous            */
    const float* v_template;  /* [V][3]                                                                        */
    const float* lbs_weights; /* [V][5]                                (FLAME.py:78)                         */
    const float* jdirs;       /* [15][n_shape+n_exp]  J_regressor . shapedirs, joint-major (j*3+c)             */
    const float* jtemplate;   /* [15]                 J_regressor . v_template                               */
    const float* l_eyelid;    /* [V][3]                                (FLAME.py:81)                         */
    const float* r_eyelid;    /* [V][3]                                (FLAME.py:82)                         */
    const int32_t* faces;     /* [F][3]                                                                       */
    const int32_t* static_faces;  const float* static_bary;  /* [n_static], [n_static][3]                     */
    const int32_t* dyn_faces;     const float* dyn_bary;     /* [n_lut][n_dyn], [n_lut][n_dyn][3]             */
    const int32_t* full_faces;    const float* full_bary;    /* [n_full], [n_full][3]                         */
    const int32_t* mp_faces;      const float* mp_bary;      /* [n_mp], [n_mp][3]                             */
} SmirkFlameModel;

/* bytes of scratch smirk_flame_forward needs for batch B (coef[B][KP] + joint transforms[B][60] + lut[B]) */
size_t smirk_flame_workspace_bytes(const SmirkFlameModel* m /*host struct*/, int B);

/* Inputs [B,*]; `neck`, `eye`, `eyelid` may be NULL (=> zeros / no eyelid term, FLAME.py:267-271,284).
 * ns_in / ne_in: number of shape / expression coefficients actually supplied (right-padded with zeros, FLAME.py:244-248).
 * Outputs: verts[B][V][3], lmk_fan[B][n_dyn+n_static][3], lmk_fan3d[B][n_full][3], lmk_mp[B][n_mp][3];
 * lut_idx_out (nullable) [B] int32 = the dynamic-contour LUT row chosen per face (parity tests; needed by the backward pass).
 * v_posed_out (nullable) [B][V][3] = vertices before skinning (lbs.py:202), saved for smirk_flame_backward.
 * `m` is a HOST struct whose members are device pointers. */
int smirk_flame_forward(const SmirkFlameModel* m, int B,
                        const float* shape, int ns_in, const float* exp, int ne_in,
                        const float* global_pose /*[B,3]*/, const float* neck /*[B,3]*/, const float* jaw /*[B,3]*/,
                        const float* eye /*[B,6]*/, const float* eyelid /*[B,2]*/,
                        float* verts, float* lmk_fan, float* lmk_fan3d, float* lmk_mp, int32_t* lut_idx_out,
                        float* v_posed_out, void* ws, size_t ws_bytes, void* stream);

/* Backward of smirk_flame_forward — what autograd computes through FLAME.forward in the reference's training step
 * (smirk_trainer.py:94-104 flame.forward on encoder outputs that require grad; lbs.py:139-225 is differentiated end to end).
 * Given dL/dvertices and dL/dlandmarks_* (each nullable = zero), writes dL/dshape[B][ns_in], dL/dexp[B][ne_in], dL/dglobal_pose[B][3],
 * dL/djaw[B][3], and when the pointers are non-NULL dL/dneck[B][3], dL/deye[B][6], dL/deyelid[B][2] (d_eyelid required iff eyelid given).
 * The dynamic-contour LUT row is piecewise constant in the pose (FLAME.py:137-153 rounds to an integer), so it carries no gradient.
 * dirs_t[KP][3*VP]: the blendshape basis of SmirkFlameModel.dirs transposed to coefficient-major (a constant the host builds once).
 * v_posed / lut_idx: the v_posed_out / lut_idx_out of the forward call on the same inputs. */
size_t smirk_flame_backward_workspace_bytes(const SmirkFlameModel* m /*host struct*/, int B);
int smirk_flame_backward(const SmirkFlameModel* m, const float* dirs_t, int B,
                         const float* shape, int ns_in, const float* exp, int ne_in,
                         const float* global_pose, const float* neck, const float* jaw, const float* eye, const float* eyelid,
                         const float* v_posed, const int32_t* lut_idx,
                         const float* g_verts, const float* g_lmk_fan, const float* g_lmk_fan3d, const float* g_lmk_mp,
                         float* d_shape, float* d_exp, float* d_global_pose, float* d_neck, float* d_jaw, float* d_eye, float* d_eyelid,
                         void* ws, size_t ws_bytes, void* stream);

/* Barycentric landmark gather on its own — replaces vertices2landmarks (lbs.py:101-137) as used by
 * utils/masking.py:170 with per-face index/bary tensors.  faces_idx[B][L] int32, bary[B][L][3] -> out[B][L][3]. */
int smirk_vertices2landmarks(const float* verts, int B, int V, const int32_t* faces /*[F][3]*/,
                             const int32_t* faces_idx, const float* bary, int L, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Renderer — replaces Renderer.forward/render/rasterize/add_directionlight (src/renderer/renderer.py:100-207,239-250),
 *            vertex_normals/face_vertices/batch_orth_proj (src/renderer/util.py:10-78) and the pytorch3d call
 *            rasterize_meshes(image_size=224, blur_radius=0, faces_per_pixel=1) (renderer.py:185-193).
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct SmirkRenderMesh {
    int32_t V;        /* vertices of the full mesh (5023)                                                       */
    int32_t Vf;       /* kept (face-region) vertices, 1787             (renderer.py:68-73)                      */
    int32_t Ff;       /* faces of the kept sub-mesh, 3408              (renderer.py:11-47)                      */
    int32_t nnz;      /* 3*Ff corner contributions of the normal CSR                                            */
    const int32_t* keep;      /* [Vf]   full-mesh vertex id of each kept vertex (sorted)                        */
    const int32_t* faces;     /* [Ff][3] sub-mesh faces, indices into the kept vertices                         */
    const int32_t* nrm_ptr;   /* [Vf+1] CSR row pointers: corner contributions per vertex, in the reference's
                                 index_add_ order (util.py:52-57: corner1 pass, corner2 pass, corner0 pass)      */
    const int32_t* nrm_face;  /* [nnz]  face id of each contribution                                            */
    const int32_t* nrm_corner;/* [nnz]  which corner (0,1,2) of that face the vertex is                         */
} SmirkRenderMesh;

size_t smirk_render_workspace_bytes(const SmirkRenderMesh* mesh /*host struct*/, int B, int H, int W);

/* verts[B][V][3], cam[B][3]=(s,tx,ty) -> transformed[B][V][3] (renderer.py:101-102), img[B][3][H][W] (NCHW, background
 * exactly 0.0).  Optional outputs (nullable): pix_to_face[B][H][W] int64 packed index b*Ff+f or -1 (renderer.py:185),
 * bary[B][H][W][3] (-1 background), zbuf[B][H][W] (-1 background), normals[B][Vf][3].  H, W <= 1024; H == W. */
int smirk_render_forward(const SmirkRenderMesh* mesh, int B, int H, int W,
                         const float* verts, const float* cam,
                         float* transformed, float* img,
                         int64_t* pix_to_face, float* bary, float* zbuf, float* normals,
                         void* ws, size_t ws_bytes, void* stream);

/* util.vertex_normals (util.py:30-62) on a FULL mesh (mesh->Vf == mesh->V, mesh->faces index the vertices directly, mesh->keep unused):
 * verts[B][V][3] -> normals[B][V][3], CSR accumulation order as in smirk_render_forward.  Used by utils/masking.py:146. */
int smirk_vertex_normals(const SmirkRenderMesh* mesh, int B, const float* verts, float* normals, void* stream);

/* Landmark projection (renderer.py:104-108): lmk[B][L][3], cam[B][3] -> out[B][L][2]. */
int smirk_project_landmarks(const float* lmk, const float* cam, int B, int L, float* out, void* stream);

/* Backward of smirk_render_forward — replaces autograd through Renderer.forward (renderer.py:100-207,239-250; util.py:30-78) plus
 * pytorch3d's RasterizeMeshes backward for blur_radius=0, faces_per_pixel=1, perspective_correct=False (gradient of the barycentric
 * weights only; pix_to_face / zbuf / dists carry none in the reference).  pix_to_face: the forward call's optional output.
 * g_img[B][3][H][W] (nullable) = dL/drendered_img, g_transformed[B][V][3] (nullable) = dL/dtransformed_vertices.
 * Writes d_verts[B][V][3] (zero outside the face region unless g_transformed is given) and d_cam[B][3] (overwritten).
 * Deterministic: no atomics, fixed summation order. */
size_t smirk_render_backward_workspace_bytes(const SmirkRenderMesh* mesh /*host struct*/, int B, int H, int W);
int smirk_render_backward(const SmirkRenderMesh* mesh, int B, int H, int W,
                          const float* verts, const float* cam, const int64_t* pix_to_face,
                          const float* g_img, const float* g_transformed,
                          float* d_verts, float* d_cam, void* ws, size_t ws_bytes, void* stream);

/* Backward of smirk_project_landmarks: g_out[B][L][2] -> d_lmk[B][L][3] (written), d_cam_accum[B][3] (ADDED to: call after
 * smirk_render_backward, or on a zeroed buffer). */
int smirk_project_landmarks_backward(const float* lmk, const float* cam, const float* g_out, int B, int L,
                                     float* d_lmk, float* d_cam_accum, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Convolution building blocks (fp32 MFMA implicit GEMM, NHWC) — replace the cuDNN/ATen kernels the reference reaches
 * through nn.Conv2d / nn.ConvTranspose2d / nn.BatchNorm2d(eval) / ReLU / MaxPool2d / ReflectionPad2d
 * (src/smirk_generator.py:13-44,88-178) and timm's conv/bn/act stacks (src/smirk_encoder.py:7-12).
 * ------------------------------------------------------------------------------------------------------------------ */
enum { SMIRK_PAD_ZERO = 0, SMIRK_PAD_REFLECT = 1 };
enum { SMIRK_ACT_NONE = 0, SMIRK_ACT_RELU = 1 };
enum { SMIRK_OUT_NHWC = 0, SMIRK_OUT_CONVT2X2 = 1 };

typedef struct SmirkConvDesc {
    int32_t B, H, W;        /* input spatial size                                                               */
    int32_t C0, C1;         /* channels of source 0 and (optional, concat along C) source 1; both % 4 == 0        */
    int32_t Cout;           /* output channels (for CONVT2X2: channels per output pixel)                        */
    int32_t KH, KW;         /* 3x3 or 1x1                                                                       */
    int32_t stride;         /* 1 (2 is accepted for 1x1 / 3x3 with the pads below)                              */
    int32_t pad_t, pad_l;   /* leading pads; trailing pads follow from Ho, Wo                                    */
    int32_t Ho, Wo;         /* output spatial size (CONVT2X2: the GEMM grid = input grid, output is 2H x 2W)    */
    int32_t pad_mode;       /* SMIRK_PAD_*                                                                      */
    int32_t act;            /* SMIRK_ACT_*  (applied after scale/shift and residual add)                         */
    int32_t out_mode;       /* SMIRK_OUT_*                                                                      */
} SmirkConvDesc;

/* out = act( (conv(cat(in0,in1), w)) * scale[n] + shift[n] + residual ).
 * w: [N][K] with K = KH*KW*(C0+C1) ordered (ky,kx,c) and N = Cout (CONVT2X2: N = 4*Cout ordered (dy,dx,co), K = C0);
 * scale/shift: [Cout] (nullable => 1 / 0); residual (nullable): NHWC like out.  in1 may be NULL when C1 == 0. */
int smirk_conv_igemm_f32(const SmirkConvDesc* d, const float* in0, const float* in1, const float* w,
                         const float* scale, const float* shift, const float* residual, float* out, void* stream);

/* Split-fp16 arithmetic mode ("f16x3"): CDNA4 has no TF32, so fp32-class accuracy on the 16-bit matrix pipe is obtained by carrying each
 * value as an fp16 pair x = hi + lo * 2^-11 and issuing 3 fp16 MFMAs per product block with fp32 accumulation (conv.hip header).
 * A "split16" tensor is NHWC with each group of 8 channels stored as 8 hi halves followed by 8 lo halves: [B][H][W][C/8][2][8] fp16 —
 * 4 bytes per element like fp32.  Same descriptor and semantics as smirk_conv_igemm_f32; in0 / in1 / residual / out are split16 tensors,
 * w is [N][K/8][2][8] halves (K ordered (ky,kx,c)), scale / shift stay fp32.  C0, C1, Cout must be multiples of 8. */
int smirk_conv_igemm_f16x3(const SmirkConvDesc* d, const void* in0, const void* in1, const void* w,
                           const float* scale, const float* shift, const void* residual, void* out, void* stream);
/* The 16-bit class BASELINE config 5 trains in (the reference: torch.autocast(bfloat16) around smirk_trainer.py:184-332): same operands, descriptor and
 * output format, ONE fp16 MFMA per product block (the hi halves only, fp32 accumulation).  Used by the train-mode autograd functions when a module's
 * `train_arith` is "f16x1"; inference never calls it. */
int smirk_conv_igemm_f16x1(const SmirkConvDesc* d, const void* in0, const void* in1, const void* w,
                           const float* scale, const float* shift, const void* residual, void* out, void* stream);
/* The generator's tail in ONE launch (smirk_generator.py:104-113 dec1conv2+norm2+relu2, :47-49,76 conv + sigmoid): 3x3 conv (zero pad 1, stride 1,
 * split16 input, Cout = 32) + scale/shift + ReLU + 1x1 conv fw[fcout][32] + fb + sigmoid -> out_nchw[B][fcout][H][W] fp32; the 32-channel
 * activation never reaches HBM.  Returns SMIRK_ERR_UNSUPPORTED when the shape is not served by the halo-patch kernel (H, W % 16, H >= 64). */
int smirk_conv3x3_tail_f16x3(const SmirkConvDesc* d, const void* in0, const void* in1, const void* w, const float* scale,
                             const float* shift, const float* fw, const float* fb, float* out_nchw, int fcout, void* stream);
/* 3x3 conv (zero pad 1, stride 1, split16 in / out) + scale/shift + ReLU WITH the 2 x 2 / 2 max-pool of its output written by the same launch
 * (smirk_generator.py:52-59: `enc2 = self.encoder2(...)`, `self.pool2(enc2)`): out [B][H][W][Cout], pooled [B][H/2][W/2][Cout], both split16.  Same
 * descriptor as smirk_conv_igemm_f16x3.  Returns SMIRK_ERR_UNSUPPORTED when no kernel with a fused pool serves the shape (today: conv_ring.hip,
 * Cout = 64, H, W multiples of 16 and >= 64, whole power-of-two 32-channel sources) — the caller then runs smirk_conv_igemm_f16x3 + smirk_maxpool2x2_split16. */
int smirk_conv3x3_pool_f16x3(const SmirkConvDesc* d, const void* in0, const void* in1, const void* w, const float* scale, const float* shift, void* out,
                             void* pooled, void* stream);
/* The generator's FIRST encoder block in ONE launch (smirk_generator.py:52-53 `enc1 = self.encoder1(x)`, `self.pool1(enc1)`; _block :88-119 =
 * Conv2d(3x3, pad 1, bias=False) + BatchNorm2d + ReLU, twice): x [B][H][W][8] split16 (the packed network input, in_channels <= 8) ->
 * conv(8 -> 32) + scale1/shift1 + ReLU -> conv(32 -> 32) + scale2/shift2 + ReLU -> e1 [B][H][W][32] split16 (the skip tensor) and
 * pooled [B][H/2][W/2][32] split16 = MaxPool2d(2, 2)(e1).  The 32-channel intermediate never reaches HBM (halo recompute, enc1_fused.hip).
 * w1: [32][72/8][2][8] halves, w2: [32][288/8][2][8] halves (K ordered (ky,kx,c), as smirk_conv_igemm_f16x3).  H, W multiples of 16.
 * smirk_enc1_fused_supported(cin_pad, features, H, W) -> 1 when smirk_generator_forward takes this path ($SMIRK_DISABLE_ENC1_FUSED=1: never). */
int smirk_enc1_fused_supported(int cin_pad, int features, int H, int W);
int smirk_enc1_fused_split16(const void* x, const void* w1, const float* scale1, const float* shift1, const void* w2, const float* scale2,
                             const float* shift2, void* e1, void* pooled, int B, int H, int W, void* stream);
/* fp32 <-> split16 conversion of n_elems values (n_elems % 8 == 0; groups of 8 consecutive values). */
int smirk_f32_to_split16(const float* in, void* out, size_t n_elems, void* stream);
int smirk_split16_to_f32(const void* in, float* out, size_t n_elems, void* stream);
/* Debugging aid ($SMIRK_F16X3_RANGE_CHECK in the Python host): audits a split16 tensor for values the format cannot carry — counts[0] += number of
 * elements that are non-finite or have |x| >= limit (the fp16 `hi` half saturates at 65504), counts[1] = max(counts[1], bits of the largest finite |x|).
 * counts: two zero-initialised uint32 on the device. */
int smirk_split16_range_check(const void* in, size_t n_elems, float limit, uint32_t* counts, void* stream);
/* split16 versions of the generator's streaming ops (smirk_generator.py:13-19,47-49,76; smirk_trainer.py:94): max-pool, fused
 * cat(a[B,Ca,H,W], b[B,Cb,H,W]) + NCHW->NHWC + split (Ca+Cb <= 8, zero-padded to one 8-channel group), final 1x1 conv + sigmoid. */
int smirk_maxpool2x2_split16(const void* in, void* out, int B, int H, int W, int C, void* stream);
int smirk_pack_generator_input_split16(const float* a, int Ca, const float* b, int Cb, void* out, int B, int H, int W, void* stream);
int smirk_conv1x1_sigmoid_nchw_split16(const void* in, const float* w /*[Cout][C] fp32*/, const float* bias, float* out,
                                       int B, int H, int W, int C, int Cout, void* stream);

/* 2x2/2 max pool, NHWC, C % 4 == 0 (smirk_generator.py:13-19). */
int smirk_maxpool2x2_nhwc(const float* in, float* out, int B, int H, int W, int C, void* stream);
/* NCHW [B,Cin,H,W] -> NHWC [B,H,W,Cpad] with zero-filled channels Cin..Cpad-1 (generator input pack). */
int smirk_nchw_to_nhwc_pad(const float* in, float* out, int B, int Cin, int H, int W, int Cpad, void* stream);
/* Same, but the 6 input channels come from two NCHW 3-channel tensors (rendered, masked) => cat fused (smirk_trainer.py:94). */
int smirk_pack_generator_input(const float* rendered, const float* masked, float* out, int B, int H, int W, void* stream);
/* final 1x1 conv (C -> Cout<=4) + bias + sigmoid, NHWC in -> NCHW out (smirk_generator.py:47-49,76). */
int smirk_conv1x1_sigmoid_nchw(const float* in, const float* w /*[Cout][C]*/, const float* bias, float* out,
                               int B, int H, int W, int C, int Cout, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Encoder-specific ops (timm tf_mobilenetv3_*_minimal_100 stacks; src/smirk_encoder.py:34-45,66-73,95-110).
 * ------------------------------------------------------------------------------------------------------------------ */
/* stem: 3x3 stride-2 TF-'same' conv 3->Cout(16) on NCHW input + BN(scale,shift) + ReLU -> NHWC [B,H/2,W/2,Cout]. */
int smirk_stem_conv_s2(const float* img_nchw, const float* w /*[Cout][27] (ky,kx,c)*/, const float* scale,
                       const float* shift, float* out, int B, int H, int W, int Cout, void* stream);
/* depthwise 3x3, stride 1 (pad 1) or 2 (TF-'same': pad 0 top/left, 1 bottom/right for even H), + scale/shift + ReLU. NHWC. */
int smirk_dwconv3x3(const float* in, const float* w /*[9][C]*/, const float* scale, const float* shift, float* out,
                    int B, int H, int W, int C, int stride, int relu, void* stream);
/* global average pool over HW then Linear: feat[B][HW][C] -> out[B][N] = mean_hw(feat) . Wt[N][C] + bias; ws = [B][C] floats of scratch. */
int smirk_gap_linear(const float* feat, const float* w, const float* bias, float* out, float* ws, int B, int HW, int C, int N,
                     void* stream);
/* split16 variants of the three encoder ops (activations in the split-fp16 format of smirk_conv_igemm_f16x3; weights / scale / shift fp32;
 * C and Cout multiples of 8): the stem writes split16, the depthwise stencil reads and writes it, the pooled head reads it. */
int smirk_stem_conv_s2_split16(const float* img_nchw, const float* w, const float* scale, const float* shift, void* out, int B, int H,
                               int W, int Cout, void* stream);
int smirk_dwconv3x3_split16(const void* in, const float* w /*[9][C]*/, const float* scale, const float* shift, void* out, int B, int H,
                            int W, int C, int stride, int relu, void* stream);
int smirk_gap_linear_split16(const void* feat, const float* w, const float* bias, float* out, float* ws, int B, int HW, int C, int N,
                             void* stream);
/* ExpressionEncoder output clamps (smirk_encoder.py:104-108), in place on params[B][n_exp+5]:
 * [n_exp, n_exp+2) clamp(0,1); [n_exp+2] relu; [n_exp+3, n_exp+5) clamp(-.2,.2). */
int smirk_expression_clamps(float* params, int B, int n_exp, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Masking utilities between Renderer and SmirkGenerator (src/utils/masking.py).
 * Random draws come from a counter-based Philox4x32-10 stream keyed by (seed, offset): reproducible, but not torch's streams.
 * ------------------------------------------------------------------------------------------------------------------ */
/* sampling weight per (image, triangle): (mean vertex-normal z < 0.05 ? face_prob : 0) * xy shoelace area   (masking.py:144-160).
 * tverts = Renderer's transformed_vertices [B][V][3]; normals = smirk_vertex_normals of them. */
int smirk_mask_face_weights(const float* tverts, const float* normals, const int32_t* faces /*[F][3]*/, const float* face_prob /*[F]*/,
                            int B, int V, int F, float* weights /*[B][F]*/, void* stream);
/* multinomial(weights, num, replacement=True) + random_barycentric (masking.py:51-68,163-166): idx[B][num] int32, bary[B][num][3]. */
int smirk_sample_faces(const float* weights, int B, int F, int num, uint64_t seed, uint64_t offset, int32_t* idx, float* bary,
                       void* stream);
/* npoints = (.5 * (1 + p) * S).long(), x and y clamped to [0, S-1] (masking.py:172-175): points[B][L][3] -> out[B][L][3] int64. */
int smirk_points_to_pixels(const float* points, int B, int L, int image_size, int64_t* out, void* stream);
/* out = max_pool2d(in, 2r+1, stride 1, padding r) on [B][1][H][W] as two separable passes (tmp = scratch of the same size).
 * complement bit 0: pool (1 - in) instead of in; bit 1: return 1 - result.  3 => 1 - maxpool(1 - in) (masking.py:78), 2 => 1 - maxpool(in) (:96).  `tmp` must alias neither `in` nor `out` (SMIRK_ERR_BAD_ARG); in == out is allowed (it takes the two-pass kernels through tmp). */
int smirk_maxpool_sq(const float* in, float* tmp, float* out, int B, int H, int W, int radius, int complement, void* stream);
int smirk_bernoulli_field(float* out, size_t n, float p, uint64_t seed, uint64_t offset, void* stream);
/* masking.py:84-101: out = extra' > 0 ? extra' : img * mask * (1 - rendered_mask), extra' = extra_points * noise * keep.
 * noise_mult (nullable, [B][C][H][W]) overrides the generated N(1, 0.05) field; gen_noise = 0 and noise_mult = NULL => no noise. */
int smirk_masking_compose(const float* img, const float* mask, const float* rendered_mask, const float* extra_points /*nullable*/,
                          const float* pmask /*nullable [B][1][H][W]: extra_points = img * pmask (demo.py:163) when extra_points is NULL*/,
                          const float* keep, const float* noise_mult, int B, int C, int H, int W, int gen_noise,
                          uint64_t seed, uint64_t offset, float* out, void* stream);
/* caller glue of demo.py:146,153-159 / smirk_trainer.py:79,86-88: rendered_mask = 1 - (rendered_img == 0).all(dim=1) -> [B][1][H][W];
 * pmask = 1 at the first rbound[b] (nullable: all) sampled pixel positions points[B][L][3] int64 (x, y, .) -> [B][1][H][W]. */
int smirk_rendered_mask(const float* rendered_img, int B, int C, int H, int W, float* out, void* stream);
int smirk_scatter_points_mask(const int64_t* points, const int64_t* rbound, int B, int L, int H, int W, float* out, void* stream);
/* transfer_pixels (masking.py:116-129): out[b,:,p2y,p2x] = img[b,:,p1y,p1x] for l < rbound[b] (rbound nullable), 0 elsewhere;
 * duplicate targets: the highest l wins.  points [B][L][3] int64 (x, y, .), winner_ws [B][H][W] int32 scratch. */
int smirk_transfer_pixels(const float* img, const int64_t* points1, const int64_t* points2, const int64_t* rbound, int B, int C,
                          int L, int H, int W, int32_t* winner_ws, float* out, void* stream);

/* One fused launch for a timm MobileNetV3 "minimal" block (smirk_encoder.py:11-21 runs them through
 * `self.encoder(img)`): InvertedResidual = conv_pw 1x1 + bn1 + ReLU -> conv_dw 3x3 (stride 1|2, TF 'SAME') + bn2 + ReLU -> conv_pwl 1x1 + bn3
 * (+ x when stride 1 and Cin == Cout); DepthwiseSeparable (wexp == NULL, mid == Cin) = conv_dw + bn1 + ReLU -> conv_pw + bn2 (+ x).
 * Activations are split16 NHWC; wexp [mid][Cin] and wproj [Cout][mid] split16 rows; wdw [9][mid] fp32; s1..s3 / b1..b3 = folded eval-mode BatchNorm
 * scale / shift.  The expanded tensors never leave the CU's LDS.  All channel counts must be multiples of 8. */
size_t smirk_mbconv_lds_bytes(int Cin, int mid, int Cout, int stride);
/* 1 if the fused kernel serves this block shape (Cin <= 48, Cout <= 96, LDS <= 64 KiB); otherwise use the per-layer kernels */
/* InvertedResidual block with WHOLE images per workgroup (stride 1, H * W <= 224 — the 14 x 14 / 7 x 7 stages, Cin a multiple of 16 up to 112, Cout <= 128:
 * timm InvertedResidual; reference call site src/smirk_encoder.py:18-21,52-55,80-83): same arguments and arithmetic as
 * smirk_mbconv_fused_split16, no halo recompute; x stays in LDS, the expanded / depthwise tensors never exist in memory.  _supported: 1 when served. */
int smirk_mbconv_image_supported(int H, int W, int Cin, int mid, int Cout, int stride);
int smirk_mbconv_image_split16(const void* x, const void* wexp, const float* s1, const float* b1, const float* wdw, const float* s2, const float* b2,
                               const void* wproj, const float* s3, const float* b3, int residual, void* out, int B, int H, int W, int Cin, int mid,
                               int Cout, void* stream);
/* Stem + first DepthwiseSeparable block of a MobileNetV3-minimal backbone in ONE launch (timm `conv_stem` + `bn1` + `blocks[0][0]`; reference call site
 * src/smirk_encoder.py:18-21,52-55,80-83 `self.encoder(img)[-1]`): img[B][3][H][W] NCHW fp32 -> 3x3 s2 conv 3->16 + BN + ReLU -> 3x3 depthwise (stride 1 | 2,
 * TF-SAME) + BN + ReLU -> 1x1 16->16 + BN (+ the stem output when residual) -> out split16 NHWC [B][Ho][Wo][16].  stem_w [16][27] fp32 (k = (ky,kx,c)),
 * dw_w [9][16] fp32, pw_w split16 rows [16][16]; scales / shifts are folded eval-mode BatchNorm.  smirk_encoder_head_supported: 1 when the backbone's
 * stem / first block have that shape (the caller keeps the three-launch sequence otherwise). */
int smirk_encoder_head_supported(int stem_cout, int kind, int cin, int mid, int cout, int stride, int skip);
int smirk_encoder_head_fused_split16(const float* img, const float* stem_w, const float* stem_scale, const float* stem_shift, const float* dw_w,
                                     const float* dw_scale, const float* dw_shift, const void* pw_w, const float* pw_scale, const float* pw_shift,
                                     int residual, void* out, int B, int H, int W, int stride, void* stream);
int smirk_mbconv_supported(int Cin, int mid, int Cout, int stride);
int smirk_mbconv_fused_split16(const void* x, const void* wexp, const float* s1, const float* b1, const float* wdw, const float* s2,
                               const float* b2, const void* wproj, const float* s3, const float* b3, int residual, void* out,
                               int B, int H, int W, int Cin, int mid, int Cout, int stride, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Whole-network entries (SURVEY.md §8(b) proposal): ONE call enqueues every layer of a module's forward on `stream`, so a host in any
 * language drives the path with a handful of calls instead of re-implementing the ~100-launch layer schedules.  All weights are passed
 * in the PACKED form the per-layer entries above take (the Python host packs them once per parameter version); activations live in the
 * caller-provided workspace.  Same conventions as everything else: no allocation, no synchronisation, 0 / negative error code.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct SmirkConvLayer {
    const void* w;       /* packed weight: [N][K] (K = (ky,kx,c)), fp32 or split16 rows as the precision says; depthwise: [9][C] fp32 */
    const float* scale;  /* [Cout] folded eval-mode BatchNorm scale (NULL = 1)                                                        */
    const float* shift;  /* [Cout] folded BatchNorm shift, or the conv bias (ConvTranspose2d)                                          */
} SmirkConvLayer;

#define SMIRK_GEN_MAX_RES 16
enum { SMIRK_PRECISION_F32 = 0, SMIRK_PRECISION_F16X3 = 1 };

/* SmirkGenerator(in_channels, out_channels, init_features, res_blocks) — src/smirk_generator.py:8-49 (parameter inventory), :51-86 (forward). */
typedef struct SmirkGeneratorWeights {
    int32_t in_channels, out_channels, features, res_blocks;
    int32_t precision;   /* SMIRK_PRECISION_*                                                                                  */
    int32_t cin_pad;     /* channels of the packed network input (8 in f16x3 mode; in_channels rounded up to 4 in f32 mode)    */
    SmirkConvLayer enc[5][2];                  /* encoder1..4, bottleneck: conv1+norm1, conv2+norm2        (:13-27, _block :88-119) */
    SmirkConvLayer res[SMIRK_GEN_MAX_RES][2];  /* resnet_blocks[k].conv_block.{1,2} and .{5,6}            (:29-32, :121-178)       */
    SmirkConvLayer up[4];                      /* upconv4, 3, 2, 1: w = [(dy,dx,co)][ci], shift = bias     (:34-45)                 */
    SmirkConvLayer dec[4][2];                  /* decoder4, 3, 2, 1                                                                 */
    const float* final_w;                      /* conv.weight [out_channels][features] fp32                (:47-49)                 */
    const float* final_b;                      /* conv.bias [out_channels]                                                          */
} SmirkGeneratorWeights;

size_t smirk_generator_workspace_bytes(const SmirkGeneratorWeights* w /*host struct*/, int B, int H, int W);
/* y[B][out_channels][H][W] = SmirkGenerator.forward(cat(a, b)) — a[B][Ca][H][W] and b[B][Cb][H][W] are NCHW fp32, Ca + Cb == in_channels
 * (b may be NULL with Cb == 0: the caller already concatenated, smirk_trainer.py:94 / demo.py:167 `torch.cat([rendered_img, masked_img], 1)`).
 * H, W multiples of 16 (4 pooling levels).  taps (nullable): 10 device pointers that receive the intermediate activations
 * enc1..enc4, bottleneck, res, dec4..dec1 in the precision's storage format (parity tests); passing taps disables the fused tail. */
int smirk_generator_forward(const SmirkGeneratorWeights* w, const float* a, int Ca, const float* b, int Cb, float* y,
                            int B, int H, int W, void* const* taps, void* ws, size_t ws_bytes, void* stream);

/* One timm MobileNetV3-"minimal" block.  kind 0 = DepthwiseSeparable (dw -> pw), 1 = InvertedResidual (pw -> dw -> pwl),
 * 2 = ConvBnAct 1x1 (pw only). */
typedef struct SmirkMbBlock {
    int32_t kind, stride, cin, mid, cout, skip;
    SmirkConvLayer pw, dw, pwl;
} SmirkMbBlock;

#define SMIRK_BACKBONE_MAX_BLOCKS 24
/* One sub-encoder of SmirkEncoder: backbone (features[-1]) -> global average pool -> Linear (+ the ExpressionEncoder clamps)
 * — src/smirk_encoder.py:14-45 (PoseEncoder), :48-73 (ShapeEncoder), :76-110 (ExpressionEncoder). */
typedef struct SmirkBackboneWeights {
    int32_t n_blocks, precision, n_out;
    int32_t clamp_n_exp;        /* >= 0: apply smirk_expression_clamps with this n_exp to the head output (smirk_encoder.py:104-108); -1: none */
    int32_t stem_cout, feat_ch; /* 16; channels of the last feature map (576 / 960)                                                  */
    SmirkConvLayer stem;        /* conv_stem [Cout][27] fp32 + bn1                                                                    */
    SmirkMbBlock blocks[SMIRK_BACKBONE_MAX_BLOCKS];
    const float* head_w;        /* Linear weight [n_out][feat_ch]                                                                     */
    const float* head_b;        /* [n_out]                                                                                            */
} SmirkBackboneWeights;

size_t smirk_backbone_workspace_bytes(const SmirkBackboneWeights* w /*host struct*/, int B, int H, int W);
/* img[B][3][H][W] NCHW in [0,1] -> out[B][n_out] fp32 (n_out == 0: no head, out may be NULL and feat_out is required).
 * feat_out (nullable): the last feature map, NHWC [B][ceil(H/32)][ceil(W/32)][feat_ch], in the precision's storage format (parity tests).  The three sub-encoders of SmirkEncoder.forward (smirk_encoder.py:123-133) are independent: call this entry once
 * per sub-encoder, each on its own stream and workspace, to let their small launches interleave. */
int smirk_backbone_forward(const SmirkBackboneWeights* w, const float* img, int B, int H, int W, float* out, void* feat_out,
                           void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Training-mode building blocks (BASELINE config 5, generator slice) — what autograd + nn.BatchNorm2d.train() compute inside the reference's
 * SmirkGenerator after `trainer.train()` (base_trainer.py:108-111; smirk_generator.py:88-119,121-178; smirk_trainer.py:184-332,365-382).
 * Activations and activation gradients are split16 NHWC ([M = B*H*W][C]); statistics, affine parameters and weight gradients are fp32.
 * Reductions are two-stage fp64 in a fixed order (bit-reproducible).  Data gradients of convolutions reuse smirk_conv_igemm_f16x3 with the
 * weights rotated by 180 degrees and Cin <-> Cout swapped (host-side repack), so they have no entry of their own.
 * ------------------------------------------------------------------------------------------------------------------ */
size_t smirk_train_reduce_workspace_bytes(int C);
/* y = [relu]((z - mean_batch) / sqrt(var_batch + eps) * gamma + beta [+ residual]);  saves mean / biased var / invstd for the backward pass and updates
 * running_mean / running_var (nullable) with `momentum` (running_var takes the unbiased variance), exactly like F.batch_norm(training=True);
 * num_batches_tracked (nullable, device int64 scalar) is incremented by the same launch (nn.BatchNorm2d.forward's `num_batches_tracked.add_(1)`). */
int smirk_bn_train_forward_split16(const void* z, size_t M, int C, const float* gamma, const float* beta, const void* residual, int relu, float eps,
                                   float momentum, float* running_mean, float* running_var, long long* num_batches_tracked, float* save_mean,
                                   float* save_var, float* save_invstd, void* y, void* ws, size_t ws_bytes, void* stream);
/* BatchNorm statistics from the producing convolution's accumulators (train mode; smirk_generator.py:88-119 `_block`: Conv2d -> BatchNorm2d -> ReLU, and the timm
 * blocks of smirk_encoder.py:7-12 — in the reference cuDNN's convolution and ATen's batch_norm are separate passes over the activation).
 * smirk_conv_igemm_stats_split16: the raw convolution of smirk_conv_igemm_f16x3 / _f16x1 (no scale / shift / residual / activation) that also writes per-tile partial
 * column sums (sum z, sum z^2) of its output: stats[rows][Cout][2] fp32, buffer of smirk_conv_stats_rows_max(d) rows; *rows (host int) = rows written, or 0 when the
 * kernel that serves this shape leaves no statistics (the caller then uses smirk_bn_train_forward_split16, which reduces the stored tensor).
 * smirk_bn_train_forward_partials_split16: smirk_bn_train_forward_split16 with the statistics taken from those P partial rows (fixed-order fp64 reduction);
 * partials_fp64 = 0: the convolutions' fp32 rows, 1: fp64 stage-1 rows of a reduction kernel. */
size_t smirk_conv_stats_rows_max(const SmirkConvDesc* d);
int smirk_conv_igemm_stats_split16(const SmirkConvDesc* d, const void* in0, const void* in1, const void* w, void* out, float* stats, int* rows, int x1, void* stream);
int smirk_bn_train_forward_partials_split16(const void* z, size_t M, int C, const float* gamma, const float* beta, const void* residual, int relu, float eps,
                                            float momentum, float* running_mean, float* running_var, long long* num_batches_tracked, float* save_mean,
                                            float* save_var, float* save_invstd, void* y, const void* partials, int P, int partials_fp64, void* stream);
/* the depthwise convolutions of the backbones' blocks in train mode: raw output + the stage-1 partial sums of its statistics in ONE launch (fp64 rows [rows][C][2],
 * buffer of 1024 rows; consumed by the entry above with partials_fp64 = 1; SMIRK_ERR_UNSUPPORTED for inputs of 2 GiB and more) */
int smirk_dwconv3x3_stats_split16(const void* in, const float* w /*[9][C]*/, void* out, int B, int H, int W, int C, int stride, double* part, int* rows, void* stream);
/* dy = dL/dy of the forward above (relu: the mask is recomputed from z); writes dz, dgamma[C], dbeta[C].  The residual's gradient is dy itself. */
int smirk_bn_train_backward_split16(const void* z, const void* dy, size_t M, int C, const float* gamma, const float* beta, const float* save_mean,
                                    const float* save_invstd, int relu, void* dz, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, void* stream);
/* Eval-mode BatchNorm with a backward pass — the frozen, .eval() generator that smirk_trainer.py:108-113 back-propagates THROUGH (emotion loss):
 * forward y = [relu]((z - running_mean) / sqrt(running_var + eps) * gamma + beta [+ residual]) and writes invstd[C] plus a zero vector zeros[C] for the
 * backward; backward dz = gamma * invstd * dy * [relu mask recomputed from z] (no batch statistics: running estimates are constants). */
int smirk_bn_eval_forward_split16(const void* z, size_t M, int C, const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                                  const void* residual, int relu, float eps, float* save_invstd, float* zeros, void* y, void* stream);
int smirk_bn_eval_backward_split16(const void* z, const void* dy, size_t M, int C, const float* gamma, const float* beta, const float* running_mean,
                                   const float* invstd, const float* zeros, int relu, void* dz, void* stream);
/* sums[c] = sum over the M rows of x[.][c]  (bias gradients) */
int smirk_colsum_split16(const void* x, size_t M, int C, float* sums, void* ws, size_t ws_bytes, void* stream);
/* nn.MaxPool2d(2, 2) backward: dx[b,2y+i,2x+j,c] = dy[b,y,x,c] at the first maximum of the window (ATen scan order), 0 elsewhere, + add (nullable:
 * a second gradient of the same tensor, e.g. the U-Net skip connection's). */
int smirk_maxpool2x2_backward_split16(const void* x, const void* dy, const void* add, void* dx, int B, int H, int W, int C, void* stream);
/* nn.ReflectionPad2d(1) backward: dxp[B][H+2][W+2][C] -> dx[B][H][W][C] (+ add, nullable) */
int smirk_reflect_pad1_backward_split16(const void* dxp, const void* add, void* dx, int B, int H, int W, int C, void* stream);
/* [B][2H][2W][C] -> [B][H][W][(dy,dx,c)]: the ConvTranspose2d(k=2,s=2) output gradient as the operand of a 1x1 convolution / weight gradient */
int smirk_space_to_depth2_split16(const void* in, void* out, int B, int H, int W, int C, void* stream);
/* final Conv2d(C, Cout<=4, 1) + sigmoid backward (smirk_generator.py:47-49,76): dy, y NCHW fp32; w [Cout][C] fp32 -> dd[B][H][W][C] split16 (gradient of the
 * conv input) and dl8[B][H][W][8] split16 (dL/dlogits, channels >= Cout zero) from which smirk_conv_wgrad_f32 / smirk_colsum_split16 give dW and db. */
int smirk_conv1x1_sigmoid_backward_split16(const float* dy, const float* y, const float* w, void* dd, void* dl8, int B, int H, int W, int C, int Cout,
                                           void* stream);
/* Weight gradient of a KHxKH (1 or 3), stride-1, pad-(KH-1)/2 (zero or reflect) convolution (autograd's grad_weight of the nn.Conv2d layers in
 * smirk_generator.py:88-178 / the timm pointwise convolutions), fp32 result:
 * dw[Cout][(ky,kx,ci)] = sum_p dz[p][co] * x[p + (ky,kx) - pad][ci]   — the packed forward weight layout.  ConvTranspose2d(k=2,s=2): call with KH = 1,
 * dz = the layer INPUT [M][Cin_t] and x = space_to_depth2(output gradient) [M][4*Cout_t] => dw[Cin_t][(dy,dx,co)].
 * Arithmetic: split-fp16 x3 on the fp16 matrix pipe (operands are consumed as stored, transposed by ds_read_b64_tr_b16; fp32 accumulate, fp32-class error
 * like the forward convolutions) by default, or the exact-fp32 MFMA kernels (v_mfma_f32_32x32x2_f32).  smirk_conv_wgrad_set_mode: 0 = exact fp32,
 * 1 / 2 = split-fp16 x3 with 1 / 2 pixel chunks per barrier, -1 = back to $SMIRK_WGRAD_F16 / the built-in default; returns the previous mode. */
int smirk_conv_wgrad_set_mode(int mode);
/* nn.Conv2d weight [Cout][cin_total][KH][KH] fp32 (KH 1 or 3), input-channel slice [cin_off, cin_off + Cin) (a U-Net decoder conv reads two sources)
 * -> the step's two split16 operand images in ONE launch (either may be NULL):
 * fwd [Cout][(ky,kx,c)], c < cin_pad (zeros beyond Cin) and dgrad [cin_pad][(ky,kx,co)] = W rotated by 180 degrees with Cin <-> Cout swapped. */
int smirk_pack_conv_weights_split16(const float* w, int Cout, int cin_total, int cin_off, int Cin, int KH, int cin_pad, void* fwd, void* dgrad, void* stream);
/* The same for EVERY weight of a network in one launch: `jobs_device` is a device array of njobs descriptors (arguments of the entry above; fwd / dgrad may be
 * NULL per job), `start` = the job's first index in the flattened list of 8-channel output vectors, total_vectors = the list's length. */
/* job kinds beyond the nn.Conv2d weights (KH = 1 / 3), selected by a negative KH — every weight re-layout of a training step in the ONE launch:
 *   SMIRK_PACK_DEPTHWISE  w = depthwise weight [C][1][3][3] (Cout = C) -> fwd = fp32 [9][C] (smirk_dwconv3x3_split16's image); C * 9 / 8 work items
 *   SMIRK_PACK_STEM       w = stem weight [Cout][3][3][3] -> fwd = fp32 [Cout][(ky,kx,c)] (smirk_stem_conv_s2_raw_split16's image); Cout * 27 work items
 *   SMIRK_PACK_CONVT2X2   w = nn.ConvTranspose2d(Cin_t, Cout_t, 2, 2) weight [Cin_t][Cout_t][2][2] with Cout = Cin_t and Cin = Cout_t in the descriptor ->
 *                         fwd = split16 [(dydx, co)][ci] (the forward 1x1 form of smirk_generator.py:30-44's upconv), dgrad = split16 [ci][(dydx, co)];
 *                         4 * Cin_t * Cout_t / 8 work items each (fwd first; either may be NULL) */
#define SMIRK_PACK_DEPTHWISE (-3)
#define SMIRK_PACK_STEM (-27)
#define SMIRK_PACK_CONVT2X2 (-2)
typedef struct SmirkPackJob {
    const float* w;
    void* fwd;
    void* dgrad;
    int32_t Cout, cin_total, cin_off, Cin, KH, cin_pad;
    unsigned long long start;
} SmirkPackJob;
int smirk_pack_conv_weights_batch_split16(const SmirkPackJob* jobs_device, int njobs, unsigned long long total_vectors, void* stream);
size_t smirk_conv_wgrad_workspace_bytes(int B, int H, int W, int Cout, int Cin, int KH);
int smirk_conv_wgrad_f32(const void* dz, const void* x, float* dw, int B, int H, int W, int Cout, int Cin, int KH, int reflect, void* ws, size_t ws_bytes,
                         void* stream);
/* the same weight gradient in the 16-bit class of smirk_conv_igemm_f16x1 (hi halves of dz and x, one MFMA per block, fp32 accumulation and reduction);
 * SMIRK_ERR_UNSUPPORTED when the fp16 weight-gradient kernels do not serve the call (operands of 2 GiB and more, $SMIRK_WGRAD_F16=0). */
int smirk_conv_wgrad_f16x1(const void* dz, const void* x, float* dw, int B, int H, int W, int Cout, int Cin, int KH, int reflect, void* ws, size_t ws_bytes,
                           void* stream);
/* the same weight gradient summed straight into the layout the caller keeps it in — what autograd hands the optimiser for `conv.weight.grad`
 * (smirk_trainer.py:367-376 backward + Adam; nn.Conv2d / nn.ConvTranspose2d parameter layouts), so no permute-copy / torch.cat follows:
 *   layout 0  packed [Cout][(ky,kx,ci)] (as smirk_conv_wgrad_f32)
 *   layout 1  nn.Conv2d weight [Cout][cin_total][KH][KH], channels [cin_off, cin_off + cin_real) — cin_real <= Cin drops padded operand channels, and the two
 *             sources of a decoder convolution (torch.cat((up, skip), 1), smirk_generator.py:65-75) fill the two channel ranges of ONE tensor in two calls
 *   layout 2  nn.ConvTranspose2d(Cout, Cin / 4, 2, 2) weight [Cout][Cin / 4][2][2] from the 1x1 form of its backward (dz = the layer's input, x = space-to-depth of
 *             the output gradient; KH = 1)
 * x1 != 0 asks for the f16x1 arithmetic where the fp16 kernels serve the call (falls back to the f32-class kernels otherwise, like the Python caller did). */
int smirk_conv_wgrad_param(const void* dz, const void* x, float* dw_param, int B, int H, int W, int Cout, int Cin, int KH, int reflect, int layout, int cin_total,
                           int cin_off, int cin_real, int x1, void* ws, size_t ws_bytes, void* stream);
/* number of smirk_conv_wgrad_param calls since load that asked for x1 and were served by the f32-class kernels instead: lets the caller tell the user that a
 * step declared f16x1 (the reference: bf16 autocast, base_trainer.py / smirk_trainer.py:349-376) mixed arithmetics, instead of doing so silently */
unsigned long long smirk_conv_wgrad_x1_fallbacks(void);

/* ---- train-mode SmirkEncoder backbones (csrc/train_encoder.hip): what `self.train()` + autograd does to the timm
 * tf_mobilenetv3_{small,large}_minimal_100 feature extractors of smirk_encoder.py:7-12 (pointwise convolutions and BatchNorm reuse the entries above).
 * Stem Conv2d(3, Cout, 3, stride 2, TF 'SAME') WITHOUT BatchNorm / ReLU (train mode normalises with batch statistics afterwards): img NCHW fp32,
 * w [Cout][(ky,kx,c)] fp32 -> out split16 [B][ceil(H/2)][ceil(W/2)][Cout] */
int smirk_stem_conv_s2_raw_split16(const float* img, const float* w, void* out, int B, int H, int W, int Cout, void* stream);
/* its weight gradient dw[Cout][(ky,kx,c)] and image gradient dimg NCHW fp32 (the gradient the cycle path returns to the generator, smirk_trainer.py:293-297) */
size_t smirk_stem_conv_s2_wgrad_workspace_bytes(int Cout);
int smirk_stem_conv_s2_wgrad_split16(const float* img, const void* dz, float* dw, int B, int H, int W, int Cout, void* ws, size_t ws_bytes, void* stream);
int smirk_stem_conv_s2_dgrad_split16(const void* dz, const float* w, float* dimg, int B, int H, int W, int Cout, void* stream);
/* depthwise 3x3 (stride 1 pad 1, or stride 2 TF 'SAME'; w [9][C] fp32 as smirk_dwconv3x3_split16 takes it): data gradient dx[B][H][W][C] (+ add, nullable:
 * the skip connection's gradient) from dz[B][ceil(H/s)][ceil(W/s)][C], and weight gradient dw[9][C] */
int smirk_dwconv3x3_dgrad_split16(const void* dz, const float* w, const void* add, void* dx, int B, int H, int W, int C, int stride, void* stream);
size_t smirk_dwconv3x3_wgrad_workspace_bytes(int C);
int smirk_dwconv3x3_wgrad_split16(const void* dz, const void* x, float* dw, int B, int H, int W, int C, int stride, void* ws, size_t ws_bytes, void* stream);
/* the same gradient written in the parameter's own layout [C][1][3][3] (conv_dw.weight of the timm blocks: no transpose-copy between backward and Adam) */
int smirk_dwconv3x3_wgrad_param_split16(const void* dz, const void* x, float* dw, int B, int H, int W, int C, int stride, void* ws, size_t ws_bytes, void* stream);
/* F.adaptive_avg_pool2d(features, 1) + nn.Linear backward (smirk_encoder.py:18-22): dout [B][N], w [N][C], pooled [B][C] (the forward's workspace) ->
 * dw [N][C], db [N] (both or neither), dfeat split16 [B][HW][C] (nullable) */
int smirk_gap_linear_backward_split16(const float* dout, const float* w, const float* pooled, float* dw, float* db, void* dfeat, int B, int HW, int C, int N,
                                      void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Launch profiler (diagnostics; the one place where the library creates HIP events and synchronises — on request only).
 * Between smirk_profile_start() and smirk_profile_stop() every kernel the library launches from the calling process is bracketed
 * by a pair of hipEvents on ITS launch stream.  stop() waits for them and reports, per launch, the kernel's name as instantiated,
 * its algorithmic flop / bytes where the dispatcher knows them (0 otherwise) and its duration.  bench.py builds `roofline` from this.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct SmirkProfileRecord {
    char kernel[120];
    double flop;      /* algorithmic floating-point operations of the launch (2*M*N*K for the GEMM-shaped kernels), 0 if not stated */
    double bytes;     /* algorithmic bytes (unique operands in + out) for the bandwidth-bound kernels, 0 if not stated             */
    float ms;         /* hipEventElapsedTime between the events recorded before and after the launch, on the launch stream          */
    int32_t _pad;
} SmirkProfileRecord;
int smirk_profile_start(void);
/* returns the number of launches recorded (may exceed cap; only the first `cap` are written), or a negative error */
int smirk_profile_stop(SmirkProfileRecord* out, int cap);
/* per-image random point budgets of demo.py:154-156 computed on the device (no host round trip):
 * rbound[b] = (int64)(n_points * (1 / mul) * rscale^rsing), rsing in {-1, +1}, rscale ~ U[1, mul) — Philox keyed by (seed, offset). */
int smirk_random_point_budget(int64_t* rbound, int B, int n_points, float mul, uint64_t seed, uint64_t offset, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Video loop pre/post-processing — replaces the cv2 / skimage calls of demo_video.py:107-214 so that a batch
 * of decoded frames stays in HBM from uint8 in to uint8 out.  uint8 images are HWC (cv2 layout), float images NCHW in [0,1].
 * ------------------------------------------------------------------------------------------------------------------ */
/* skimage.transform.warp(img, inverse_map, output_shape=(Ho,Wo), preserve_range=True).astype(uint8) (demo_video.py:124,163,203): bilinear,
 * constant 0 outside, float64; mats[N][6] = first two rows of the 3x3 matrix taking OUTPUT (col,row,1) to INPUT (x,y).
 * Outputs (either nullable): out_f32_nchw = value/255 with optional R<->B swap (cvtColor + /255 + permute, demo_video.py:133-135),
 * out_u8_hwc. */
int smirk_warp_affine_u8(const uint8_t* src /*[N][Hs][Ws][3]*/, int N, int Hs, int Ws, const double* mats, int Ho, int Wo,
                         int swap_rb, float* out_f32_nchw, uint8_t* out_u8_hwc, void* stream);
/* cv2.resize(img, (Wo, Ho)) INTER_LINEAR, 8-bit fixed-point path (demo_video.py:134); same outputs as above. */
int smirk_resize_linear_u8(const uint8_t* src, int N, int Hs, int Ws, int Ho, int Wo, int swap_rb,
                           float* out_f32_nchw, uint8_t* out_u8_hwc, void* stream);
/* (x*255).astype(uint8) + optional channel swap of a float NCHW panel into columns [col0, col0+W) of a uint8 HWC grid that is
 * grid_w pixels wide = torch.cat(panels, dim=3) ... cvtColor (demo_video.py:170-172,209-214). */
int smirk_f32_nchw_to_u8_grid(const float* src, int N, int H, int W, int swap_rb, uint8_t* grid, int grid_w, int col0, void* stream);
/* torch.Tensor(u8).permute(2,0,1).float()/255 with optional channel swap (demo_video.py:165,169). */
int smirk_u8_hwc_to_f32_nchw(const uint8_t* src, int N, int H, int W, int swap_rb, float* dst, void* stream);
/* F.interpolate(x, (Ho, Wo), mode='bilinear') (align_corners=False; demo_video.py:167,207) on NC planes. */
int smirk_interp_bilinear_f32(const float* src, int NC, int H, int W, int Ho, int Wo, float* dst, void* stream);
/* create_mask (datasets/base_dataset.py:9-15): landmarks[N][L][stride>=2] (x, y, ...) float, truncated to int32; out[N][1][H][W] = 0 inside
 * the closed convex hull, 1 outside.  L <= 1024. */
int smirk_hull_mask(const float* landmarks, int N, int L, int stride, int H, int W, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SMIRK_HIP_H */
