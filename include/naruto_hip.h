/*
 * naruto_hip.h -- C ABI of libnaruto_hip.so: NARUTO's neural-implicit mapping / uncertainty hot path
 * (Co-SLAM-derived joint hash grid + OneBlob + two tiny MLPs, SDF-weighted compositing, uncertainty
 * aggregation, mapping losses) as hand-written HIP kernels for gfx950 (MI355X).
 *
 * The reference has no FFI: its seam is the Python attribute surface of one nn.Module
 * (reference src/slam/coslam/coslam.py:65).  Each entry point below names
 * the reference code it replaces.  Conventions:
 *   - every pointer is a DEVICE pointer to fp32 data unless it says "host";
 *   - nothing here allocates, frees, synchronises or throws; work is enqueued on `stream`
 *     (a hipStream_t passed as void*; NULL = the legacy default stream);
 *   - return value 0 = ok, negative = error (naruto_last_error() gives the text);
 *   - gradients are ACCUMULATED (+=) into the buffers of NarutoGrads, so the caller zeroes them
 *     (this is what torch's .grad accumulation needs, reference coslam.py:368-399);
 *   - a NarutoField handle is immutable after creation and may be shared by streams/threads.
 */
#ifndef NARUTO_HIP_H
#define NARUTO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NARUTO_MAX_LEVELS 16
#define NARUTO_OK 0
#define NARUTO_ERR_INVALID (-22)      /* bad argument / unsupported configuration */
#define NARUTO_ERR_LAUNCH (-5)        /* HIP launch failure */

typedef struct NarutoField NarutoField;

/* Static description of the scene representation (reference: JointEncodingNaruto.__init__,
 * scene_rep.py:26-36; Co-SLAM get_encoder / tcnn HashGrid config; decoder.py:82-97). */
typedef struct NarutoFieldDesc {
    uint32_t n_levels;            /* tcnn n_levels; this build supports 16                      */
    uint32_t n_features;          /* tcnn n_features_per_level; this build supports 2           */
    uint32_t log2_hashmap_size;   /* config grid.hash_size; 4 .. 24                              */
    uint32_t base_resolution;     /* 16                                                         */
    float    per_level_scale;     /* exp2(log2(desired_res / base_res) / (n_levels-1))           */
    uint32_t n_bins;              /* OneBlob bins per input dim; this build supports 16          */
    uint32_t hidden_dim;          /* SDF net hidden width; this build supports 32                */
    uint32_t geo_feat_dim;        /* this build supports 15                                      */
    uint32_t hidden_dim_color;    /* this build supports 32                                      */
    uint32_t uncert_dims[3];      /* uncertainty voxel grid [Nx,Ny,Nz] (scene_rep.py:49-56)      */
    float    bbox_min[3];         /* config mapping.bound[:,0]                                   */
    float    bbox_max[3];         /* config mapping.bound[:,1]                                   */
    float    trunc;               /* training.trunc                                              */
    float    sc_factor;           /* data.sc_factor                                              */
    int32_t  white_bkgd;          /* training.white_bkgd                                         */
    uint32_t mlp_mode;            /* NARUTO_MLP_FP32 (exact fp32 MFMA chain: the parity mode) or
                                     NARUTO_MLP_BF16 (bf16 operands, fp32 accumulate, on v_mfma_f32_32x32x16_bf16: the
                                     speed mode; the reference's counterpart is its half-precision tcnn FullyFusedMLP
                                     option, decoder.py:43-59)                                  */
} NarutoFieldDesc;
#define NARUTO_MLP_FP32 0u
#define NARUTO_MLP_BF16 1u

/* Learnable parameters, in the reference's own layouts (state_dict tensors):
 *   table        embed_fn.params                         [n_entries * 2]
 *   uncert_grid  uncert_grid                             [Nx, Ny, Nz]
 *   sdf_w0       decoder.sdf_net.model.0.weight          [32, 80]  (in = 32 hash feats ++ 48 OneBlob)
 *   sdf_w1       decoder.sdf_net.model.2.weight          [16, 32]  (out = sdf ++ 15 geo feats)
 *   col_w0       decoder.color_net.model.0.weight        [32, 63]  (in = 48 OneBlob ++ 15 geo feats)
 *   col_w1       decoder.color_net.model.2.weight        [3, 32]                                   */
typedef struct NarutoParams {
    const float* table;
    const float* uncert_grid;
    const float* sdf_w0;
    const float* sdf_w1;
    const float* col_w0;
    const float* col_w1;
} NarutoParams;

typedef struct NarutoGrads {       /* same shapes; any pointer may be NULL = "not needed" */
    float* table;
    float* uncert_grid;
    float* sdf_w0;
    float* sdf_w1;
    float* col_w0;
    float* col_w1;
} NarutoGrads;

/* Where the M query points come from.  Either x != NULL: already-normalised points [M,3]
 * (query_sdf / query_color_sdf, scene_rep.py:98-148), or rays: M = n_rays * n_samples points
 * o + d * z, normalised by the bounding box (run_network [Co-SLAM], called at scene_rep.py:183-184). */
typedef struct NarutoPoints {
    const float* x;          /* [M,3] or NULL            */
    const float* rays_o;     /* [n_rays,3]               */
    const float* rays_d;     /* [n_rays,3]               */
    const float* z_vals;     /* [n_rays,n_samples]       */
    uint32_t     n_samples;
} NarutoPoints;

const char* naruto_last_error(void);
int naruto_version(void);

int  naruto_field_create(const NarutoFieldDesc* desc, NarutoField** out);
void naruto_field_destroy(NarutoField* f);
/* Level tables the library derived (tcnn GridEncodingTemplated constructor): host arrays of
 * n_levels (scale, resolution, size) and n_levels+1 (offset, in entries). */
int  naruto_field_levels(const NarutoField* f, float* scale, uint32_t* resolution, uint32_t* size, uint32_t* offset);
uint64_t naruto_field_n_entries(const NarutoField* f);

/* A1 -- depth sampling, scene_rep.py:158-180.  target_d may be NULL (then n_samples uniform depths,
 * scene_rep.py:171-173); rand [n_rays,S] may be NULL (perturb == 0).  S = n_samples_d + n_range_d
 * (or n_range_d if n_samples_d == 0, or n_samples if target_d == NULL).  z_vals [n_rays,S]. */
int naruto_sample_z(uint32_t n_rays, const float* target_d, float near_, float far_, uint32_t n_samples_d,
                    uint32_t n_range_d, float range_d, uint32_t n_samples, const float* rand,
                    float* z_vals, void* stream);

/* A3 alone -- embed_fn(x): query_sdf(embed=True), scene_rep.py:109-111.  feat [M,32] level-major. */
int naruto_hash_encode_fwd(const NarutoField* f, uint32_t M, const float* x, const float* table,
                           float* feat, void* stream);
/* its backward (tcnn HashGrid backward): d_table += scatter(d_feat).
 * workspace: naruto_scatter_workspace(f, M) bytes for lists of up to M points (per-split partial tables of the LDS-tiled
 * scatter; count matrix + 12-byte (corner, contribution) items of the binned scatter that serves levels of more than 2^17
 * entries).  No level uses global float atomics: the result is bitwise reproducible for any log2_hashmap_size <= 24.
 * naruto_field_scatter_overwrites(f): 1 if the scatter can WRITE the table gradient (NARUTO_BWD_OVERWRITE_TABLE_GRAD, the
 * fused optimiser) -- always, unless the debug switch NARUTO_DEBUG_SCATTER_ATOMIC sent the large levels through global atomics. */
size_t naruto_scatter_workspace(const NarutoField* f, uint32_t M);
int naruto_field_scatter_overwrites(const NarutoField* f);
int naruto_hash_encode_bwd(const NarutoField* f, uint32_t M, const float* x, const float* d_feat,
                           const float* d_feat_scale /* device scalar multiplying d_feat, or NULL */,
                           float* d_table, void* workspace, void* stream);

/* Feature-grid smoothness term of the mapping loss -- Co-SLAM CoSLAM.smoothness [not in tree], called by
 * get_loss_from_ret (coslam.py:166-169): TV of the hash features on a (sample_points-1)^3 lattice placed
 * at a random offset.  rand6 (device) = offset_rand[3] ++ jitter_rand[3] in [0,1).  Outputs: loss [1],
 * x_out [n^3,3] (the normalised lattice points) and d_feat [n^3,32] = d(loss)/d(features), which
 * naruto_hash_encode_bwd(x_out, d_feat, d_feat_scale = cotangent of the loss) turns into the table gradient. */
size_t naruto_smoothness_workspace(uint32_t sample_points);
int naruto_smoothness_fwd(const NarutoField* f, const float* table, uint32_t sample_points, float voxel_size,
                          float margin, const float* rand6, float* x_out, float* d_feat, float* loss,
                          void* workspace, void* stream);

/* A2-A5 fused -- calc_embedding + embedpos_fn + decoder (scene_rep.py:58-64,132-148, decoder.py:29-41,
 * 99-116).  Outputs (any may be NULL):
 *   raw        [M,5]  (r,g,b pre-sigmoid, sdf, uncert_raw)            -- query_color_sdf / run_network
 *   sdf_uncert [M,2]  (sdf, uncert_raw); colour net skipped when raw==NULL -- query_sdf(return_uncert)
 *   geo        [M,15]                                                  -- query_sdf(return_geo)
 *   feat_save  [16,M,2] hash features kept for naruto_query_bwd (training only)                    */
int naruto_query_fwd(const NarutoField* f, const NarutoParams* p, uint32_t M, const NarutoPoints* pts,
                     float* raw, float* sdf_uncert, float* geo, float* feat_save, void* stream);

/* Backward of naruto_query_fwd (autograd of the above through nn.Linear / tcnn / grid_sample).
 * d_raw [M,5] required; d_geo [M,15] optional (NULL = 0).  feat_save from the forward call.
 * active_idx / n_active (both NULL, or both given): a list of the point indices to process and its length in
 * DEVICE memory -- every point NOT in the list must have an all-zero cotangent (see naruto_compact_active).
 * workspace: naruto_query_bwd_workspace(M) bytes, contents undefined on entry and exit.
 * extra (optional): E more points whose feature cotangents are already known (the smoothness lattice of
 * naruto_smoothness_fwd); they are appended to the scatter's point list so that ONE scatter pass produces the
 * whole table gradient.  flags: NARUTO_BWD_OVERWRITE_* make the reductions WRITE the weight / table gradients
 * instead of accumulating (saves the caller the zero fill).
 * Workspace: naruto_query_bwd_workspace(f, M + E). */
typedef struct NarutoExtraPoints {
    const float* x;        /* [E,3] normalised points                         */
    const float* d_feat;   /* [E,32] cotangent of their hash features         */
    const float* scale;    /* device scalar multiplying d_feat, or NULL (= 1)  */
    uint32_t     n;        /* E                                               */
} NarutoExtraPoints;
#define NARUTO_BWD_OVERWRITE_WEIGHT_GRADS 1u
#define NARUTO_BWD_OVERWRITE_TABLE_GRAD 2u
/* naruto_train_backward in two calls (data parallel: the small MLP-gradient bucket is all-reduced while the table scatter runs):
 * MLP_ONLY = loss backward, compaction, MLP backward, weight gradients (complete after this call); TABLE_ONLY = the table scatter
 * over the point list the MLP_ONLY call left in the workspace.  Not with the fused optimiser. */
#define NARUTO_TRAIN_BWD_MLP_ONLY 4u
#define NARUTO_TRAIN_BWD_TABLE_ONLY 8u
/* Forward and backward issued back to back (single process): naruto_train_forward(finalize = NARUTO_TRAIN_FWD_DEFER_TAIL) stops
 * after the loss stage, and naruto_train_backward(flags | NARUTO_TRAIN_BWD_DEFERRED_TAIL) starts with ONE launch that is the loss
 * tail (one workgroup: losses[10], the iteration counter), the composite backward and the compaction -- instead of three.
 * losses / sums are then valid after the BACKWARD call.  Both must be given together; above 4096 rays both calls run the
 * ordinary sequence (same results). */
#define NARUTO_TRAIN_FWD_DEFER_TAIL 2
#define NARUTO_TRAIN_BWD_DEFERRED_TAIL 16u
/* Data parallel counterpart: the forward ran with finalize = 0 and t->sums now holds the ALL-REDUCED sums.  The backward (one piece
 * or its MLP_ONLY phase) then replaces naruto_train_finalize | composite backward | compaction by the same single launch, whose
 * extra workgroup turns the sums into losses[0..7] and the total.  Do not call naruto_train_finalize in addition. */
#define NARUTO_TRAIN_BWD_SUMS_GIVEN 32u
/* ... and its five-launch form (round 5): naruto_train_forward(finalize = NARUTO_TRAIN_FWD_SUMS_TV_LATER) stops at this rank's sums like
 * finalize = 0, but where the launch plan allows it the forward samples its own depths and only ENCODES the smoothness lattice (no
 * k_sample_encode launch); naruto_train_backward(NARUTO_TRAIN_BWD_SUMS_GIVEN | NARUTO_TRAIN_BWD_TV_MOVED) then evaluates the term in its
 * first launch and adds its value to losses[8] / losses[9] at its end.  losses[8] is 0 in between.  Both must be given together. */
#define NARUTO_TRAIN_FWD_SUMS_TV_LATER 3
#define NARUTO_TRAIN_BWD_TV_MOVED 64u
/* The model's sub-modules called on their own (forward only; the query entry points above never need them -- they evaluate all of
 * this in registers).  naruto_oneblob_fwd = embedpos_fn(x) (tcnn OneBlob, 16 bins): x [M,3] -> out [M,48].
 * naruto_decoder_fwd, by `part`:
 *   NARUTO_DECODER_FULL      decoder(embed, embed_pos) (decoder.py:99-116):  a = embed [M,33] (channel 0 = uncertainty sample, 1..32 =
 *                            hash features), b = embed_pos [M,48] -> out [M,5] = (rgb pre-sigmoid, sdf, the uncertainty channel)
 *   NARUTO_DECODER_SDF_NET   sdf_net(cat(embed, embed_pos)) (decoder.py:29-41): a = [M,81], b unused -> out [M,17] = (sdf, geo15, uncertainty)
 *   NARUTO_DECODER_COLOR_NET color_net(cat(embed_pos, geo)):                    a = [M,63], b unused -> out [M,3] (pre-sigmoid)          */
#define NARUTO_DECODER_FULL 0
#define NARUTO_DECODER_SDF_NET 1
#define NARUTO_DECODER_COLOR_NET 2
int naruto_oneblob_fwd(const NarutoField* f, uint32_t M, const float* x, float* out, void* stream);
/* calc_embedding's channel 0 on its own (scene_rep.py:58-64): out[m] = trilinear sample of uncert_grid at the normalised point x[m]
 * (grid_sample semantics of the reference's call: align_corners=False, zero padding, x <-> z transposed).  Forward only. */
int naruto_uncert_sample(const NarutoField* f, uint32_t M, const float* x, const float* uncert_grid, float* out, void* stream);
int naruto_decoder_fwd(const NarutoField* f, const NarutoParams* p, uint32_t M, int part, const float* a, const float* b, float* out, void* stream);
size_t naruto_query_bwd_workspace(const NarutoField* f, uint32_t M);
int naruto_query_bwd(const NarutoField* f, const NarutoParams* p, uint32_t M, const NarutoPoints* pts,
                     const float* feat_save, const float* d_raw, const float* d_geo,
                     const uint32_t* active_idx, const uint32_t* n_active,
                     const NarutoExtraPoints* extra, uint32_t flags,
                     const NarutoGrads* g, void* workspace, void* stream);

/* A1-A7 in ONE launch -- render_rays as an inference call (scene_rep.py:150-225; eval-mode forward, planner-side queries): depth
 * sampling as naruto_sample_z, the field query of naruto_query_fwd and the compositing of naruto_composite_fwd per ray, raw kept
 * in LDS.  Outputs (any may be NULL): rgb [N,3], depth, disp, acc, depth_var, uncert_map [N], weights [N,S], raw [N,S,5],
 * z_vals [N,S] -- the per-sample ones cost their global writes only when asked for.  Sampling arguments as naruto_sample_z
 * (target_d NULL: n_samples uniform depths); rand [N,S] or rng {seed, counter} or neither (no jitter).  Not differentiable:
 * training goes through naruto_train_forward / the autograd operators. */
typedef struct NarutoRender {
    uint32_t n_rays;
    const float *rays_o, *rays_d, *target_d;
    float near_, far_;
    uint32_t n_samples_d, n_range_d;
    float range_d;
    uint32_t n_samples;
    const float* rand;
    const uint64_t* rng;
    float *rgb, *depth, *disp, *acc, *depth_var, *uncert_map, *weights, *raw, *z_vals;
} NarutoRender;
int naruto_render_fwd(const NarutoField* f, const NarutoParams* p, const NarutoRender* r, void* stream);

/* A6+A7 -- sdf2weights [Co-SLAM] + raw2outputs (scene_rep.py:66-96).  Outputs (any may be NULL):
 * rgb [N,3], disp [N], acc [N], weights [N,S], depth [N], depth_var [N], uncert_map [N]. */
int naruto_composite_fwd(const NarutoField* f, uint32_t n_rays, uint32_t S, const float* raw,
                         const float* z_vals, float* rgb, float* disp, float* acc, float* weights,
                         float* depth, float* depth_var, float* uncert_map, void* stream);
/* Backward: cotangents of the outputs (any may be NULL = 0) -> d_raw [N,S,5].
 * accumulate != 0 adds into d_raw instead of overwriting it. */
int naruto_composite_bwd(const NarutoField* f, uint32_t n_rays, uint32_t S, const float* raw,
                         const float* z_vals, const float* d_rgb, const float* d_disp, const float* d_acc,
                         const float* d_weights, const float* d_depth, const float* d_depth_var,
                         const float* d_uncert_map, float* d_raw, int accumulate, void* stream);

/* A8 -- the mapping losses of JointEncodingNaruto.forward (scene_rep.py:246-285 + Co-SLAM
 * get_sdf_loss/get_masks).  Three steps so that data-parallel ranks can all-reduce the sums in
 * between (the loss weights depend on GLOBAL sample counts):
 *   1. naruto_loss_sums     : this rank's rays -> sums[NARUTO_LOSS_NSUMS] (fp64, device)
 *   2. (optional) all-reduce (SUM) of slots [0, NARUTO_LOSS_SLOT_MINUNCERT) over ranks; slot
 *      NARUTO_LOSS_SLOT_MINUNCERT is a minimum (MIN-reduce it, or keep it per rank: it only feeds an assertion)
 *   3. naruto_loss_finalize : sums (+ total ray count over all ranks) -> losses[8] =
 *      {rgb_loss, depth_loss, sdf_loss, fs_loss, psnr, uncert_loss, min(uncert_map), n_valid_depth}
 * workspace: naruto_loss_workspace(n_rays) bytes. */
#define NARUTO_LOSS_NSUMS 16
#define NARUTO_LOSS_SLOT_MINUNCERT 9
size_t naruto_loss_workspace(uint32_t n_rays);
int naruto_loss_sums(const NarutoField* f, uint32_t n_rays, uint32_t S, const float* raw, const float* z_vals,
                     const float* rgb, const float* depth, const float* uncert_map, const float* target_rgb,
                     const float* target_d, float depth_trunc, float rgb_missing, double* sums,
                     float* losses /* optional: single-process shortcut, = finalize(sums, n_rays) */,
                     void* workspace, void* stream);
int naruto_loss_finalize(const double* sums, uint64_t n_rays_total, uint32_t S, float* losses, void* stream);
/* Backward of the whole loss block down to d_raw [N,S,5] (composite backward fused in):
 * loss_grad [6] = d(total)/d{rgb_loss, depth_loss, sdf_loss, fs_loss, psnr(ignored), uncert_loss},
 * device array (e.g. the weights of get_loss_from_ret, coslam.py:154-174). */
int naruto_loss_bwd(const NarutoField* f, uint32_t n_rays, uint32_t S, const float* raw, const float* z_vals,
                    const float* target_rgb, const float* target_d, float depth_trunc, float rgb_missing,
                    const double* sums, uint64_t n_rays_total, const float* loss_grad, float* d_raw,
                    uint32_t* ray_count, void* stream);
/* The mapping losses leave the cotangent of every sample behind the surface band identically zero, and
 * samples are depth-sorted, so the non-zero part of each ray is a PREFIX: naruto_loss_bwd can report its
 * length per ray (ray_count [n_rays], optional), and naruto_compact_active turns the lengths into the
 * flat list naruto_query_bwd consumes: active_idx [<= n_rays*S], n_active [1]; ray_offset [n_rays] scratch. */
int naruto_compact_active(uint32_t n_rays, uint32_t S, const uint32_t* ray_count, uint32_t* ray_offset,
                          uint32_t* active_idx, uint32_t* n_active, void* stream);

/* A10 helper -- one fused Adam step over a flat fp32 buffer (torch.optim.Adam semantics incl. L2
 * weight_decay, reference coslam.py:409-419).  The 1-based step count comes from `step`, or -- when
 * step_dev != NULL -- from device memory (int32), which keeps the launch valid under hipGraph replay. */
int naruto_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, uint64_t n,
                     float lr, float beta1, float beta2, float eps, float weight_decay, uint32_t step,
                     const int32_t* step_dev, void* stream);

/* A9 caller -- get_map_volumes' post-processing (coslam_utils.py:89-95): sdf_uncert [M,2] from
 * naruto_query_fwd -> out [2,M] = (uncertainty volume: softplus(raw)+0.01 where 0 <= sdf < 0.5 else 0 | sdf volume). */
int naruto_map_volumes(uint32_t M, const float* sdf_uncert, float* out, void* stream);

/* N1 ("next" row) -- ActiveRaySampler.sample_rays (reference src/slam/coslam/active_ray_sampler.py:77-149) on the
 * device.  Input batch of n_total rays = [oversampled keyframe rays ..., n_cur current-frame rays]; base =
 * mapping.sample (2048), K = num_uncert_sample (500), n_tail = ceil(n_cur / oversample_mul).  Candidates are rays
 * [base, n_total - n_tail); the K with the SMALLEST cached-uncertainty value at their measured end point
 * (round((o + d*depth - bbox_min) * voxel_scale), clipped; numpy argpartition semantics, ties by lower index) are
 * moved to the front: out = [K selected | rays [0, base-K) | last n_tail rays], base + n_tail rows.
 * uncert_vol [X,Y,Z] fp32 on the device; vol_dims / bbox_min are HOST arrays of 3.
 * workspace: naruto_active_ray_workspace(n_total, K) bytes. */
size_t naruto_active_ray_workspace(uint32_t n_total, uint32_t K);
int naruto_active_ray_select(uint32_t n_total, uint32_t base, uint32_t K, uint32_t n_tail, const float* rays_o,
                             const float* rays_d, const float* target_s, const float* target_d,
                             const float* uncert_vol, const uint32_t* vol_dims, const float* bbox_min,
                             float voxel_scale, float* out_o, float* out_d, float* out_s, float* out_t,
                             void* workspace, void* stream);

/* naruto_active_ray_select with the candidates' keys given (keys[j] for candidate base + j, as NarutoRayBatch.keys_out leaves them): the
 * lookup -- two dependent trips to memory per candidate -- is skipped.  At most 8 192 candidates (NARUTO_ERR_INVALID beyond). */
int naruto_active_ray_select_keyed(uint32_t n_total, uint32_t base, uint32_t K, uint32_t n_tail, const float* rays_o,
                                   const float* rays_d, const float* target_s, const float* target_d, const uint32_t* keys,
                                   float* out_o, float* out_d, float* out_s, float* out_t, void* stream);

/* N2 ("next" row) -- camera-frame directions to world rays (coslam.py:342-344): rays_d[r] = R[pose_id[r]] . d_cam[r],
 * rays_o[r] = t[pose_id[r]]; poses [P,4,4] row-major camera-to-world, pose_id int64 [n]. */
int naruto_rays_to_world(uint32_t n, const float* d_cam, const int64_t* pose_id, const float* poses, float* rays_o,
                         float* rays_d, void* stream);

/* N2, store side -- one BA batch from a device-resident keyframe ray store (coslam.py:310-344 + Co-SLAM
 * KeyFrameDatabase.sample_global_rays [not in tree]): n_global DISTINCT rays drawn from the n_kf*rays_per_kf stored rows
 * (python random.sample semantics: without replacement), n_cur distinct pixels of the current frame (all pixels, or those
 * listed in cur_list = the valid-depth pixels), rotated to world with the pose of their keyframe
 * (frame_ids[kf] / keyframe_every; the current frame uses the LAST pose).  The distinct draw is a keyed Feistel
 * permutation of [0, n) with cycle walking: sample element i = perm(i); key = (seed, counter).
 * naruto_perm_index is the same permutation on the host (salt 2: store draw, 3: current-frame draw, 1: naruto_sample_distinct). */
typedef struct NarutoRayBatch {
    const float* store;        /* [n_kf*rays_per_kf, 7] (direction 3, rgb 3, depth 1), device                       */
    uint32_t n_kf, rays_per_kf;
    const int64_t* frame_ids;  /* [n_kf] device                                                                     */
    int64_t keyframe_every;
    uint32_t n_global;
    const float* current;      /* [pixels, 7] rays of the current frame                                             */
    const uint32_t* cur_list;  /* optional [n_cur_pop] admissible pixel indices; NULL: pixels 0..n_cur_pop-1         */
    uint64_t n_cur_pop;
    uint32_t n_cur;
    const float* poses;        /* [n_poses,4,4] camera-to-world                                                      */
    uint32_t n_poses;
    uint64_t seed, counter;
    float *rays_o, *rays_d, *target_s, *target_d;   /* [n_global+n_cur,3] x3, [n_global+n_cur]                      */
    int64_t* ids_out;          /* optional [n_global+n_cur]: pose index per ray, -1 for current-frame rays           */
    /* For a launch that is CAPTURED in a hipGraph and replayed over a growing store (optional, device memory):
     * rng = {seed, counter} -- the draw is keyed by (rng[0] ^ seed, rng[1] + counter), e.g. the trainer's iteration state, which
     * the training forward advances once per iteration; dyn = {n_kf, n_poses, n_cur_pop} replaces the three host values, so new
     * keyframes / poses / another current frame need no re-capture as long as n_global and n_cur stay the same.             */
    const uint64_t* rng;
    const uint64_t* dyn;
    /* optional (round 5): the active ray sampler's lookup done where the rows are in registers -- keys_out[r - key_base] = the sortable key
     * of row r's cached-uncertainty value (what naruto_active_ray_select derives from the row: round((o + d*depth - key_bbox_min) *
     * key_voxel_scale), clipped) for rows key_base <= r < n_global + n_cur - key_tail; consumed by naruto_active_ray_select_keyed.
     * keys_out NULL: off (the other key_* fields are then ignored). */
    uint32_t* keys_out;
    uint32_t key_base, key_tail;
    const float* key_vol;      /* [X,Y,Z] fp32, device                                                               */
    uint32_t key_dims[3];
    float key_bbox_min[3];
    float key_voxel_scale;
} NarutoRayBatch;
int naruto_assemble_rays(const NarutoRayBatch* b, void* stream);
/* N2 + N1 in ONE launch: naruto_assemble_rays | naruto_active_ray_select without the intermediate oversampled batch (coslam.py:310-359 as
 * one step of the mapping iteration).  Row r of the virtual batch is what naruto_assemble_rays would have written (b's own output
 * buffers and ids_out are ignored and may be NULL); base / K / n_tail / volume arguments and the result are naruto_active_ray_select's.
 * At most 8 192 candidates (n_global + n_cur - base - n_tail): NARUTO_ERR_INVALID beyond -- use the two calls. */
int naruto_assemble_select(const NarutoRayBatch* b, uint32_t base, uint32_t K, uint32_t n_tail, const float* uncert_vol,
                           const uint32_t* vol_dims, const float* bbox_min, float voxel_scale, float* out_o, float* out_d,
                           float* out_s, float* out_t, void* stream);
int naruto_sample_distinct(uint64_t n, uint32_t count, uint64_t seed, uint64_t counter, int64_t* out, void* stream);
uint64_t naruto_perm_index(uint64_t i, uint64_t n, uint64_t seed, uint64_t counter, uint64_t salt);

/* N3 ("next" row) -- the planner's uncertainty aggregation in goal space (reference src/planner/naruto_planner.py,
 * NarutoPlanner.uncertainty_aggregation_v2 :596-735), consuming the volumes of naruto_map_volumes.
 * naruto_goal_targets: the target observations (:629-632) -- the top_k largest uncertainty voxels (ties: lower flat index),
 *   listed in flat-index order and thinned to top_k_subset entries at positions floor(i*top_k/subset); targets int32
 *   [subset,3] voxel indices.  (The reference takes whatever numpy's argpartition leaves in the last `subset` slots: an
 *   unspecified subset of the top_k.)  dims: HOST array {X,Y,Z}; workspace naruto_goal_targets_workspace() bytes.
 * naruto_goal_aggregate (:637-710): collections[g][k] = uncert[target k] if min_dist < |goal g - target k| < max_dist
 *   (voxels), goal g is not on the border and sdf >= safe_sdf at the goal and its 6 neighbours, and the sdf is > 0 at the
 *   30 points of the segment goal -> target (truncated to voxels); else 0.  aggregated[g] = sum_k collections[g][k].
 *   goal_idx int32 [G,3], targets int32 [k,3]. */
size_t naruto_goal_targets_workspace(uint32_t n_voxels, uint32_t top_k);
int naruto_goal_targets(const uint32_t* dims, const float* uncert_vol, uint32_t top_k, uint32_t top_k_subset,
                        int32_t* targets, void* workspace, void* stream);
int naruto_goal_aggregate(const uint32_t* dims, const float* uncert_vol, const float* sdf_vol, uint32_t n_goals,
                          const int32_t* goal_idx, uint32_t n_targets, const int32_t* targets, float min_dist,
                          float max_dist, float safe_sdf, float* collections, float* aggregated, void* stream);

/* N4 ("next" row) -- the dense volume -> mesh path (reference src/slam/coslam/coslam_utils.py:100-226 extract_mesh,
 * callers coslam.py:421-492).  The reference pushes a host-built lattice through query_sdf in 65 536-point chunks with a
 * copy per chunk and runs the third-party `marching_cubes` module on the CPU (coslam_utils.py:26,145).
 * naruto_lattice_points: x[(i*Y + j)*Z + k] = (tx[i], ty[j], tz[k]) -- the (already normalised) lattice of
 *   coslam_utils.py:124-133 expanded on the device; feed x to naruto_query_fwd.  dims: HOST array {X,Y,Z}.
 * naruto_mesh_count: marching cubes over sdf_vol [X,Y,Z] (z fastest), pass 1: counts[0] = vertices, counts[1] =
 *   triangles (device uint64[2]); bit c of a cell's case = (double)value < isolevel at corner (c&1, (c>>1)&1, (c>>2)&1);
 *   cells with a corner |value| > truncation emit nothing; workspace: naruto_mesh_workspace(dims) bytes, kept for
 * naruto_mesh_emit: pass 2: vertices float64 [V,3] in lattice-index coordinates (one per crossed lattice edge, at
 *   t = (isolevel - v0) / (v1 - v0) in float64, ordered by (owner voxel, axis)), triangles int32 [F,3] ordered by
 *   (cell, case-table order), normals towards larger values.  At most cap_* entries are written. */
int naruto_lattice_points(const uint32_t* dims, const float* tx, const float* ty, const float* tz, float* x,
                          void* stream);
size_t naruto_mesh_workspace(const uint32_t* dims);
int naruto_mesh_count(const uint32_t* dims, const float* sdf_vol, double isolevel, double truncation,
                      void* workspace, uint64_t* counts, void* stream);
int naruto_mesh_emit(const uint32_t* dims, const float* sdf_vol, double isolevel, const void* workspace,
                     uint64_t cap_vertices, uint64_t cap_triangles, double* vertices, int32_t* triangles,
                     void* stream);

/* All parameter tensors of one optimiser in a single launch (<= 8 segments, per-segment lr / eps / weight_decay,
 * shared betas and step). */
typedef struct NarutoAdamSeg {
    float* param; const float* grad; float* exp_avg; float* exp_avg_sq;
    uint64_t n; float lr, eps, weight_decay;
    uint32_t step_lag;     /* this tensor's step number is the launch's minus step_lag (torch.optim.Adam counts steps PER PARAMETER:
                              one that was added later, or sat a step out without a gradient, lags behind); 0 in the mapping loop */
} NarutoAdamSeg;
#define NARUTO_ADAM_ADVANCE 1u   /* step_dev = int32[2] {completed steps, 0}: this launch is step step_dev[0]+1 and stores it back */
#define NARUTO_ADAM_ZERO_GRAD 2u /* zero every gradient once consumed (the segments' grad buffers are written) */
int naruto_adam_multi(const NarutoAdamSeg* segs /* host array */, uint32_t n_segs, float beta1, float beta2,
                      uint32_t step, int32_t* step_dev, uint32_t flags, void* stream);

/* ---- The mapping iteration as two calls: naruto_amd.trainer.MappingTrainer's fast path ----------------------------
 * What JointEncodingNaruto.forward (scene_rep.py:227-287) + get_loss_from_ret incl. Co-SLAM smoothness
 * (coslam.py:154-174) + loss.backward() do in one global_BA iteration (coslam.py:361-399), as few launches as
 * possible: side work rides in a bigger launch as extra workgroups, the small reductions share one tail launch
 * (naruto_train.hip).  Same kernels' arithmetic as the modular entry points above.
 *   naruto_train_forward : z_vals, raw, feat_save, rgb, depth, uncert_map, sums[16]; with finalize != 0 also
 *                          losses[10] = {rgb, depth, sdf, fs, psnr, uncert, min(uncert_map), n_valid,
 *                                        smoothness term, total = sum_i loss_weights[i] * losses[i]}
 *   (data parallel: finalize = 0, all-reduce sums[0..9), naruto_train_finalize)
 *   naruto_train_backward: gradient of the total w.r.t. the parameters in g (table / MLP weights written or
 *                          accumulated per flags as in naruto_query_bwd; uncert_grid always accumulated) */
typedef struct NarutoTrainStep {
    uint32_t n_rays, n_samples_d, n_range_d;          /* S = n_samples_d + n_range_d samples per ray          */
    uint32_t perturb;                                 /* != 0: stratified depth jitter (training.perturb > 0) */
    float near_, far_, range_d, depth_trunc, rgb_missing;
    uint32_t smooth_points;                           /* 0: no smoothness term (else Co-SLAM sample_points)   */
    float smooth_voxel, smooth_margin;
    float smooth_grad_scale;                          /* extra factor on the term's gradient (0 = 1; 1/world) */
    uint64_t n_rays_total;                            /* rays over all ranks (0: n_rays)                      */
    const float *rays_o, *rays_d, *target_rgb, *target_d;      /* [N,3] [N,3] [N,3] [N]                       */
    const float *rand;                                /* [N,S] depth jitter in [0,1), with perturb (or rng)   */
    const float *rand6;                               /* [6] lattice placement, with smooth_points (or rng)   */
    uint64_t *rng;                                    /* {seed, counter}: where rand / rand6 is NULL the kernels
                                                         draw their own numbers (splitmix64 keyed by seed, counter,
                                                         index); the counter advances once per forward.        */
    const float *loss_weights;                        /* [10] device: d(total)/d(losses[i]); slots 4,6,7,9 ignored */
    float *z_vals, *raw, *feat_save;                  /* [N,S] [N,S,5] [16][N*S][2]                           */
                                                      /* feat_save is PRIVATE to the forward / backward pair of one step: level-major as
                                                       * written above, or sample-major [N*S][16][2] where the forward runs in Morton
                                                       * order of the samples (tables > 64 MB, batches >= 4 M samples; round 6).  Same
                                                       * size either way; naruto_train_backward knows which from the same launch plan.  */
    float *rgb, *depth, *uncert_map;                  /* [N,3] [N] [N] (any may be NULL)                      */
    double *sums;                                     /* [NARUTO_LOSS_NSUMS]                                  */
    float *losses;                                    /* [10]                                                 */
    float *d_raw;                                     /* [N,S,5]          (backward)                          */
    uint32_t *ray_count, *ray_offset, *active_idx, *n_active;  /* [N] [N] [N*S] [1]  (backward)               */
    void *workspace;                                  /* naruto_train_workspace() bytes                       */
    const float *loss_weight_parts[10];               /* naruto_train_backward only, optional: per-slot device SCALARS added to
                                                         loss_weights (which may then be NULL = zeros) -- the cotangents autograd
                                                         hands back for the scalar losses of an unchanged caller's weighted sum
                                                         (coslam.py:154-174); gathered into the workspace by one tiny launch     */
    float *min_uncert_running;                        /* optional [1]: every iteration folds its min(uncert_map) into this word
                                                         (minimum; a NaN sticks) -- the reference's per-forward
                                                         `assert uncert_map.min() > 0` (scene_rep.py:280) as a value the host can
                                                         read whenever it likes, graph replays included; initialise to +inf   */
} NarutoTrainStep;
/* Optimiser in the backward (single process): the launch that finishes the gradients applies torch.optim.Adam
 * (amsgrad off, L2 weight decay; reference create_optimizer, coslam.py:409-419) to the table and the MLP weights in
 * place.  Tensor order: table, sdf_w0, sdf_w1, col_w0, col_w1.  (Levels of more than 2^17 entries are stepped by the last
 * kernel of the binned scatter, one 8 192-entry slice per workgroup.) */
typedef struct NarutoFusedAdam {
    float* param[5]; float* exp_avg[5]; float* exp_avg_sq[5];
    float lr[5], eps[5], weight_decay[5];
    float beta1, beta2;
    const int32_t* step_dev;              /* device int32: this step's 1-based number                               */
    /* optional (round 5): the NEXT iteration's ray batch -- naruto_assemble_rays' work rides in the launch that finishes the gradients
     * (nothing of this iteration reads the ray buffers any more by then; a device-side rng is read AFTER this iteration's forward
     * advanced it, i.e. it keys the next iteration's draw).  NULL: off. */
    const struct NarutoRayBatch* next_batch;
} NarutoFusedAdam;
size_t naruto_train_workspace(const NarutoField* f, const NarutoTrainStep* t);
int naruto_train_forward(const NarutoField* f, const NarutoParams* p, const NarutoTrainStep* t, int finalize, void* stream);
int naruto_train_finalize(const NarutoField* f, const NarutoTrainStep* t, void* stream);
int naruto_train_backward(const NarutoField* f, const NarutoParams* p, const NarutoTrainStep* t, const NarutoGrads* g,
                          uint32_t flags, const NarutoFusedAdam* opt /* NULL: gradients only; else g's table / weight
                          pointers may be NULL (gradients not materialised) */, void* stream);

/* Measurement aid (bench.py): ONLY the field-query launch of naruto_train_forward, exactly as the iteration issues it (one wave per ray
 * with early termination when S % 64 == 0 -- and then with the loss stage riding in the same launch, k_query_fwd_loss); t->z_vals
 * must hold a previous forward's depths. */
int naruto_debug_train_query_fwd(const NarutoField* f, const NarutoParams* p, const NarutoTrainStep* t, void* stream);
/* profiling: a device buffer of 16 x (ray workgroups) uint64 into which the packed training forward (k_query_fwd_loss_packed) stamps the
 * shader clock at the start and behind each step of every workgroup's first chunk (tools/fwd_timeline.py); NULL switches it off. */
int naruto_debug_fwd_timeline(void* device_buffer);
/* profiling (bench.py's roofline): k_hash_scatter_lds alone, over the point list the preceding naruto_train_backward left in the
 * workspace, in the launch shape of the iteration; writes the scatter's partial tables only (no gradient, no parameter). */
int naruto_debug_train_scatter(const NarutoField* f, const NarutoParams* p, const NarutoTrainStep* t, void* stream);

/* Measurement aid (bench.py, workloads whose table fits no cache): one launch that reads RANDOM 64-byte lines out of `table`
 * (table_bytes of it) -- the access pattern of the hash gather on the T = 2^22 levels, without the kernel around it; *n_lines_out
 * (host) = distinct lines the launch requests.  Timed with HIP events it gives the memory system's random-line rate, the ceiling
 * `roofline.random_line_roof` prices the gather against (tools/hbm_random_line_bench.hip: the standalone sweep). */
int naruto_debug_random_lines(const float* table, uint64_t table_bytes, uint32_t iters, float* sink, uint64_t* n_lines_out, void* stream);

/* Hardware self-checks used by the GPU tests: the MFMA / permlane layouts the kernels rely on.
 * out: device buffer of 64*16 floats; returns 0 and fills out (see tests/test_gpu_parity.py: test_mfma_layout, test_permlane32_swap). */
int naruto_debug_mfma_layout(const float* a, const float* b, float* out, void* stream);
int naruto_debug_permlane_swap(const float* v0, const float* v1, float* out, void* stream);
/* a [32,16], b [16,32] fp32 (rounded to bf16 inside) -> out [64*16]: lane l, reg r of D = A.B by v_mfma_f32_32x32x16_bf16 with
 * A lane l = A[l&31][8*(l>>5) + e], B lane l = B[8*(l>>5) + e][l&31], e = 0..7 (test_mfma_bf16_layout). */
int naruto_debug_mfma_bf16_layout(const float* a, const float* b, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NARUTO_HIP_H */
