/* clid_native.h -- C ABI of libclid_native.so: hand-written gfx950 (MI355X / CDNA4) HIP kernels for
 * CLID-SLAM's per-scan SDF training inner loop.
 *
 * The reference has NO native/FFI interface for this path: the path sits
 * behind Python methods.  Each entry point below therefore names the reference *Python* function
 * whose torch-op sequence it replaces (paths relative to the reference root).  The Python host
 * (the .py files of clid-slam_amd/, same class names and signatures as the reference) binds these through
 * ctypes; INTEGRATION.md shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer borrowed for the call unless the name ends in _host;
 *     no ownership transfer, no hidden allocation (workspaces are passed in);
 *   - `stream` is a hipStream_t (pass torch.cuda.current_stream().cuda_stream); all work is
 *     enqueued asynchronously on it, nothing synchronises;
 *   - return value: 0 = ok, negative = error; clid_last_error() returns a message;
 *   - indices are int32 on the device (the reference uses int64; the shim converts at the edge),
 *     -1 = invalid neighbour, exactly as in the reference;
 *   - compile-time shape contract (all shipped configs): feature_dim F = 8, position dim 3
 *     (pos_encoding_band = 0) => decoder input D = 11, one hidden layer H = 64 with bias + ReLU,
 *     query_nn_k K = 6.  Other shapes are rejected with CLID_E_SHAPE, never silently mis-computed.
 */
#ifndef CLID_NATIVE_H
#define CLID_NATIVE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CLID_F 8   /* config.feature_dim            (utils/config.py:127) */
#define CLID_D 11  /* F + 3                         (model/decoder.py:27-33) */
#define CLID_H 64  /* config.geo_mlp_hidden_dim     (utils/config.py:171) */
#define CLID_K 6   /* config.query_nn_k             (utils/config.py:546) */
#define CLID_MLP_PARAMS (CLID_H * CLID_D + CLID_H + CLID_H + 1) /* 833: W1|b1|W2|b2 */
#define CLID_GRAD_FEAT_OFFSET 836 /* feature gradients start here in the fused gradient buffer (16 B aligned) */
/* Accumulation-row layout of the fused gradient buffer (clid_train_args.grad_stride):
 *   8  (or 0): [836 | (M+1) rows of F = 8 floats]                      -- the layout of round 1
 *   16       : [848 | (M+1) rows of 16 floats = 8 gradients | certainty increment | 7 unused]
 * With 16-float rows (64-byte aligned) the scatter of one (query, neighbour) pair -- 8 gradient adds and the
 * certainty add of np.py:714 -- is ONE memory-side atomic request instead of two (the atomic units retire
 * ~17 G requests/s whatever their width up to 64 bytes: tools/ubench_atomic.hip); clid_train_adam folds the
 * certainty column into `cert` and zeroes it. */
#define CLID_GRAD_FEAT_OFFSET16 848
#define CLID_GRAD_ROW16 16
#define CLID_GRAD_OFFSET(stride) ((stride) == CLID_GRAD_ROW16 ? CLID_GRAD_FEAT_OFFSET16 : CLID_GRAD_FEAT_OFFSET)

#define CLID_OK 0
#define CLID_E_ARG (-1)
#define CLID_E_SHAPE (-2)
#define CLID_E_HIP (-3)
#define CLID_E_P2P_TIMEOUT (-4) /* a flag wait of the peer-mapped exchange gave up: the call's sums are invalid, repeat it over RCCL */

const char* clid_last_error(void);
int clid_abi_version(void);
/* Copy at most 256 bytes from device memory to `host_dst` through a pinned landing buffer and synchronise `stream`: the
 * one read-back of a map-maintenance call (data-dependent counts). */
int clid_read_back(const void* device_src, int32_t bytes, void* host_dst, void* stream);

/* ---- neural-point map view ------------------------------------------------------------------
 * What the kernels read of `NeuralPoints` (model/neural_points.py:79-133).  `tab`/`pos4` are a
 * compact mirror built by clid_table_build(); the remaining pointers alias the shim's own torch
 * tensors (local_* arrays for query_locally=True, global arrays otherwise). */
typedef struct clid_map_view {
  const int32_t* tab;      /* [2^log2cap][4] buckets of 4 keys (slot numbers, -1 empty, filled in order) */
  const float* tab_pos;    /* [2^log2cap][4][4] per key: x, y, z, point id (int bits) -> one load per hit */
  const float* pos4;       /* [M][4] xyz0 of the points addressed by the ids in `tab` */
  float* feat;             /* [(M+1)][F] latent features, last row = padding (np.py:532) */
  float* cert;             /* [M] point certainties (read; +atomicAdd when training_mode) */
  int32_t* ts_update;      /* [M] last-update stamps (atomicMax when query_ts given), may be NULL */
  const int32_t* delta;    /* [P] (sum_c dx_c * prime_c) mod buffer_size, non-negative */
  const uint32_t* filter;  /* [2^log2filter / 32] probe prefilter: bit set for every key of `tab` (may be NULL) */
  int32_t log2cap;
  int32_t M;
  int32_t P;               /* neighbor_K, 81 for num_nei_cells=2, search_alpha=0.5 (np.py:931-969) */
  int32_t buffer_size;     /* config.buffer_size, < 2^30 */
  float resolution;        /* voxel_size_m */
  float max_valid_dist2;   /* 3*((num_nei_cells+1)*res)^2 */
  int32_t layer_norm;      /* config.layer_norm_on */
  int32_t log2filter;      /* bits of `filter`, >= 8 per key recommended (false-positive rate <= 12 %) */
  int32_t weighted_first;  /* config.weighted_first (utils/config.py; True in every shipped config), read by the fused inference
                              entries clid_sdf_grad_x / clid_sdf_query / clid_track_model: 1 = blend the neighbours'
                              decoder inputs, decode once; 0 = decode every neighbour, blend the K SDFs */
  int32_t stencil_nc;      /* num_nei_cells of the search neighbourhood (np.py:931-969): `stencil_rows` has (2 nc + 1)^2 entries */
  /* cell directory (clid_cdir_build; all three NULL = not built): the local window's voxel occupancy as the searches see it */
  const int32_t* cdir_hdr;     /* [CLID_CDIR_HDR_INTS] device-resident header: origin, dims, validity (written by the build) */
  const uint32_t* cdir_words;  /* [words + 1][2] per 32 z-adjacent cells: occupancy bits | rank of the first hit (24 bits) + the
                                  next word's low 8 bits */
  const float* cdir_pos;       /* [hits][4] x, y, z, point id (int bits) of every occupied cell, in cell order */
  const uint32_t* stencil_rows; /* [(2 nc + 1)^2] per (dx, dy) row of the search neighbourhood: bit b set <=> (dx, dy, b - nc) is
                                  one of the P probe offsets */
} clid_map_view;

/* ---- cell directory of the local window (csrc/celldir.hip) ---------------------------------------------------------
 * A probe of cell c in the reference is buffer_pt_index[hash(c) mod B] -> travel-distance filter -> global2local
 * (model/neural_points.py:984-1009, 595-598), 81 times per query, and 3 of 4 probes find nothing.  Which cells yield a
 * local point does not depend on the query, so it is resolved once per table: over the bounding box of the window's points
 * (+ a margin) one bit per voxel = "the reference's chain for this cell's slot yields a local id >= 0" -- evaluated by
 * looking the cell's slot up in the compact table, so a point returned by a foreign colliding cell, a point shadowed in its
 * slot and the same point returned by two cells are all reproduced exactly -- plus, per 32 z-adjacent cells, the rank of the
 * word's first hit in `cdir_pos`, the (x, y, z, id) rows of all hits in cell order (x-major, then y, then z: the probe order
 * of np.py:931-969).  A stencil row of 2 nc + 1 z-adjacent cells is then ONE 8-byte load + bit tests instead of 2 nc + 1 x
 * (hash fold, prefilter bit, 4-key bucket compare), and the hits of a row are consecutive rows of `cdir_pos`.
 * Everything is sized on the device (no read-back): `words_cap` / `hits_cap` bound the arrays, the header's `valid` word
 * says whether the window fitted; the searches fall back to probing the table itself when it did not, and for query points
 * farther than nc cells outside the box (whose probes can only meet foreign collisions).
 *   hdr_out    [CLID_CDIR_HDR_INTS] i32, words_out [words_cap + 1][2] u32, pos_out [hits_cap][4] f32,
 *   scratch    [words_cap / 32 + 2] i32 (hit counts per 32 words: the rank scan)
 *   pos4 / n   the window's points (clid_table_build's pos4_out), tab / tab_pos / filter: its table */
#define CLID_TRACK_COPIES 16
#define CLID_CDIR_HDR_INTS 16
#define CLID_CDIR_MARGIN_XY 8  /* cells around the points' box along x and y (along z: >= 2, whatever the column's words leave) */
int clid_cdir_build(const float* pos4, int32_t n, const int32_t* tab, const float* tab_pos, int32_t log2cap,
                    const uint32_t* filter, int32_t log2filter, int64_t buffer_size, float resolution, int32_t* hdr_out,
                    uint32_t* words_out, int64_t words_cap, float* pos_out, int64_t hits_cap, int32_t* scratch, void* stream);

/* Builds the compact probe table for one (map, window, time-filter) state.
 * Replaces nothing in the reference by itself: it is the exact, smaller equivalent of indexing
 * `buffer_pt_index[hash]` followed by the travel-distance filter and `global2local`
 * (model/neural_points.py:984-1009, 595-598): slot s keeps id j iff buffer_pt_index[s] == ids[j]
 * and (no time filtering or |travel[cur_ts]-travel[ts_create[ids[j]]]| < diff_travel).
 *   ids            [n] int64 global point index of local point j, or NULL for identity (global map)
 *   neural_points  [Mg][3], buffer_pt_index [buffer_size] int64, point_ts_create [Mg] int32
 *   tab_out        [2^log2cap][4] int32 keys (4-key buckets, 2^log2cap >= n/2 ... load <= 0.5 keys/bucket
 *                  recommended), tab_pos_out [2^log2cap][4][4] f32, pos4_out [n][4]
 *   filter_out     [2^log2filter / 32] u32 or NULL: one-hash Bloom filter over the stored slot numbers; the
 *                  chunked search kernel keeps it in LDS and skips the probes it rules out (never a false
 *                  negative, so results are unchanged) */
int clid_table_build(const int64_t* ids, int32_t n, const float* neural_points,
                     const int64_t* buffer_pt_index, int64_t buffer_size, float resolution,
                     const int32_t* point_ts_create, const float* travel_dist, int32_t cur_ts,
                     int32_t time_filtering, float diff_travel, int32_t* tab_out, float* tab_pos_out,
                     int32_t log2cap, float* pos4_out, uint32_t* filter_out, int32_t log2filter, void* stream);

/* NeuralPoints.radius_neighborhood_search (model/neural_points.py:971-1030).
 * dist2_out [N][P] f32, idx_out [N][P] int32 (ids of the view, -1 invalid). */
int clid_radius_search(const clid_map_view* mv, const float* x, int32_t N, float* dist2_out,
                       int32_t* idx_out, void* stream);

/* NeuralPoints.query_certainty (model/neural_points.py:1032-1051) on the GLOBAL map, probing the reference's own
 * buffer_pt_index [buffer_size] int64 directly (no mirror): cert_out[n] = max over the P cells of the certainty of the
 * point stored there (0 where the cell is empty or holds a colliding foreign point).  delta [P] as in clid_map_view. */
int clid_query_certainty(const int64_t* buffer_pt_index, int64_t buffer_size, const float* neural_points,
                         const float* point_certainties, const int32_t* delta, int32_t P, float resolution,
                         float max_valid_dist2, const float* x, int32_t N, float* cert_out, void* stream);

/* NeuralPoints.query_feature (model/neural_points.py:553-769), geometry features.
 *   query_ts  [N] int32 or NULL;  training_mode: certainty/ts side effects (np.py:708-733)
 *   weighted_first: feat_out [N][D] else [N][K][D]
 *   w_out [N][K], idx_out [N][K] int32 (sorted by distance, -1 invalid), nn_out [N] int32
 *   (count over all P probes, np.py:600-602), cert_out [N] (pre-update certainties, np.py:654). */
int clid_query_fwd(const clid_map_view* mv, const float* x, const int32_t* query_ts, int32_t N,
                   int32_t training_mode, int32_t weighted_first, float* feat_out, float* w_out,
                   int32_t* idx_out, int32_t* nn_out, float* cert_out, void* stream);

/* Backward of clid_query_fwd for autograd callers (what autograd does for np.py:553-769):
 *   g_feat [N][D] (weighted_first) or [N][K][D]; g_w [N][K] or NULL
 *   g_theta (+=, atomics) [(M+1)][F] or NULL;  g_x_out [N][3] or NULL
 * idx/w are the tensors clid_query_fwd returned for the same x. */
int clid_query_bwd(const clid_map_view* mv, const float* x, const int32_t* idx, const float* w,
                   int32_t N, int32_t weighted_first, const float* g_feat, const float* g_w,
                   float* g_theta, float* g_x_out, void* stream);

/* Decoder.sdf (model/decoder.py:58-82): sdf = scale*(W2 relu(W1 f + b1) + b2); rows = N (or N*K). */
int clid_mlp_sdf_fwd(const float* W1, const float* b1, const float* W2, const float* b2,
                     float sdf_scale, const float* feat, int32_t rows, float* sdf_out, void* stream);
/* its backward: g_feat_out [rows][D] or NULL; g_mlp (+=) [833] = dW1|db1|dW2|db2 or NULL. */
int clid_mlp_sdf_bwd(const float* W1, const float* b1, const float* W2, const float* b2,
                     float sdf_scale, const float* feat, const float* g_sdf, int32_t rows,
                     float* g_feat_out, float* g_mlp, void* stream);

/* Fused inference: query (training_mode=False) + Decoder.sdf + analytic d sdf/d x, i.e.
 * query_feature -> sdf -> utils/tools.py:298-311 get_gradient, as used by
 * utils/error_state_iekf.py:209-227.  grad_out [N][3], nn_out [N], cert_out [N] (may be NULL). */
int clid_sdf_grad_x(const clid_map_view* mv, const float* W1, const float* b1, const float* W2,
                    const float* b2, float sdf_scale, const float* x, int32_t N, float* sdf_out,
                    float* grad_out, int32_t* nn_out, float* cert_out, void* stream);

/* Dense inference for meshing, the SDF part of Mesher.query_points (utils/mesher.py:38-163): sdf_out [N]
 * (0 where no neighbour exists, :122-128) and nn_out [N] for the marching-cubes mask (:156-161). */
int clid_sdf_query(const clid_map_view* mv, const float* W1, const float* b1, const float* W2,
                   const float* b2, float sdf_scale, const float* x, int32_t N, float* sdf_out,
                   int32_t* nn_out, void* stream);

/* Tracking measurement model, IEKFOM.h_model (utils/error_state_iekf.py:176-264):
 * p_map = R p_imu + t, sdf + analytic gradient at p_map, validity mask (nn >= min_nn, min < |g| < max, and -- with
 * mv->weighted_first == 0 -- std of the neighbours' SDFs < max_sdf_std, :217-225 / :236-241),
 * and -- when normal_eq != NULL -- the float64 sums update_iterated (:299-305) needs instead of the N x 18 H:
 *   normal_eq [CLID_TRACK_COPIES][32] float64 (+=, zero it first): 16 partial copies (one per 256 bytes; a block adds to copy
 *   block mod 16, so the atomics do not queue on one cache line) whose SUM over the copies is
 *     [0..20]  upper triangle (row-major) of the 6x6 block of S = H^T R_inv H,  [21..26] H^T R_inv z,  [27] number of valid points
 * rot_host [9] row-major / pos_host [3]: HOST floats (the filter state lives on the host).
 * Per-point outputs (any may be NULL): sdf [N], grad [N][3], pmap [N][3], valid [N] int32. */
int clid_track_model(const clid_map_view* mv, const float* W1, const float* b1, const float* W2,
                     const float* b2, float sdf_scale, const float* rot_host, const float* pos_host,
                     int32_t min_nn, float min_grad_norm, float max_grad_norm, float max_sdf_std, const float* pc_imu,
                     int32_t N, float* sdf_out, float* grad_out, float* pmap_out, int32_t* valid_out,
                     double* normal_eq, void* stream);
/* the same with the pose in DEVICE memory (rot_dev [9] row-major fp32, pos_dev [3] fp32): the reference's filter keeps
 * x.rot / x.pos in device tensors (utils/error_state_iekf.py:176-186); no host round trip in front of the launch */
int clid_track_model_dev(const clid_map_view* mv, const float* W1, const float* b1, const float* W2,
                     const float* b2, float sdf_scale, const float* rot_dev, const float* pos_dev,
                     int32_t min_nn, float min_grad_norm, float max_grad_norm, float max_sdf_std, const float* pc_imu,
                     int32_t N, float* sdf_out, float* grad_out, float* pmap_out, int32_t* valid_out,
                     double* normal_eq, void* stream);

/* The measurement model iterated (utils/error_state_iekf.py:286-305 calls h_model up to max_iteration times per scan with the
 * SAME points and map; only the pose moves): everything of a clid_track_model call that stays fixed over those iterations, built
 * once per scan by the host (ABI 6).  clid_track_model_call then takes the pose and the reduction buffers only:
 *   normal_eq   [CLID_TRACK_COPIES][32] float64, ALREADY ZERO (may be NULL: per-point outputs only);
 *   zero_next   NULL or another [CLID_TRACK_COPIES][32] buffer the launch clears for the next call (a ring of three: accumulate
 *               into one, clear the next, the previous result stays readable) -- no fill launch per evaluation;
 *   result      NULL or the DEVICE address of 32 float64 in pinned, host-mapped memory (clid_pinned_alloc): a one-block launch
 *               behind the model adds the copies up, writes sums [0..27] there and then `epoch` (as a double) into [31]; the
 *               host polls [31] -- no copy, no stream synchronisation, no torch launches for the 28 numbers. */
typedef struct clid_track_call {
  clid_map_view mv;
  const float* W1; const float* b1; const float* W2; const float* b2;
  float sdf_scale;
  int32_t min_nn;
  float min_grad_norm, max_grad_norm, max_sdf_std;
  int32_t N;
  const float* pc_imu;
  float* sdf_out; float* grad_out; float* pmap_out; int32_t* valid_out;   /* per-point outputs, any may be NULL */
} clid_track_call;
int clid_track_model_call(const clid_track_call* c, const float* rot, const float* pos, int32_t pose_on_device,
                          double* normal_eq, double* zero_next, double* result, double epoch, void* stream);
/* The outputs of IEKFOM.h_model (utils/error_state_iekf.py:243-262) from the per-point outputs of a clid_track_model call, on
 * the device: clid_track_valid_count counts the valid points per block of 256 (block_prefix [ceil(N / 256) + 1] int32 receives
 * the EXCLUSIVE prefix, the total last) and writes the total as a double into result[29], then `epoch` into result[31] (pinned,
 * host-mapped: the host polls, allocates the four outputs and calls) clid_track_rows, which compacts in point order:
 *   z [Nv] f64 = sdf,  H [Nv][18] f64 = [p_imu x (R^T g) | g | 0 ... 0] (fp32 products, like the reference's fp32 bmm's),
 *   valid_points [Nv][3] f32 (map frame),  r_inv [Nv] f64 = 1 / (1 + (|g| - 1)^2) * 0.4 / (0.4 + z^2) * 1000. */
int clid_track_valid_count(const int32_t* valid, int32_t N, int32_t* block_prefix, double* result, double epoch, void* stream);
int clid_track_rows(const clid_track_call* c, const float* rot, const float* pos, int32_t pose_on_device,
                    const int32_t* block_prefix, double* z_out, double* H_out, float* valid_points_out, double* r_inv_out,
                    void* stream);
/* `bytes` of pinned host memory mapped into the device's address space: *host_out for the CPU, *dev_out for kernels */
int clid_pinned_alloc(int64_t bytes, void** host_out, void** dev_out);
void clid_pinned_free(void* host_ptr);

/* utils/loss.py:44-62 sdf_bce_loss (weighted, mean) + eikonal term (utils/mapper.py:779-798):
 * loss_out[0..2] = total, bce, eikonal (+=, zero it first); d_pred_out [N]; d_g_out [Ng][3]. */
int clid_loss_fwd_bwd(const float* pred, const float* label, const float* weight, int32_t N,
                      float sigma, int32_t weighted, const float* g, int32_t Ng, float weight_e,
                      float* loss_out, float* d_pred_out, float* d_g_out, void* stream);

/* torch.optim.Adam single-tensor step (utils/tools.py:205-255), optionally zeroing
 * the gradient in the same pass.  step >= 1. */
int clid_adam_step(float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1,
                   float beta2, float eps, float weight_decay, int32_t step, int32_t zero_grad,
                   void* stream);

/* ---- the fused mapping iteration (Mapper.mapping body, utils/mapper.py:642-836) -------------- */
typedef struct clid_train_args {
  /* sample pool (utils/mapper.py:84-97) and the batch indices get_batch drew (mapper.py:473-500) */
  const float* pool_coord;   /* [S][3] global_coord_pool */
  const float* pool_label;   /* [S] */
  const int32_t* pool_ts;    /* [S] */
  const float* pool_weight;  /* [S] */
  const int64_t* index;      /* [bs] this iteration's batch (this rank's shard) */
  int32_t bs;                /* local batch size */
  int32_t decimation;        /* config.gradient_decimation */
  int64_t batch_offset;      /* position of index[0] in the GLOBAL batch (multi-GPU); 0 otherwise */
  float fd_eps;              /* voxel_size_m * num_grad_step_ratio (mapper.py:703) */
  float inv_n_main;          /* 1 / global batch size   (BCE mean) */
  float inv_n_eik;           /* 1 / global decimated count (eikonal mean) */
  float sigma;               /* sdf_scale used by the loss (mapper.py:71) */
  float weight_e;
  int32_t loss_weight_on;
  int32_t eikonal_mode;      /* 0 off, 1 numerical (default), 2 analytic */
  int32_t train_decoder;     /* 0 after freeze_model (utils/tools.py:314) */
  /* decoder (model/decoder.py) */
  float* W1; float* b1; float* W2; float* b2;
  float sdf_scale;
  int32_t defer_reduce;      /* 1: leave the per-block partials of the 833 decoder gradients / loss sums in `ws`
                                for clid_train_adam(.., this, ..) to fold into its launch (single GPU);
                                0: reduce here so `grad` is complete on return (needed before an all-reduce) */
  /* gradients: one contiguous buffer [833 (+3 pad) | (M+1)*F] so a single all-reduce covers it */
  float* grad;
  /* workspace, sized by clid_train_workspace_floats() */
  float* ws;
  float* loss_out;           /* [4] total,bce,eik,unused (+=) */
  int32_t debug_flags;       /* 0 in production; bit 0 / bit 1 suppress the certainty / gradient atomics (timing ablation);
                                bit 2 makes clid_train_search take its full-depth path for every wave (tests) */
  int32_t grad_stride;       /* floats per feature row of `grad`: 0 or 8 (compact), 16 (see CLID_GRAD_ROW16) */
  /* ---- per-call switches (ABI 3: these were process-global setters in ABI 2; the library now keeps NO state between
   * calls except what the caller's buffers hold) */
  int32_t decode_variant;    /* kernel of clid_train_decode (numerical-eikonal / no-eikonal modes):
                                  0  k_train_fused8<2>: 16 lanes per query, decoder on the VALU, dW1 on fp32 MFMA (round 1);
                                  1  k_decode_tile<f32>: one wave per 16-query tile, the decoder contractions of
                                     model/decoder.py:58-82 and their transposes on v_mfma_f32_16x16x4_f32 (exact fp32);
                                  2  k_decode_tile<bf16>: the same on v_mfma_f32_16x16x32_bf16, fp32 accumulation
                                     (BASELINE.json configs[2]; outside the 1e-4 parity bar, reports its own error).
                                1 / 2 need grad_stride == 16 and fall back to 0 otherwise (clid_train_decode_kernel tells) */
  int32_t pipeline;          /* schedule of clid_mapping_run / clid_mapping_run_dist: 1 = clid_train_search over a chunk of
                                iterations, then per iteration clid_train_decode + Adam; 0 = per iteration the fused
                                search+decode kernel of clid_train_fwd_bwd + Adam */
  float* sdf_dbg;            /* test aid, normally NULL: the tile decode kernels also store the SDF they predict for every
                                record slot, [n_tasks][8] floats in the order of the search records */
  struct clid_prof* prof;    /* measurement aid, normally NULL: clid_profile_create() */
  /* ---- touched-row bookkeeping of the hoisted-search loop (NULL = off: dense all-reduce payload, dense Adam sweep).
   * The searches of a chunk know every map row the chunk's iterations will touch before the first decode runs:
   *   touch_ws  clid_touch_workspace_bytes(Mcap, chunk) bytes, ALL ZERO when first handed over and owned by the loop until
   *             the mapping() call ends.  Layout: flags u8 [chunk][touch_stride] first (clid_train_search sets
   *             flags[i][j] = 1 for every neighbour row j of iteration i of the chunk; with several ranks the caller
   *             MAX-all-reduces this block before clid_train_touch_scan), the rest is internal (bit words, word prefix
   *             sums, per-iteration counts, first-touch iteration per row).
   *   touch_iter  the iteration's position inside its chunk (clid_train_decode / clid_train_adam).
   *   cbuf      world > 1 only: the compact exchange buffer [848 decoder gradients | 8 x n gradient columns | n certainty
   *             increments] of the iteration's n touched rows in ascending row order, capacity 848 + 9 (M + 1) floats;
   *             clid_train_decode (defer_reduce == 0) packs into it and zeroes the accumulation rows it read, the caller
   *             all-reduces 848 + 9 n floats, clid_train_adam reads the gradients from it. */
  uint8_t* touch_ws;
  int64_t touch_stride;      /* clid_touch_stride(Mcap) of a capacity Mcap >= M: the layout stays put while the map grows */
  int32_t touch_iter;
  int32_t touch_all;         /* clid_mapping_run_dist, compact exchange: 1 = every row of the local map counts as touched in
                              * every iteration (lists of M rows: no flag exchange, no count read-back -- for local maps
                              * small enough that the ranks' batches reach all of it anyway; the packed layout and the
                              * peer-mapped transport stay).  Rows nobody touched carry zero gradients: their update is zero. */
  float* cbuf;
  /* world > 1, compact exchange only: peer-mapped exchange buffers (clid_p2p_*, below).  clid_mapping_run_dist then packs
   * every iteration into the object's current buffer instead of `cbuf` and sums it over the ranks with ONE launch per
   * rank instead of an RCCL ring; NULL, a rank count other than the communicator's, or a capacity below
   * 848 + 9 (M + 1) floats keep RCCL for the call. */
  struct clid_p2p* p2p;
  /* 1 = `neuralpoints.weighted_first: False` (utils/mapper.py:679-680): every neighbour's own decoder input is decoded and
   * the K SDFs are blended with the IDW weights, for the samples and the shifted copies alike (csrc/train_wf0.hip).
   * Hoisted schedule (clid_train_search / clid_train_decode / clid_mapping_run*), eikonal modes 0 and 1, plain exchange;
   * 0 (the reference's default, every shipped config) = decode the blended input.  clid_train_decode then keeps two
   * intermediate values per shifted copy in the copy's own record slot (the label / weight fields, unused for a copy): the
   * one case in which it writes to `rec`. */
  int32_t decode_each_neighbour;
  /* ---- overlapped search schedule of clid_mapping_run (ABI 6; NULL = every search of a chunk runs in front of the chunk's
   * decode -> Adam chain).  With a schedule object (clid_sched_create: a side stream, optionally confined to a CU mask, and a
   * ring of events) only iteration 0's search stays in front of the first decode; the searches of iterations 1 .. n-1 --
   * which read nothing training writes (positions, directory, sample indices) -- go out on the side stream in launches of
   * `side_group` iterations (0 = growing groups 1, 2, 4, 8, ...) on grids of at most `side_blocks` blocks (0 = the resident
   * grid), beside the chain; the decode of the first iteration of a group waits for the group's event.  Same records, same
   * results.  Single GPU, dense Adam sweep (touch_ws == NULL) only; otherwise the field is ignored. */
  struct clid_sched* sched;
  int32_t side_group;
  int32_t side_blocks;
  /* ---- config.ekional_add_to (utils/mapper.py:779-789; ABI 6): 0 = "all" (every shipped config), 1 = "surface", 2 =
   * "freespace": the eikonal term is the mean over the decimated samples with |sdf_label| < eik_mask_range
   * (config.surface_sample_range_m) / over the others.  The size of that subset is data: clid_train_search (and
   * clid_mapping_run per chunk) counts it per iteration and leaves 1 / size in eik_inv_n [>= 32] floats (device), position =
   * touch_iter; the decode / Adam launches normalise by it (an empty subset: 0 -- the reference's mean of an empty tensor is NaN).
   * Tile decode kernels (decode_variant 1 / 2), numerical eikonal term, one rank; other combinations are rejected. */
  int32_t eik_mask;
  float eik_mask_range;
  float* eik_inv_n;
  /* ---- ABI 7: two more branches of the reference's loop body.
   * main_loss_type = config.main_loss_type (utils/mapper.py:751-767): 0 "bce" (utils/loss.py:44-62; every shipped config),
   * 1 "sdf_l1", 2 "sdf_l2" (sdf_diff_loss, utils/loss.py:9-17: the sample weight is ALWAYS applied -- the caller sets
   * loss_weight_on = 1 so that the search records carry it), 3 "zhong" (sdf_zhong_loss, utils/loss.py:66-84, trunc_dist None).
   * Non-zero values run on the tile decode kernels of the hoisted schedule (eikonal modes 0 / 1, weighted_first); loss_out[1]
   * is then that loss.
   * pool_pose != NULL = Mapper.ba_done_flag (utils/mapper.py:646-658): `pool_coord` holds the samples in their SENSOR frames
   * (Mapper.coord_pool) and a sample drawn from frame ts = pool_ts[s] is moved to the world frame by row ts of
   * pool_pose [n_pose][12] (fp32, the upper 3 x 4 block of used_poses[ts], row-major) as utils/tools.py:612-636 does:
   * ((r0 x + r1 y) + r2 z) + t per coordinate, unfused.  Needs pool_ts; ts is clamped to [0, n_pose). */
  int32_t main_loss_type;
  int32_t n_pose;
  const float* pool_pose;
  /* config.proj_correction_on (utils/mapper.py:712-714, "[not used]" there): the label of a sample is scaled by
   * |cos(g, x - origin)|, g = the analytic d sdf / d x of the sample (in the autograd graph: the BCE term then also back-propagates
   * through g), origin = the translation of frame_pose[ts] -- frame_pose [n_frame_pose][12] as pool_pose's layout (they may be the
   * same array).  Analytic eikonal mode (eikonal_mode 2) on the hoisted schedule with weighted_first; 0 = off. */
  int32_t proj_correction;
  int32_t n_frame_pose;
  const float* frame_pose;
  /* config.consistency_loss_on (utils/mapper.py:716-741, 770-776): 1 - cos between the analytic gradient g of drawn samples and of
   * randomly shifted copies of them.  The copies are a second batch (own clid_train_args: their coordinates as its pool, zero
   * weights); both batches run the analytic iteration twice per training iteration (eikonal_mode 2, hoisted records):
   *   g_out != NULL    probe: clid_train_decode only evaluates g and stores it, [bs][3]; no side effects, no gradients;
   *   clid_consistency_couple turns the two g arrays into the term's value and dL/dg of both batches;
   *   c_extra != NULL  the backward proper with that dL/dg [bs][3] added to the eikonal term's (and proj_correction's) own.
   * partial_row0: first row of the launch's per-block partials in the workspace (the second batch goes behind the first);
   * partial_rows_extra: rows behind this batch's own that clid_train_adam also folds in. */
  float* g_out;
  const float* c_extra;
  int32_t partial_row0;
  int32_t partial_rows_extra;
  /* ---- ABI 8: world > 1 with the DENSE exchange (cbuf == NULL, defer_reduce == 0), tile decode kernels: dec_copies != NULL =
   * [n_dec_copies][848] floats that are part of the all-reduced buffer (Mapper puts them behind the accumulation rows of `grad`),
   * all zero when handed over.  The decode launch's blocks then ADD their decoder-gradient sums into copy (block % n_dec_copies)
   * -- 833 gradients and the two raw loss sums, as a partial row holds them; an address sees blocks / n_dec_copies adds instead of
   * a reduction launch between decode and all-reduce summing per-block rows (same-line atomics serialise at ~170 ns each) --;
   * clid_train_adam adds the copies up per column, finishes the losses and zeroes the copies.  The sharded iteration is then
   * decode -> all-reduce -> Adam.  NULL: per-block partial rows + the reduction launch (single-GPU probes, the compact exchange). */
  float* dec_copies;
  int32_t n_dec_copies;
  int32_t dec_ranks;         /* the ranks the all-reduce sums over (>= 1): the copies' two loss columns then hold the raw sums of ALL
                              * ranks; clid_train_adam adds 1 / dec_ranks of the normalised total to this rank's loss_out */
} clid_train_args;

/* Schedule object for clid_train_args.sched.  cu_mask / mask_words as hipExtStreamCreateWithCUMask takes them (bit i set =
 * the side stream's kernels may use compute unit i in the driver's numbering: on an 8-XCD part consecutive bits walk the
 * XCDs first, tools/cu_mask_census.py); NULL / 0 = an ordinary non-blocking stream.  priority: 0 default, -1 the device's
 * lowest, +1 its highest.  The object belongs to one host thread and one device. */
typedef struct clid_sched clid_sched;
int clid_sched_create(const uint32_t* cu_mask, int32_t mask_words, int32_t priority, clid_sched** out);
void clid_sched_destroy(clid_sched* s);
/* measurement aid: one word per block of a launch of n_blocks x 64 threads on the object's side stream (sched NULL: on
 * `stream`) = (XCC id << 16) | the low 16 bits of HW_ID (CU, shader array and engine ids); every block holds its CU for
 * hold_cycles so the dispatcher spreads the grid.  Synchronises. */
int clid_debug_cu_census(clid_sched* s, int32_t* out_host, int32_t n_blocks, int32_t hold_cycles, void* stream);

/* Touched-row workspace (see clid_train_args.touch_ws).  clid_train_touch_scan turns the chunk's flags (after the
 * cross-rank MAX when sharded) into per-iteration bit sets + prefix sums + counts and the per-row first-touch iteration
 * (call-global: it0 = index of the chunk's first iteration in the mapping() call; it0 == 0 resets it), and clears the
 * flags for the next chunk.  counts_host != NULL: the n_it per-iteration row counts are copied back and `stream` is
 * synchronised (the sharded loop needs them to size its all-reduces); NULL: nothing synchronises. */
int64_t clid_touch_stride(int32_t M);
int64_t clid_touch_workspace_bytes(int32_t M, int32_t chunk_iters);
int clid_train_touch_scan(const clid_train_args* t, int32_t M, int32_t n_it, int32_t it0, int32_t* counts_host, void* stream);
/* iterations clid_mapping_run / clid_mapping_run_dist put into one search launch for these arguments (<= 32) */
int32_t clid_train_chunk_iters(const clid_train_args* t);

int64_t clid_train_workspace_floats(int32_t bs, int32_t decimation, int32_t eikonal_mode);
/* forward + loss + backward of one iteration; leaves summed gradients in args->grad */
int clid_train_fwd_bwd(const clid_map_view* mv, const clid_train_args* args, void* stream);
/* Adam on decoder (when trained) + features from args->grad; zeroes args->grad */
typedef struct clid_adam_args {
  float* feat; float* grad; float* m; float* v;       /* feat/m/v: [(M+1)*F]; grad as above [833|..] */
  float* W1; float* b1; float* W2; float* b2;          /* decoder params */
  float* m_mlp; float* v_mlp;                          /* [833] each */
  int64_t n_feat;                                      /* (M+1)*F */
  float lr, beta1, beta2, eps, weight_decay;
  int32_t step;                                        /* 1-based, restarts every mapping() call */
  int32_t train_decoder;
  int32_t grad_stride;                                 /* as clid_train_args.grad_stride */
  float* cert;                                         /* [n_cert] local_point_certainties: += column 8 of the 16-float rows */
  int32_t n_cert;                                      /* M (0 / NULL with 8-float rows) */
  int32_t pad0;
} clid_adam_args;
/* `t` = the args of the clid_train_fwd_bwd / clid_train_decode call this step belongs to (may be NULL for a stand-alone
 * step on a complete `grad`).  t->defer_reduce: the reduction of that launch's per-block partials (their number is a
 * function of t) is folded into this launch.  t->touch_ws: only rows touched so far in this mapping() call are
 * visited -- exact, because a row never touched has g = m = v = 0 and receives a zero update (utils/tools.py:205-255
 * semantics unchanged; weight_decay != 0 disables the skip) -- and rows not touched by THIS iteration skip the gradient
 * read; t->cbuf: gradients and certainty increments come from the all-reduced compact buffer. */
int clid_train_adam(const clid_adam_args* a, const clid_train_args* t, void* stream);

/* Single-GPU Mapper.mapping loop in one call: iteration `it` uses index_base + it*index_stride (int64
 * elements), writes its losses to loss_base + 4*it and runs Adam step it+1 (state restarts per call). */
int clid_mapping_run(const clid_map_view* mv, const clid_train_args* t, const clid_adam_args* a,
                     int32_t iters, const int64_t* index_base, int64_t index_stride, float* loss_base,
                     void* stream);

/* The two halves of clid_train_fwd_bwd as separate calls (numerical-eikonal / no-eikonal modes).  Within one
 * Mapper.mapping call the positions, the voxel table and the drawn indices are fixed, so the neighbour searches
 * of all iterations do not depend on the training state: clid_train_search resolves n_iter iterations' batches
 * in ONE launch (iteration i draws from index_base + i*index_stride, int64 elements; t->index is ignored) and
 * parks each iteration's winners in rec_out (clid_train_search_floats(.., 1) floats per iteration);
 * clid_train_decode is clid_train_fwd_bwd for one iteration starting from its records. */
int64_t clid_train_search_floats(int32_t bs, int64_t batch_offset, int32_t decimation, int32_t eikonal_mode,
                                 int32_t n_iter);
/* wave tasks (records of 192 floats: 8 query slots each) of one iteration.  An iteration's block in rec_out is
 * [tasks x 192 floats of records | ceil(tasks / 2) x 128 words of tile number blocks]: the second part holds, per tile of
 * the matrix-core decode kernels (two consecutive tasks), the tile's (query, neighbour) pairs numbered per distinct map row
 * -- resolved by clid_train_search because it depends on the records only, not on the training state.  Written for
 * iterations of at most 2048 tiles (one tile per wave in the decode launch, whose length is then one tile's dependent
 * chain) on local maps of at most 2^17 points (probe table within one L2); otherwise the decode launch numbers its tiles
 * in place and the blocks stay unwritten (larger iterations do not even carry them). */
int32_t clid_train_search_tasks(int32_t bs, int64_t batch_offset, int32_t decimation, int32_t eikonal_mode);
int clid_train_search(const clid_map_view* mv, const clid_train_args* t, int32_t n_iter,
                      const int64_t* index_base, int64_t index_stride, float* rec_out, void* stream);
int clid_train_decode(const clid_map_view* mv, const clid_train_args* t, const float* rec, void* stream);
/* rows of per-block partials the forward/backward launch of `t` leaves in the workspace (a function of the arguments alone) */
int32_t clid_train_partial_rows(const clid_train_args* t);
/* config.consistency_loss_on (utils/mapper.py:770-776): value and gradients of weight_c * mean_j (1 - cos(g_main[near_index[j]],
 * g_near[j])) (F.cosine_similarity: each vector divided by max(norm, 1e-8)).  c_main [n_main][3] is zeroed and receives
 * dL/dg_main (several copies may name one sample), c_near [n_c][3] = dL/dg_near; loss_out[3] += the mean, loss_out[0] +=
 * weight_c * the mean. */
int clid_consistency_couple(const float* g_main, const float* g_near, const int64_t* near_index, int32_t n_c, int32_t n_main,
                            float weight_c, float* c_main, float* c_near, float* loss_out, void* stream);

/* ---- multi-GPU: RCCL inside the C ABI (new; the reference is single-GPU, slam.py:11) ---------------------------
 * One process per GPU.  Rank 0 creates a 128-byte id (clid_comm_unique_id), the host distributes it (e.g.
 * torch.distributed.broadcast), every rank calls clid_comm_init (collective: ncclCommInitRank on the current HIP
 * device).  clid_comm_allreduce is ncclAllReduce in place on `stream`: dtype 0 = float32, 1 = int32, 2 = uint8; op_max 0 =
 * SUM, 1 = MAX.  RCCL is resolved with dlopen at first use: the library loads without it, these calls then return an
 * error; clid_comm_available() tells (1 / 0) whether it resolves in this process, so that the ranks can agree BEFORE
 * anyone enters the collective clid_comm_init (a rank that cannot load RCCL would leave the others waiting in it). */
typedef struct clid_comm clid_comm;
int clid_comm_unique_id(uint8_t* id_out_host /* [128] */);
int clid_comm_init(const uint8_t* id_host /* [128] */, int32_t rank, int32_t world, clid_comm** comm_out);
int clid_comm_size(const clid_comm* comm); /* number of ranks (ncclCommCount), < 0 on error */
int clid_comm_available(void);
int clid_comm_allreduce(clid_comm* comm, void* buf, int64_t count, int32_t dtype, int32_t op_max, void* stream);
int clid_comm_destroy(clid_comm* comm);

/* ---- gradient exchange over peer-mapped buffers (csrc/p2p.hip; new -- the reference is single-GPU) --------------------
 * The per-iteration payload of the sharded loop is 0.1 - 2.5 MB: a ring all-reduce over P GPUs is bound by its 2 (P - 1)
 * latency steps there.  xGMI is all-to-all, so one launch per rank does: barrier | rank r sums slice r of every rank's
 * buffer in rank order into its own | barrier | copies the other slices from their owners -- two hops, bit-identical
 * sums on every rank.
 *   clid_p2p_create    allocates 2 exchange buffers of capacity_bytes (used alternately) + flag words on the current device
 *                      and writes this rank's export blob (clid_p2p_blob_bytes() bytes, HIP IPC handles);
 *   clid_p2p_connect   takes the blobs of ALL ranks in rank order (the host gathers them with its own mechanism, like the
 *                      RCCL id) and maps the peers' memory;
 *   clid_p2p_selftest  collective: exchanges of exactly representable patterns on both buffers, verified on the device;
 *                      0 = every element right and no wait timed out on this rank.  The caller agrees on the result over
 *                      the ranks (MIN) before handing the object to clid_train_args.p2p;
 *   clid_p2p_buffer    the buffer the NEXT clid_p2p_allreduce works on: fill it, exchange, read the sums from it;
 *   clid_p2p_allreduce in-place SUM of its first count_floats floats over the ranks (one launch on `stream`; collective:
 *                      every rank issues the same sequence of exchanges);
 *   clid_p2p_set_timeout  how long a flag wait polls before it gives up (default 600 s, the order of an RCCL watchdog);
 *   clid_p2p_status    CLID_OK unless a flag wait gave up ON THIS RANK (CLID_E_P2P_TIMEOUT: the sums since then are
 *                      undefined and the object must not be used further); synchronises `stream`;
 *   clid_p2p_agree     collective over the RCCL communicator of the same ranks: MAX of the error words, so every rank
 *                      gets CLID_E_P2P_TIMEOUT when any rank timed out (clid_mapping_run_dist ends with it: the host then
 *                      restores the state it saved before the call and repeats the call over RCCL -- Mapper.mapping).
 * At most 8 ranks, one node. */
typedef struct clid_p2p clid_p2p;
int64_t clid_p2p_blob_bytes(void);
int clid_p2p_create(int32_t rank, int32_t world, int64_t capacity_bytes, clid_p2p** out, uint8_t* blob_out_host);
int clid_p2p_connect(clid_p2p* p, const uint8_t* blobs_host /* [world][clid_p2p_blob_bytes()] */);
int clid_p2p_selftest(clid_p2p* p, void* stream);
int32_t clid_p2p_world(const clid_p2p* p);
int64_t clid_p2p_capacity(const clid_p2p* p); /* bytes of one exchange buffer */
void* clid_p2p_buffer(clid_p2p* p);
int clid_p2p_allreduce(clid_p2p* p, int64_t count_floats, void* stream);
/* bitwise OR over the ranks of `bytes` bytes at `buf` (any device memory; staged through the current exchange buffer), in
 * place: the MAX of 0 / 1 flag bytes -- the touched-row flags of a chunk in clid_mapping_run_dist. */
int clid_p2p_allreduce_or(clid_p2p* p, void* buf, int64_t bytes, void* stream);
int clid_p2p_status(clid_p2p* p, void* stream);
int clid_p2p_set_timeout(clid_p2p* p, double seconds);
int clid_p2p_agree(clid_p2p* p, clid_comm* comm, void* stream);
int clid_p2p_destroy(clid_p2p* p);
/* test aid: raise this rank's error word as a timed-out flag wait would */
int clid_debug_p2p_fail(clid_p2p* p, void* stream);
/* test aid: device-to-device copy on `stream` (the exchange buffers are not tensors of the host framework) */
int clid_debug_copy(void* dst, const void* src, int64_t bytes, void* stream);

/* Mapper.mapping on one rank of a data-parallel group in ONE host call: this rank's slice of
 * every batch (index_base = its first element of iteration 0, row stride index_stride; t->batch_offset, t->inv_n_main,
 * t->inv_n_eik carry the global lattice phase and normalisers), per iteration decode/backward -> RCCL all-reduce (SUM) on
 * `stream` -> the identical Adam step; afterwards the losses [iters][4] (SUM) and mv->ts_update [M] (MAX) are merged.
 * What is all-reduced per iteration:
 *   dense    (t->touch_ws or t->cbuf NULL) the fused gradient buffer `t->grad` [grad_floats = 848 + 16 (M + 1)];
 *   compact  (both given) [848 decoder gradients | 9 floats per map row THIS iteration touches]: once per chunk of
 *            iterations the touched-row flags (M bytes per iteration) are MAX-reduced over the ranks and the list lengths
 *            come back to the host (one synchronisation per chunk); see clid_train_args.touch_ws.  With t->p2p the
 *            compact buffer is summed over peer-mapped memory instead of by RCCL (the flags, losses and stamps still go
 *            through `comm`), and the call ends with clid_p2p_agree (one more synchronisation; CLID_E_P2P_TIMEOUT on
 *            every rank if a flag wait gave up on any).
 * With the tile decode kernels the certainty increments travel with the gradient rows; with kernel 0 (dense only) the
 * caller merges its certainty deltas itself.  exchanged_floats_host (may be NULL): 4-byte words this rank contributed to
 * all-reduces during the loop (payload accounting for the benches). */
int clid_mapping_run_dist(const clid_map_view* mv, const clid_train_args* t, const clid_adam_args* a, int32_t iters,
                          const int64_t* index_base, int64_t index_stride, float* loss_base, clid_comm* comm,
                          int64_t grad_floats, int64_t* exchanged_floats_host, void* stream);

/* the kernel (0 / 1 / 2, clid_train_args.decode_variant) clid_train_decode launches for these arguments: the tile kernels cover the
 * numerical / no-eikonal modes on 16-float accumulation rows, the rest runs on kernel 0.  With
 * kernels 1 / 2 the certainty increments travel in column 8 of the accumulation rows (merged by clid_train_adam,
 * and all-reduced with the gradients when world > 1); kernel 0 adds them to `cert` directly. */
int clid_train_decode_kernel(const clid_map_view* mv, const clid_train_args* t);

/* ---- sample + label generation in front of the path ("next" row N2) ---------------------------------
 * What the kernels read of `LocalPointCloudMap` (model/local_point_cloud_map.py:11-36): the reference's own
 * direct-mapped voxel table over the raw scan points, probed as is. */
typedef struct clid_cloud_view {
  const int64_t* buffer_pt_index; /* [buffer_size] raw-point index per voxel-hash slot, -1 = empty */
  const float* points;            /* [n_points][3] local_point_cloud_map (world frame) */
  const int32_t* neighbor_idx;    /* [P][3] cell offsets of set_search_neighborhood (:74-96), 7 by default */
  int64_t buffer_size;            /* config.local_buffer_size */
  int32_t n_points;
  int32_t P;
  float resolution;               /* config.local_voxel_size_m */
  float max_valid_range;          /* 1.732 * (num_nei_cells + 1) * resolution */
  float eta_threshold;            /* 0.2  (estimate_plane default, :157) */
  float dist_threshold;           /* 0.1 */
} clid_cloud_view;

/* The sampler's keys of utils/config.py + the sensor pose of the frame. */
typedef struct clid_sampler_params {
  float surface_sample_range_m, free_sample_begin_ratio, free_sample_end_dist_m, dist_weight_scale, max_range;
  int32_t surface_sample_n, free_front_n, free_behind_n, dist_weight_on, behind_dropoff_on;
  float pose[12];                 /* rows 0..2 of the sensor->world transform, row-major, cast to f32 */
  int32_t reserved[2];
} clid_sampler_params;

/* LocalPointCloudMap.region_specific_sdf_estimation (model/local_point_cloud_map.py:98-153) with
 * estimate_plane (:156-201): points [n][3] world frame -> |SDF| estimate [n], surface mask [n] (0/1). */
int clid_region_sdf(const clid_cloud_view* cloud, const float* points, int32_t n, float* sdf_abs_out,
                    uint8_t* surface_mask_out, void* stream);

/* DataSampler.sample (cloud != NULL, utils/data_sampler.py:260-402) / DataSampler.sample_pin geometry
 * (cloud == NULL, :16-258) in one launch.  points [n_rays][3] sensor frame; the random draws are inputs in
 * the reference's own layout and order (z_surface = randn [surface_sample_n * n_rays], u_front = rand
 * [free_front_n * n_rays], u_behind = rand [free_behind_n * n_rays], sample-major).  Outputs are DENSE and
 * ray-major, n_rays * (1 + surface_n + front_n + behind_n) rows: coord (sensor frame), sdf label, weight
 * (negative = free-space sample) and keep (0 = near-surface sample without a raw point around it; the
 * reference returns coord[keep], label[keep], weight[keep]). */
int clid_sample_frame(const clid_cloud_view* cloud, const clid_sampler_params* p, const float* points,
                      int32_t n_rays, const float* z_surface, const float* u_front, const float* u_behind,
                      float* coord_out, float* label_out, float* weight_out, uint8_t* keep_out, void* stream);

/* ---- map maintenance on the device ("next" row N4) ----------------------------------------------------
 * voxel_down_sample_torch (utils/tools.py:639-682): indices of one point per voxel -- the one closest to the
 * voxel centre, distance quantised to 1000 levels, lowest index among equals -- in ascending order of the
 * reference's linear voxel id (stride = max cell coordinate, aliasing included).  points [n][3];
 * workspace of clid_voxel_workspace_bytes(n) bytes; idx_out [n] int64 (capacity).  Returns the number of
 * voxels m >= 0 (idx_out[0..m) valid) or a negative error.  The output size is data dependent, so this call
 * synchronises `stream` once (the caller needs m to size its tensors). */
/* transform_torch (utils/tools.py:590-609): out[i] = R points[i] + t; pose12_host = rows 0..2 of the 4x4
 * transform, row-major, HOST memory (passed by value to the kernel). */
int clid_transform_points(const float* points, int32_t n, const float* pose12_host, float* out, void* stream);
int64_t clid_voxel_workspace_bytes(int32_t n);
int clid_voxel_down_sample(const float* points, int32_t n, float voxel_size, void* workspace, int64_t* idx_out,
                           void* stream);
/* The same in two halves: _launch enqueues everything (value may be NULL: distance to the voxel centre), _finish makes
 * the round trip (and orders with the library sort where the device-side ordering gave up) and returns m.  A caller with
 * other work to enqueue puts it between the two; nothing else may use `workspace` in between.  n_dev != NULL (device int64):
 * the number of points is *n_dev <= n, read on the device -- for points whose count the caller has not read back yet (the
 * k_vox_insert fallback inside _finish needs the same points and count still in place). */
int clid_voxel_down_sample_launch(const float* points, int32_t n, float voxel_size, const float* value, const int64_t* n_dev,
                                  void* workspace, int64_t* idx_out, void* stream);
int clid_voxel_down_sample_finish(int32_t n, void* workspace, int64_t* idx_out, void* stream);
/* The whole pass without a host round trip (1 <= n <= 2^21 points): count_out (device int64[2]) receives [number of occupied
 * voxels m | 1 if the voxel ids were too wide for the device-side ordering -- a bounding box beyond 2^17 voxels per axis; idx_out
 * is then unordered and the caller must treat the call as failed when it reads the block], idx_out[0..m) the indices in
 * ascending voxel id.  A bucket of the device-side ordering beyond its capacity is ranked by counting on the device (slow,
 * exact) instead of by the library sort of _finish, which needs m on the host. */
int clid_voxel_down_sample_async(const float* points, int32_t n, float voxel_size, const int64_t* n_dev, void* workspace,
                                 int64_t* idx_out, int64_t* count_out, void* stream);
/* voxel_down_sample_min_value_torch (utils/tools.py:685-724): as above, but the point of a voxel with the smallest
 * `value` [n] (>= 0; quantised to 1000 levels of its maximum, lowest index among equals) is taken -- the selection
 * NeuralPoints.recreate_hash makes with |ts - cur_ts| or (max certainty - certainty) (model/neural_points.py:864-882).
 * A maximum of 0 (the reference divides 0 / 0 there) selects the lowest index of every voxel. */
int clid_voxel_down_sample_min_value(const float* points, int32_t n, float voxel_size, const float* value, void* workspace,
                                     int64_t* idx_out, void* stream);

/* Table fill of NeuralPoints.recreate_hash (model/neural_points.py:858-860, 886-892 / 918-925): buffer_pt_index[:] = -1,
 * then buffer_pt_index[slot(points[src])] = src for p = 0..m-1 with src = idx[p] (idx == NULL: src = p); of several
 * entries naming one slot the last one stays (the reference's CPU result).  points [.][3]. */
int clid_map_rehash(const float* points, const int64_t* idx, int32_t m, float resolution, int64_t* buffer_pt_index,
                    int64_t buffer_size, void* stream);
/* Rows idx[0..m) of the global arrays into fresh ones (NeuralPoints.prune_map :795-808, the merging branch of
 * recreate_hash :898-913): points [.,3], orient [.,4], ts_create / ts_update i32, cert f32, feat [.,8] whose output row m
 * (the padding row) is input row `pad_src_row`. */
int clid_map_gather(const int64_t* idx, int32_t m, int64_t pad_src_row, const float* points, const float* orient,
                    const int32_t* ts_create, const int32_t* ts_update, const float* cert, const float* feat,
                    float* points_out, float* orient_out, int32_t* ts_create_out, int32_t* ts_update_out, float* cert_out,
                    float* feat_out, void* stream);
/* Selection of NeuralPoints.prune_map (model/neural_points.py:779-791): ascending indices of the points that STAY
 * (certainty >= thre, or -- unless global_prune -- still inside the travel-distance window) -> keep_idx_out [n capacity],
 * their number -> count_out[0] (device).  workspace: clid_map_prune_workspace_bytes(n). */
int64_t clid_map_prune_workspace_bytes(int64_t n);
int clid_map_prune_select(const int32_t* ts_update, const float* cert, int64_t n, const float* travel_dist, int32_t cur_ts,
                          float certainty_thre, float diff_travel_dist, int32_t global_prune, int64_t* keep_idx_out,
                          int64_t* count_out, void* workspace, void* stream);

/* Training-pool maintenance of Mapper.process_frame (utils/mapper.py:297-392) in one enqueue: the pool `a` (n_a samples)
 * followed by this frame's samples `b` (n_b) -- coord / global_coord [.,3] f32, sdf label, weight f32, time i32 -- are
 * filtered by ||global_coord - origin||^2 < radius2 (float64, like the reference under type promotion), when more than
 * `capacity` survive `kept - capacity` uniform picks with replacement are dropped (:352-361; generator = splitmix64 of
 * `seed`), and the survivors are compacted IN ORDER into the *_out arrays (capacity n_a + n_b rows; must not alias the
 * inputs).  counts_out [3] int64 (device): samples kept, how many of them come from `b`, scratch.  Nothing synchronises:
 * the caller reads counts_out when it needs the sizes.  workspace: clid_pool_workspace_bytes(n_a + n_b). */
int64_t clid_pool_workspace_bytes(int64_t n_total);
int clid_pool_filter(const float* coord_a, const float* gcoord_a, const float* label_a, const float* weight_a,
                     const int32_t* time_a, int64_t n_a, const float* coord_b, const float* gcoord_b, const float* label_b,
                     const float* weight_b, const int32_t* time_b, int64_t n_b, const double* origin_host, double radius2,
                     int64_t capacity, uint64_t seed, float* coord_out, float* gcoord_out, float* label_out,
                     float* weight_out, int32_t* time_out, int64_t* counts_out, void* workspace, const int64_t* n_b_dev,
                     void* stream); /* n_b_dev != NULL (device int64): only the first min(*n_b_dev, n_b) samples of `b` exist */
/* The same with a gate: scatter_after_event != NULL (a recorded hipEvent_t) holds the five-array compaction -- the one launch of
 * the frame that fills every wave slot of the chip -- back until the event; the flag / list / drop passes in front of it run at
 * once.  Mapper.process_frame records the event behind the map growth's voxel pass (small dependent launches that starve beside
 * the compaction). */
int clid_pool_filter_after(const float* coord_a, const float* gcoord_a, const float* label_a, const float* weight_a,
                           const int32_t* time_a, int64_t n_a, const float* coord_b, const float* gcoord_b, const float* label_b,
                           const float* weight_b, const int32_t* time_b, int64_t n_b, const double* origin_host, double radius2,
                           int64_t capacity, uint64_t seed, float* coord_out, float* gcoord_out, float* label_out,
                           float* weight_out, int32_t* time_out, int64_t* counts_out, void* workspace, const int64_t* n_b_dev,
                           void* stream, void* scatter_after_event);

/* LocalPointCloudMap.update_map (model/local_point_cloud_map.py:43-72) after the voxel down-sampling of the scan: the
 * `samples` whose voxel slot in table_old is still empty are appended to the map, the map is cropped to `map_size` around
 * the sensor (float64 when the caller's sensor position is) and table_new [buffer_size] int64 is rebuilt from scratch
 * (-1 fill, largest index per slot).  points_out: capacity n_map + n_samples rows, must not alias map_points.
 * counts_out [2] int64 (device): points in the new map, samples that found their slot empty.  Nothing synchronises. */
int64_t clid_cloud_workspace_bytes(int64_t n_total);
int clid_cloud_update(const float* map_points, int64_t n_map, const float* samples, int64_t n_samples, const int64_t* table_old,
                      int64_t* table_new, int64_t buffer_size, float resolution, const double* sensor_pos_host, double map_size,
                      int32_t pos_is_f64, float* points_out, int64_t* counts_out, void* workspace, const int64_t* sample_idx,
                      const int64_t* n_samples_dev, void* stream);
/* sample_idx / n_samples_dev != NULL (device; the idx_out / count_out of a clid_voxel_down_sample_async still in flight):
 * sample i is row sample_idx[i] of `samples`, i < *n_samples_dev <= n_samples (the bound the grids are sized for). */

/* The insert of NeuralPoints.update (model/neural_points.py:346-437) for the voxel-down-sampled `samples` [n][3]: slot of
 * each sample's voxel in buffer_pt_index, take test (empty | held point farther than sqrt(far_dist2) | held point stale
 * by `diff_travel` of travelled distance; test_on = 0 takes every sample: empty map / reboot frame), table update with the
 * reference's sequential semantics (the LAST sample naming a slot decides it), and the append of position, identity
 * orientation, stamps = cur_ts, certainty 0 at rows base + rank of the global arrays (which must have room for n more
 * rows).  count_out [1] int64 (device) = points added.  Features are the caller's (feature_std * randn). */
int64_t clid_map_insert_workspace_bytes(int32_t n);
int clid_map_insert(const float* samples, int32_t n, int64_t* buffer_pt_index, int64_t buffer_size, float resolution,
                    float* neural_points, float* point_orientations, int32_t* ts_create, int32_t* ts_update, float* certainties,
                    int64_t base, const float* travel_dist, int32_t cur_ts, int32_t test_on, int32_t temporal, float far_dist2,
                    float diff_travel, int64_t* count_out, void* workspace, const int64_t* sample_idx, const int64_t* n_dev,
                    float* features_zero, void* stream);
/* sample_idx / n_dev != NULL (device): sample i is row sample_idx[i] of `samples`, i < *n_dev <= n -- the voxel down-sampling
 * in front of the insert is then still in flight (clid_voxel_down_sample_async) and nothing waits for it on the host.
 * features_zero != NULL ([rows + 1][8] floats, the global feature array): the rows of the added points and the padding row
 * behind the last of them are zeroed (geo_feature_std == 0) -- the caller cannot size a fill without the count. */

/* NeuralPoints.reset_local_map (model/neural_points.py:439-536): the local window = points whose creation (or mid)
 * stamp lies within `diff_travel` of travelled distance (or `diff_ts_local` frames) of cur_ts -- dropped when it holds
 * fewer than 100 points, restricted to stamps >= reboot_ts when reboot_map -- AND within sqrt(radius2) of the sensor
 * (pos_is_f64: the caller's sensor position tensor is float64, so the reference's test runs in float64).
 * Outputs (capacity n rows, n + 1 where noted): local_ids [n] int64 ascending global indices, global2local [n + 1] int64
 * (-1 outside, the padding element -1), local_mask [n + 1] bool (padding element true), and the gathered local arrays
 * (points [.,3], orientations [.,4], certainties, update stamps, features [(m + 1), F] with the global padding row last).
 * counts_out [2] int64 (device): points inside the time window, m = local points.  Nothing synchronises.
 * n_extra_dev != NULL (device int64, e.g. the count_out of a clid_map_insert still in flight): the map holds n + *n_extra_dev
 * points, at most n_upper -- grids, the scan, the workspace (clid_local_window_workspace_bytes(n_upper)) and the output
 * capacities then cover n_upper rows and the kernels read the exact size on the device, so the insert and the window
 * selection of a frame share ONE read-back.
 * local_capacity (< 0: n_upper): rows the LOCAL output arrays hold (local_ids, points, orientations, certainties, stamps;
 * features local_capacity + 1) -- a caller that knows the local map is much smaller than the map sizes them from its last m;
 * rows beyond the capacity are not written, counts_out[1] still reports m, and the caller repeats the call when m exceeds it
 * (global2local / local_mask are always n_upper + 1). */
int64_t clid_local_window_workspace_bytes(int64_t n);
int clid_local_window(const float* neural_points, const int32_t* ts_create, const int32_t* ts_update, const float* travel_dist,
                      int64_t n, int32_t cur_ts, int32_t use_mid_ts, int32_t temporal, int32_t use_travel_dist,
                      float diff_travel, int32_t diff_ts_local, int32_t reboot_ts, int32_t reboot_map,
                      const double* sensor_pos_host, double radius2, int32_t pos_is_f64, const float* point_orientations,
                      const float* point_certainties, const float* geo_features, int64_t* local_ids_out,
                      int64_t* global2local_out, uint8_t* local_mask_out, float* local_points_out, float* local_orient_out,
                      float* local_cert_out, int32_t* local_ts_out, float* local_feat_out, int64_t* counts_out,
                      void* workspace, const int64_t* n_extra_dev, int64_t n_upper, int64_t local_capacity, void* stream);

/* NeuralPoints.assign_local_to_global (model/neural_points.py:538-549) in one launch: local features (n + 1 rows of
 * F, the last one the padding row -> global row pad_row), certainties and update stamps back to the global arrays
 * at ids [n] int64 (= nonzero(local_mask[:-1]), ascending). */
int clid_local_to_global(const int64_t* ids, int32_t n, int64_t pad_row, const float* local_feat,
                         const float* local_cert, const int32_t* local_ts, float* global_feat, float* global_cert,
                         int32_t* global_ts, void* stream);

/* Sampler output -> this frame's pool rows and the points that grow the map (utils/mapper.py:240-283, :297-310) in one
 * flag pass + one scan + one scatter.  Inputs: the [n] rows clid_sample_frame wrote (coord [n,3] sensor frame, label,
 * weight, keep mask), the pose (12 floats, row-major 3x4, host), near_range = surface_sample_range_m * map_surface_ratio,
 * the frame stamp.  Outputs (capacity n rows each, stable order): rows with keep != 0 -> coord_out, gcoord_out (= pose *
 * coord, transform_torch arithmetic), label_out, weight_out, stamp_out; of those, rows with |label| < near_range ->
 * update_out (world frame).  counts_out (device, 2 x int64) = {kept, near}. */
int64_t clid_sample_compact_workspace_bytes(int64_t n);
int clid_sample_compact(const float* coord, const float* label, const float* weight, const uint8_t* keep, int64_t n,
                        const float* pose12_host, float near_range, int32_t stamp, float* coord_out, float* gcoord_out,
                        float* label_out, float* weight_out, int32_t* stamp_out, float* update_out, int64_t* counts_out,
                        void* workspace, void* stream);

/* Newly observed samples of a frame (utils/mapper.py:400-423): certainty of the sample's probe cells in the GLOBAL map
 * (NeuralPoints.query_certainty with the stencil `delta` [P], model/neural_points.py:1032-1051) < certainty_thre and
 * |sdf_label| < label_max; idx_out = index_offset + position of the selected samples, ascending; count_out (device, 1 x
 * int64).  pool_counts != NULL (device, the counts_out of clid_pool_filter still in flight): x / sdf_label are the whole
 * pool arrays, the frame's samples are their last pool_counts[1] of pool_counts[0] rows, n is an upper bound of that
 * number and index_offset is ignored -- the selection then needs no read-back between the two calls. */
int64_t clid_new_sample_workspace_bytes(int64_t n);
int clid_new_sample_select(const int64_t* buffer_pt_index, int64_t buffer_size, const float* neural_points,
                           const float* point_certainties, const int32_t* delta, int32_t P, float resolution,
                           float max_valid_dist2, const float* x, const float* sdf_label, int64_t n, float certainty_thre,
                           float label_max, int64_t index_offset, int64_t* idx_out, int64_t* count_out,
                           const int64_t* pool_counts, void* workspace, void* stream);

/* Per-call preparation of Mapper.mapping in one launch: zero `zero_floats` floats at `zero_base` (16-byte aligned,
 * multiple of 4: the fused gradient buffer, the Adam state and the loss rows restart every call, utils/mapper.py:634) and
 * draw the batches of all `iters` iterations, index_out [iters][bs] int64, composed as utils/mapper.py:473-500:
 * bs - bs_new uniform rows of the pool [0, pool_count) followed by bs_new uniform picks from new_idx [n_new] (bs_new may
 * be 0).  The generator is counter-based on (seed, counter, position): identical on every rank.  index_out may be NULL
 * (reset only).
 * sort_workspace != NULL (clid_mapping_prep_workspace_bytes; needs pool_coord [pool_count, 3] world frame and the voxel
 * size `resolution`): every 16 384-sample segment of every iteration's batch -- the same draws -- is written in Morton
 * order of the samples' voxels by a second launch (three stable 8-bit counting passes in LDS on (Morton code, draw
 * position): a function of the draws alone, identical on every rank).  The loss and gradient sums do not depend on the
 * order; neighbouring queries then share neural points, which is what the kernels' per-tile row merging and the caches
 * feed on.  `decimation` > 1 (config.gradient_decimation while the numerical eikonal term is on): the order is applied WITHIN
 * two classes -- the positions col with col % decimation == 0 of a batch (its eikonal subset coord[::decimation],
 * utils/mapper.py:700-704) receive the draws of those positions, the other positions the other draws, each class in (Morton
 * code, draw position) order -- so the ordered batch has the SAME eikonal subset as the draws in their own order, sample for
 * sample (a uniform random tenth, not a spatially stratified one); 1 = one class.  clid_debug_prep_draw is the host
 * restatement of one uniform draw (tests).
 * [col0, col0 + ncols) (ncols <= 0: all of them) = the columns of every iteration's batch this rank needs (its shard of
 * a sharded run): only they are guaranteed to be written -- widened to whole 16 384-sample segments when ordering, because
 * a segment is ordered as a unit -- so a rank of a data-parallel group pays for its slice, not for the global batch. */
int64_t clid_mapping_prep_workspace_bytes(int32_t iters, int32_t bs);
int clid_mapping_prep(float* zero_base, int64_t zero_floats, int64_t* index_out, int32_t iters, int32_t bs, int32_t bs_new,
                      int64_t pool_count, const int64_t* new_idx, int64_t n_new, uint64_t seed, uint64_t counter,
                      const float* pool_coord, float resolution, void* sort_workspace, int32_t col0, int32_t ncols,
                      int32_t decimation, void* stream);
int64_t clid_debug_prep_draw(uint64_t seed, uint64_t counter, uint64_t e, uint64_t range);
/* Test entry: the exclusive prefix sum every compaction of the map / pool maintenance runs on (hand-written: tiles of 4096
 * elements per 1024-thread block; one launch up to 4 096 elements, two beyond), out[i] = in[0] + .. + in[i - 1], in != out;
 * elem_bytes 4 (int32) or 8 (uint64); scratch of clid_debug_scan_scratch_bytes(n) bytes. */
int64_t clid_debug_scan_scratch_bytes(int64_t n);
int clid_debug_scan(const void* in, void* out, int64_t n, int32_t elem_bytes, void* scratch, void* stream);

/* ---- measurement aid (bench.py roofline leg; not part of the reference's interface) -----------
 * A profiler object handed in through clid_train_args.prof: the training entry points then issue each kernel through
 * hipExtLaunchKernelGGL with start / stop events (the dispatch's own begin / end time stamps, the clock rocprofv3
 * --kernel-trace reads).  clid_profile_read synchronises the device and returns summed elapsed ms per kernel:
 * out[0] = fused (or decode) kernel, out[1] = search kernel of the hoisted-search loop, out[2] = partial reduce (+ pack),
 * out[3] = adam, out[4] = empty event-pair overhead (ms), out[5] = touched-row scan; *iters_host = decode launches
 * recorded; the recorded spans are dropped.  One object per host thread that launches. */
typedef struct clid_prof clid_prof;
clid_prof* clid_profile_create(void);
int clid_profile_read(clid_prof* prof, double* out_host /* [6] */, int* iters_host, void* stream);
void clid_profile_destroy(clid_prof* prof);

/* CPU-only test aid: enumerate the fused kernel's task -> query mapping (see csrc/train.hip). */
int clid_debug_task_cover(int32_t bs, int64_t batch_offset, int32_t decimation, int32_t eikonal_mode,
                          int32_t* main_count_host, int32_t* fd_count_host, int32_t* n_tasks_host);

#ifdef __cplusplus
}
#endif
#endif /* CLID_NATIVE_H */
