// Per-frame sample + label generation ("next" row N2 of SURVEY.md section 8f): the step that fills the
// training pool in front of the hot path.
//
//   k_region_sdf     LocalPointCloudMap.region_specific_sdf_estimation (model/local_point_cloud_map.py:98-153)
//                    + estimate_plane (:156-201): per sample, P (= 7) probes of the raw-point voxel table, the
//                    4 nearest raw points, a total-least-squares plane through them, and |SDF| = distance to
//                    that plane when it is trustworthy, else the distance to the nearest raw point.
//   k_sample_frame   DataSampler.sample / sample_pin (utils/data_sampler.py:16-402): ONE launch replaces the
//                    reference's ~60 elementwise torch ops per frame: each thread produces one (ray, sample)
//                    pair directly in the ray-major output order, near-surface samples run the region
//                    estimate on their world position in the same thread.
//
// The reference's own table (`buffer_pt_index`, int64, direct-mapped by the voxel hash) is probed as is: it
// is rebuilt by every update_map and read once per frame, so a compact mirror would cost more than it saves.
// HBM-bound integer/gather work: no LDS, no MFMA; all probe loads of a thread are issued before use.
#include "common.hpp"

namespace clid {

constexpr long long kCloudPrime0 = 73856093LL, kCloudPrime1 = 19349663LL, kCloudPrime2 = 83492791LL;  // lpcm.py:27-29
constexpr int kCloudChunk = 8;  // probes resolved per batch of loads

struct CloudView {
  const long long* table;
  const float* pts;
  const int* nb;
  long long B;
  double inv_B;  // 1 / B: the slot arithmetic below
  int n_pts, P;
  float res, max_range, eta, thr;
};

__device__ __forceinline__ float dot4(const float* a, const float* b) {
  return fmaf(a[3], b[3], fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0])));
}

// One-sided (Hestenes) Jacobi SVD of the 4 x 3 matrix of centred points held as 3 columns: after the sweeps the
// columns are orthogonal, their norms are the singular values and V's columns the right singular vectors.
// Works on the matrix itself (not on A^T A), so small singular values keep their relative accuracy in fp32.
__device__ __forceinline__ void jacobi_4x3(float a[3][4], float v[3][3]) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) v[i][j] = (i == j) ? 1.f : 0.f;
#pragma unroll 1
  for (int sweep = 0; sweep < 6; ++sweep) {
#pragma unroll
    for (int pq = 0; pq < 3; ++pq) {
      const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
      const float alpha = dot4(a[p], a[p]), beta = dot4(a[q], a[q]), gamma = dot4(a[p], a[q]);
      if (gamma * gamma > 1e-14f * alpha * beta) {
        const float zeta = (beta - alpha) / (2.f * gamma);
        const float t = copysignf(1.f, zeta) / (fabsf(zeta) + sqrtf(1.f + zeta * zeta));
        const float c = 1.f / sqrtf(1.f + t * t), s = c * t;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float ap = a[p][i], aq = a[q][i];
          a[p][i] = c * ap - s * aq;
          a[q][i] = s * ap + c * aq;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const float vp = v[p][i], vq = v[q][i];
          v[p][i] = c * vp - s * vq;
          v[q][i] = s * vp + c * vq;
        }
      }
    }
  }
}

// h mod B, non-negative -- what `fmod` with the sign of h followed by python-style indexing of a negative slot amounts to
// (lpcm.py:38-41, 112-116).  A 64-bit remainder by a run-time divisor is a ~100-instruction software routine and every
// sample takes P = 7 of them: for |h| < 2^52 (cells within +-1e6 of the origin) the quotient comes from one fp64 multiply,
// off by at most one, and the remainder is corrected in integers -- exact.
__device__ __forceinline__ long long cloud_slot_of(long long h, long long B, double inv_B) {
  if (h > -(1LL << 52) && h < (1LL << 52)) {
    const long long q = (long long)floor((double)h * inv_B);
    long long r = h - q * B;
    if (r < 0) r += B;
    if (r >= B) r -= B;
    return r;
  }
  long long s = h % B;
  return s < 0 ? s + B : s;
}

// |SDF| estimate of one world-frame sample; returns the surface flag (at least one raw point around it)
__device__ __forceinline__ bool region_estimate(const CloudView& cv, float x, float y, float z, float* sdf_abs) {
  const long long cx = (long long)floorf(fdiv(x, cv.res)), cy = (long long)floorf(fdiv(y, cv.res)),
                  cz = (long long)floorf(fdiv(z, cv.res));
  float bd[4], bx[4], by[4], bz[4];
  const int last = cv.n_pts > 0 ? cv.n_pts - 1 : 0;
  // torch.topk over a row of "no neighbour" distances returns some of the -1 probes, i.e. the LAST raw point
  // (python index -1); their coordinates only matter when all 4 are real, so any consistent filler works
#pragma unroll
  for (int k = 0; k < 4; ++k) { bd[k] = cv.max_range; bx[k] = by[k] = bz[k] = 0.f; }
  for (int t0 = 0; t0 < cv.P; t0 += kCloudChunk) {
    long long id[kCloudChunk];
#pragma unroll
    for (int t = 0; t < kCloudChunk; ++t) {
      const int o = t0 + t;
      const bool in = o < cv.P;
      const int oo = in ? o : 0;
      const long long h = (cx + cv.nb[3 * oo]) * kCloudPrime0 + (cy + cv.nb[3 * oo + 1]) * kCloudPrime1 +
                          (cz + cv.nb[3 * oo + 2]) * kCloudPrime2;
      const long long s = cloud_slot_of(h, cv.B, cv.inv_B);  // fmod keeps the sign; the table is then indexed python-style (lpcm.py:38-41,112-116)
      id[t] = in ? cv.table[s] : -1;
    }
    float px[kCloudChunk], py[kCloudChunk], pz[kCloudChunk];
#pragma unroll
    for (int t = 0; t < kCloudChunk; ++t) {
      const long long src = id[t] >= 0 ? id[t] : last;
      px[t] = cv.pts[src * 3 + 0]; py[t] = cv.pts[src * 3 + 1]; pz[t] = cv.pts[src * 3 + 2];
    }
#pragma unroll
    for (int t = 0; t < kCloudChunk; ++t) {
      if (id[t] < 0) continue;
      const float dx = fsub(px[t], x), dy = fsub(py[t], y), dz = fsub(pz[t], z);
      float d = sqrtf(fadd(fadd(fmul(dx, dx), fmul(dy, dy)), fmul(dz, dz)));
      float qx = px[t], qy = py[t], qz = pz[t];
#pragma unroll
      for (int k = 0; k < 4; ++k) {  // sorted insert, earlier probe wins ties
        if (d < bd[k]) {
          const float td = bd[k], tx = bx[k], ty = by[k], tz = bz[k];
          bd[k] = d; bx[k] = qx; by[k] = qy; bz[k] = qz;
          d = td; qx = tx; qy = ty; qz = tz;
        }
      }
    }
  }
  const bool surface = bd[0] < cv.max_range;
  float out = bd[0];
  if (bd[3] < cv.max_range) {  // four real neighbours: try the plane (lpcm.py:127-146)
    const float mx = ((bx[0] + bx[1]) + (bx[2] + bx[3])) * 0.25f, my = ((by[0] + by[1]) + (by[2] + by[3])) * 0.25f,
                mz = ((bz[0] + bz[1]) + (bz[2] + bz[3])) * 0.25f;
    float a[3][4], v[3][3];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[0][i] = bx[i] - mx; a[1][i] = by[i] - my; a[2][i] = bz[i] - mz; }
    jacobi_4x3(a, v);
    float sv[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) sv[j] = sqrtf(dot4(a[j], a[j]));
    int jm = 0;
    if (sv[1] < sv[jm]) jm = 1;
    if (sv[2] < sv[jm]) jm = 2;
    const float s_min = sv[jm];
    const float o1 = sv[(jm + 1) % 3], o2 = sv[(jm + 2) % 3];
    const float s_mid = fminf(o1, o2);
    if (s_min / (s_mid + 1e-6f) <= cv.eta) {  // lpcm.py:172-178
      float nx = v[0][0], ny = v[0][1], nz = v[0][2];
      if (jm == 1) { nx = v[1][0]; ny = v[1][1]; nz = v[1][2]; }
      if (jm == 2) { nx = v[2][0]; ny = v[2][1]; nz = v[2][2]; }
      const float off = -1.0f * fadd(fadd(fmul(nx, mx), fmul(ny, my)), fmul(nz, mz));
      float worst = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        worst = fmaxf(worst, fabsf(fadd(fadd(fadd(fmul(bx[i], nx), fmul(by[i], ny)), fmul(bz[i], nz)), off)));
      if (worst <= cv.thr) out = fabsf(fadd(fadd(fadd(fmul(nx, x), fmul(ny, y)), fmul(nz, z)), off));  // :147-149
    }
  }
  *sdf_abs = surface ? out : cv.max_range;
  return surface;
}

__global__ void __launch_bounds__(256) k_region_sdf(CloudView cv, const float* __restrict__ pts, int n,
                                                    float* __restrict__ sdf_abs, unsigned char* __restrict__ mask) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float d;
  const bool s = region_estimate(cv, pts[i * 3 + 0], pts[i * 3 + 1], pts[i * 3 + 2], &d);
  sdf_abs[i] = d;
  mask[i] = s ? 1 : 0;
}

struct SampleParams {
  float sigma, begin_ratio, end_dist, w_scale, max_range;
  int ns, nf, nb, dist_weight_on, dropoff_on, region;
  float T[12];
};

// thread = (ray, sample slot k): k = 0 exact hit | 1..ns near-surface | then nf in front | then nb behind, in the output's
// ray-major order.  (Measured and dropped: the near-surface pairs first, slot-major, so that no lane of a wave sits out the
// region estimate -- 85 instead of 80 us per 9e5-sample launch: the outputs then go out as scattered partial lines.)
__global__ void __launch_bounds__(256) k_sample_frame(CloudView cv, SampleParams sp, const float* __restrict__ pts,
                                                      int n_rays, const float* __restrict__ z_s,
                                                      const float* __restrict__ u_f, const float* __restrict__ u_b,
                                                      float* __restrict__ coord, float* __restrict__ label,
                                                      float* __restrict__ weight, unsigned char* __restrict__ keep) {
  const int n_all = 1 + sp.ns + sp.nf + sp.nb;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)n_rays * n_all) return;
  const int ray = (int)(t / n_all), k = (int)(t - (long long)ray * n_all);
  const float x = pts[ray * 3 + 0], y = pts[ray * 3 + 1], z = pts[ray * 3 + 2];
  const float dist = sqrtf(fadd(fadd(fmul(x, x), fmul(y, y)), fmul(z, z)));  // ds.py:37-39
  const float two_sigma = 2.0f * sp.sigma;
  float disp, ratio;
  bool surface_type = true;
  if (k == 0) {  // the measured point itself
    disp = 0.f;
    ratio = 1.f;
  } else if (k <= sp.ns) {  // ds.py:46-54
    disp = fmul(z_s[(size_t)(k - 1) * n_rays + ray], sp.sigma);
    ratio = fadd(fdiv(disp, dist), 1.0f);
  } else if (k <= sp.ns + sp.nf) {  // ds.py:69-79
    const float hi = fsub(1.0f, fdiv(two_sigma, dist));
    ratio = fadd(fmul(u_f[(size_t)(k - 1 - sp.ns) * n_rays + ray], fsub(hi, sp.begin_ratio)), sp.begin_ratio);
    disp = fmul(fsub(ratio, 1.0f), dist);
    surface_type = false;
  } else {  // ds.py:89-103
    const float hi = fadd(fdiv(sp.end_dist, dist), 1.0f);
    const float lo = fadd(1.0f, fdiv(two_sigma, dist));
    ratio = fadd(fmul(u_b[(size_t)(k - 1 - sp.ns - sp.nf) * n_rays + ray], fsub(hi, lo)), lo);
    disp = fmul(fsub(ratio, 1.0f), dist);
    surface_type = false;
  }
  const float sx = fmul(x, ratio), sy = fmul(y, ratio), sz = fmul(z, ratio);
  float lab = -disp;  // projective label: behind the surface negative (ds.py:172-175, 214 / :351)
  bool kp = true;
  if (sp.region && k >= 1 && k <= sp.ns) {  // ds.py:343-356: region-specific |SDF| of the WORLD position
    const float wx = fmaf(sz, sp.T[2], fmaf(sy, sp.T[1], fmaf(sx, sp.T[0], sp.T[3])));
    const float wy = fmaf(sz, sp.T[6], fmaf(sy, sp.T[5], fmaf(sx, sp.T[4], sp.T[7])));
    const float wz = fmaf(sz, sp.T[10], fmaf(sy, sp.T[9], fmaf(sx, sp.T[8], sp.T[11])));
    float d;
    kp = region_estimate(cv, wx, wy, wz, &d);
    lab = disp < 0.f ? d : -d;
  }
  float w = 1.0f;
  if (surface_type && sp.dist_weight_on)  // ds.py:143-152: far surface samples weigh less, [0.6, 1.4]
    w = fsub(1.0f + sp.w_scale * 0.5f, fmul(fdiv(dist, sp.max_range), sp.w_scale));
  if (sp.dropoff_on && !sp.region) {  // ds.py:154-164 (sample_pin only)
    const float hi = sp.end_dist, lo = 0.2f * sp.end_dist;
    float dw = fdiv(fsub(hi, disp), hi - lo);
    dw = fminf(fmaxf(dw, 0.f), 1.f);
    w = fmul(w, fadd(fmul(dw, 0.8f), 0.2f));
  }
  if (!surface_type) w = -w;  // the sign of the weight flags free-space samples (ds.py:167)
  coord[t * 3 + 0] = sx; coord[t * 3 + 1] = sy; coord[t * 3 + 2] = sz;
  label[t] = lab;
  weight[t] = w;
  keep[t] = kp ? 1 : 0;
}

}  // namespace clid

using namespace clid;

static int cloud_view_from(const clid_cloud_view* c, CloudView* out, const char* who) {
  if (!c || !c->buffer_pt_index || !c->neighbor_idx || (c->n_points > 0 && !c->points) || c->buffer_size <= 0 ||
      c->P <= 0 || c->n_points < 0 || !(c->resolution > 0.f)) {
    clid_set_error("%s: bad raw-point map view", who);
    return CLID_E_ARG;
  }
  out->table = reinterpret_cast<const long long*>(c->buffer_pt_index);
  out->pts = c->points;
  out->nb = c->neighbor_idx;
  out->B = c->buffer_size;
  out->inv_B = 1.0 / (double)c->buffer_size;
  out->n_pts = c->n_points;
  out->P = c->P;
  out->res = c->resolution;
  out->max_range = c->max_valid_range;
  out->eta = c->eta_threshold;
  out->thr = c->dist_threshold;
  return CLID_OK;
}

extern "C" int clid_region_sdf(const clid_cloud_view* cloud, const float* points, int32_t n, float* sdf_abs_out,
                               uint8_t* surface_mask_out, void* stream) {
  CloudView cv;
  if (int e = cloud_view_from(cloud, &cv, "clid_region_sdf")) return e;
  if (n < 0 || (n > 0 && (!points || !sdf_abs_out || !surface_mask_out))) {
    clid_set_error("clid_region_sdf: bad argument");
    return CLID_E_ARG;
  }
  if (n == 0) return CLID_OK;
  if (cv.n_pts == 0) {
    clid_set_error("clid_region_sdf: the raw-point map is empty");
    return CLID_E_ARG;
  }
  hipLaunchKernelGGL(k_region_sdf, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, cv, points, n,
                     sdf_abs_out, surface_mask_out);
  CLID_CHECK_LAUNCH();
  return CLID_OK;
}

extern "C" int clid_sample_frame(const clid_cloud_view* cloud, const clid_sampler_params* p, const float* points,
                                 int32_t n_rays, const float* z_surface, const float* u_front, const float* u_behind,
                                 float* coord_out, float* label_out, float* weight_out, uint8_t* keep_out,
                                 void* stream) {
  if (!p || n_rays < 0 || p->surface_sample_n < 0 || p->free_front_n < 0 || p->free_behind_n < 0) {
    clid_set_error("clid_sample_frame: bad argument");
    return CLID_E_ARG;
  }
  if (n_rays == 0) return CLID_OK;
  if (!points || !coord_out || !label_out || !weight_out || !keep_out || (p->surface_sample_n > 0 && !z_surface) ||
      (p->free_front_n > 0 && !u_front) || (p->free_behind_n > 0 && !u_behind)) {
    clid_set_error("clid_sample_frame: null argument");
    return CLID_E_ARG;
  }
  CloudView cv = {};
  SampleParams sp;
  sp.region = cloud ? 1 : 0;
  if (cloud) {
    if (int e = cloud_view_from(cloud, &cv, "clid_sample_frame")) return e;
    if (cv.n_pts == 0) {
      clid_set_error("clid_sample_frame: the raw-point map is empty");
      return CLID_E_ARG;
    }
  }
  sp.sigma = p->surface_sample_range_m;
  sp.begin_ratio = p->free_sample_begin_ratio;
  sp.end_dist = p->free_sample_end_dist_m;
  sp.w_scale = p->dist_weight_scale;
  sp.max_range = p->max_range;
  sp.ns = p->surface_sample_n;
  sp.nf = p->free_front_n;
  sp.nb = p->free_behind_n;
  sp.dist_weight_on = p->dist_weight_on;
  sp.dropoff_on = p->behind_dropoff_on;
  for (int i = 0; i < 12; ++i) sp.T[i] = p->pose[i];
  const long long total = (long long)n_rays * (1 + sp.ns + sp.nf + sp.nb);
  hipLaunchKernelGGL(k_sample_frame, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, cv, sp,
                     points, n_rays, z_surface, u_front, u_behind, coord_out, label_out, weight_out, keep_out);
  CLID_CHECK_LAUNCH();
  return CLID_OK;
}
