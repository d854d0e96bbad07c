// Weight operands of MANY convolutions in ONE launch (include/cfun_hip.h: cfun_weight_prepare).
//
// Every conv of a training step needs its OIDHW weight (the checkpoint layout, backbone.py:14-23, mask_branch.py:23-89)
// in the layout its kernels read: packed [tap][Ci][CoP] (forward) and [tap][Co][CiP] (data gradient), the Winograd-
// transformed U of k_conv_wino (1-D / 2-D, mirrored taps for the data gradient) or the parity-folded 2x2x2 weights of a
// stride-2 conv's data gradient.  Done per conv that is 2 - 3 launches of a few microseconds each -- ~130 of a step's ~1250
// launches (k_transpose_pad2 78, k_wino2_weights 44, k_wino_weights 2, k_fold_s2_dgrad_weights 4; profiles/round4_*).  Here
// a job table describes all of them and one grid does the lot: a workgroup owns a 16 (co) x 16 (ci) block of one job's
// weight with all its taps (T <= 27) in LDS and writes every operand of that block.  Jobs may gather rows / columns of
// the source (the per-RoI Dropout3d slices of mask_branch._dropout_pair: no index_select launches either).
// The arithmetic of the transforms is the one of k_wino_weights / k_wino2_weights (conv3d_wino.hip): same bits.
#include "common.h"

namespace {

constexpr int kBlock = 256;
constexpr int TS = 28;      // LDS floats per (co, ci) pair: T <= 27 taps

__device__ __forceinline__ float4 wino_row(float g0, float g1, float g2) {
  return make_float4(g0, 0.5f * ((g0 + g2) + g1), 0.5f * ((g0 + g2) - g1), g2);
}

__global__ void __launch_bounds__(kBlock)
k_weight_prepare(const CfunWeightJob* __restrict__ jobs, int njobs) {
  __shared__ float s[16 * 16 * TS];      // [co][ci][tap]
  __shared__ int sj;
  const int tid = threadIdx.x;
  if (tid == 0) {
    int j = 0;
    while (j + 1 < njobs && jobs[j + 1].block_begin <= blockIdx.x) ++j;
    sj = j;
  }
  __syncthreads();
  const CfunWeightJob J = jobs[sj];
  const int local = (int)(blockIdx.x - J.block_begin);
  const int tiles_ci = (J.Ci + 15) >> 4;
  const int co0 = (local / tiles_ci) * 16, ci0 = (local % tiles_ci) * 16;
  const int T = J.T, Co = J.Co, Ci = J.Ci;
  const int CoP = (Co + 15) & ~15, CiP = (Ci + 15) & ~15;

  // ---- the block's [16 co][16 ci][T] piece of the source (rows / columns gathered through the index lists)
  for (int i = tid; i < 16 * 16 * T; i += kBlock) {
    const int co = i / (16 * T), r = i - co * 16 * T, ci = r / T, t = r - ci * T;
    const int gco = co0 + co, gci = ci0 + ci;
    float v = 0.f;
    if (gco < Co && gci < Ci) {
      const int64_t sco = J.co_idx ? J.co_idx[gco] : gco, sci = J.ci_idx ? J.ci_idx[gci] : gci;
      v = J.w[(sco * J.src_ci + sci) * T + t];
    }
    s[(co * 16 + ci) * TS + t] = v;
  }
  __syncthreads();

  // ---- forward operand
  if (J.fwd_kind == CFUN_WOP_PACK) {            // wp[t][ci][CoP]
    for (int i = tid; i < 256 * T; i += kBlock) {
      const int t = i >> 8, ci = (i >> 4) & 15, co = i & 15;
      if (ci0 + ci < Ci) J.fwd[((int64_t)t * Ci + ci0 + ci) * CoP + co0 + co] = s[(co * 16 + ci) * TS + t];
    }
  } else if (J.fwd_kind == CFUN_WOP_WINO1) {    // u[r9][ci][CoP] x 4 points
    float4* u = reinterpret_cast<float4*>(J.fwd);
    for (int i = tid; i < 256 * 9; i += kBlock) {
      const int r9 = i >> 8, ci = (i >> 4) & 15, co = i & 15;
      const float* g = s + (co * 16 + ci) * TS + r9 * 3;
      if (ci0 + ci < Ci) u[((int64_t)r9 * Ci + ci0 + ci) * CoP + co0 + co] = wino_row(g[0], g[1], g[2]);
    }
  } else if (J.fwd_kind == CFUN_WOP_WINO2) {    // u[(dz*4 + py)][ci][CoP] x 4 points
    float4* u = reinterpret_cast<float4*>(J.fwd);
    for (int i = tid; i < 256 * 3; i += kBlock) {
      const int dz = i >> 8, ci = (i >> 4) & 15, co = i & 15;
      if (ci0 + ci >= Ci) continue;
      const float* g = s + (co * 16 + ci) * TS + dz * 9;      // [ky][kx]
      float4 t[3];                                            // per kx: the 4 y-points
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) t[kx] = wino_row(g[kx], g[3 + kx], g[6 + kx]);
      const int64_t base = ((int64_t)(dz * 4) * Ci + ci0 + ci) * CoP + co0 + co, step = (int64_t)Ci * CoP;
      u[base] = wino_row(t[0].x, t[1].x, t[2].x);
      u[base + step] = wino_row(t[0].y, t[1].y, t[2].y);
      u[base + 2 * step] = wino_row(t[0].z, t[1].z, t[2].z);
      u[base + 3 * step] = wino_row(t[0].w, t[1].w, t[2].w);
    }
  }

  // ---- data-gradient operand (the conv over g has C_in = Co, C_out = Ci and reads tap 26 - t where it mirrors)
  if (J.dgrad_kind == CFUN_WOP_PACKT) {         // wpT[t][co][CiP]
    for (int i = tid; i < 256 * T; i += kBlock) {
      const int t = i >> 8, co = (i >> 4) & 15, ci = i & 15;
      if (co0 + co < Co) J.dgrad[((int64_t)t * Co + co0 + co) * CiP + ci0 + ci] = s[(co * 16 + ci) * TS + t];
    }
  } else if (J.dgrad_kind == CFUN_WOP_WINO1_T) {
    float4* u = reinterpret_cast<float4*>(J.dgrad);
    for (int i = tid; i < 256 * 9; i += kBlock) {
      const int r9 = i >> 8, co = (i >> 4) & 15, ci = i & 15;
      const float* g = s + (co * 16 + ci) * TS + 26 - r9 * 3;      // taps 26 - (r9*3 + dx)
      if (co0 + co < Co) u[((int64_t)r9 * Co + co0 + co) * CiP + ci0 + ci] = wino_row(g[0], g[-1], g[-2]);
    }
  } else if (J.dgrad_kind == CFUN_WOP_WINO2_T) {
    float4* u = reinterpret_cast<float4*>(J.dgrad);
    for (int i = tid; i < 256 * 3; i += kBlock) {
      const int dz = i >> 8, co = (i >> 4) & 15, ci = i & 15;
      if (co0 + co >= Co) continue;
      const float* g = s + (co * 16 + ci) * TS + 26 - dz * 9;      // tap (dz,ky,kx) -> 26 - ((dz*3 + ky)*3 + kx)
      float4 t[3];
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) t[kx] = wino_row(g[-kx], g[-3 - kx], g[-6 - kx]);
      const int64_t base = ((int64_t)(dz * 4) * Co + co0 + co) * CiP + ci0 + ci, step = (int64_t)Co * CiP;
      u[base] = wino_row(t[0].x, t[1].x, t[2].x);
      u[base + step] = wino_row(t[0].y, t[1].y, t[2].y);
      u[base + 2 * step] = wino_row(t[0].z, t[1].z, t[2].z);
      u[base + 3 * step] = wino_row(t[0].w, t[1].w, t[2].w);
    }
  } else if (J.dgrad_kind == CFUN_WOP_S2FOLD) {
    // data gradient of a stride-2 3x3x3 conv as a 2x2x2 conv with depth-to-space (conv3d.hip: k_fold_s2_dgrad_weights):
    // wd[tap' = (a,b,c)][co][q*Ci + ci], q = output parity; per axis parity 0 reads tap 1 at offset 0 only, parity 1 reads
    // tap 2 at offset 0 and tap 0 at offset 1
    const int CoPd = (8 * Ci + 15) & ~15;
    for (int i = tid; i < 256 * 64; i += kBlock) {
      const int e = i >> 8, co = (i >> 4) & 15, ci = i & 15;
      if (co0 + co >= Co || ci0 + ci >= Ci) continue;
      const int tap = e >> 3, q = e & 7;
      const int par[3] = {q >> 2, (q >> 1) & 1, q & 1}, off[3] = {tap >> 2, (tap >> 1) & 1, tap & 1};
      int t[3];
      bool ok = true;
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        if (par[d] == 0) { t[d] = 1; ok = ok && off[d] == 0; }
        else t[d] = off[d] == 0 ? 2 : 0;
      }
      J.dgrad[((int64_t)tap * Co + co0 + co) * CoPd + q * Ci + ci0 + ci] = ok ? s[(co * 16 + ci) * TS + (t[0] * 3 + t[1]) * 3 + t[2]] : 0.f;
    }
    if (ci0 == 0)      // the padding columns [8*Ci, CoPd) of this block's rows
      for (int i = tid; i < 8 * 16 * 16; i += kBlock) {
        const int tap = i >> 8, co = (i >> 4) & 15, col = 8 * Ci + (i & 15);
        if (co0 + co < Co && col < CoPd) J.dgrad[((int64_t)tap * Co + co0 + co) * CoPd + col] = 0.f;
      }
  }
}


// ---- fold of "nearest x2 up-sampling -> k^3 conv (pad k/2)" into a 3x3x3 conv on the LOW-resolution input whose 8 output
// parities are channels (cfun_amd.weights.fold_up2_weight; mask_branch.py:108-116, 216-218).  Hi-res tap t of output parity p
// reads the low-res offset a = floor((p + t - k/2) / 2) + 1 in {0, 1, 2} per axis; taps that land on the same low-res voxel are
// summed.  wf [8 * cqp][Ci][27] from w [Co][Ci][k^3] (parity groups padded from Co to cqp channels with zero rows), and the
// transpose for the weight's gradient.  One launch each way instead of pad + batched matmul (+ sum) of torch: the two Tensile
// GEMMs and ~6 glue launches per folded weight and direction that rounds 3-4 had on the path (5 folded weights per step).
__device__ __forceinline__ int fold_a(int p, int t, int k) { return ((p + t - (k >> 1)) >> 1) + 1; }      // (>>: floor)

__global__ void __launch_bounds__(256)
k_fold_up2_fwd(const float* __restrict__ w, float* __restrict__ wf, int Co, int Ci, int k, int cqp, int64_t total) {
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    int64_t r = idx;
    const int abc = (int)(r % 27); r /= 27;
    const int ci = (int)(r % Ci); r /= Ci;
    const int oc = (int)(r % cqp);
    const int q = (int)(r / cqp);
    float acc = 0.f;
    if (oc < Co) {
      const int a = abc / 9, b = (abc / 3) % 3, c = abc % 3, pz = q >> 2, py = (q >> 1) & 1, px = q & 1;
      const float* src = w + ((int64_t)oc * Ci + ci) * k * k * k;
      for (int t = 0; t < k; ++t) {
        if (fold_a(pz, t, k) != a) continue;
        for (int u = 0; u < k; ++u) {
          if (fold_a(py, u, k) != b) continue;
          for (int v = 0; v < k; ++v)
            if (fold_a(px, v, k) == c) acc += src[(t * k + u) * k + v];
        }
      }
    }
    wf[idx] = acc;
  }
}

__global__ void __launch_bounds__(256)
k_fold_up2_bwd(const float* __restrict__ g, float* __restrict__ dw, int Co, int Ci, int k, int cqp, int64_t total) {
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    int64_t r = idx;
    const int v = (int)(r % k); r /= k;
    const int u = (int)(r % k); r /= k;
    const int t = (int)(r % k); r /= k;
    const int ci = (int)(r % Ci);
    const int oc = (int)(r / Ci);
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {      // each parity reads this hi-res tap at exactly one low-res offset
      const int abc = (fold_a(q >> 2, t, k) * 3 + fold_a((q >> 1) & 1, u, k)) * 3 + fold_a(q & 1, v, k);
      acc += g[(((int64_t)q * cqp + oc) * Ci + ci) * 27 + abc];
    }
    dw[idx] = acc;
  }
}

}  // namespace

extern "C" {

int cfun_weight_prepare_plan(CfunWeightJob* jobs, int32_t njobs, int64_t* nblocks) {
  if (njobs < 0 || (njobs > 0 && !jobs) || !nblocks) return CFUN_EINVAL;
  int64_t total = 0;
  for (int j = 0; j < njobs; ++j) {
    CfunWeightJob& J = jobs[j];
    if (!J.w || J.Co <= 0 || J.Ci <= 0 || J.T <= 0 || J.T > 27 || J.src_ci <= 0) return CFUN_EINVAL;
    if ((J.fwd_kind != CFUN_WOP_NONE && !J.fwd) || (J.dgrad_kind != CFUN_WOP_NONE && !J.dgrad)) return CFUN_EINVAL;
    const bool fk = J.fwd_kind == CFUN_WOP_NONE || J.fwd_kind == CFUN_WOP_PACK || J.fwd_kind == CFUN_WOP_WINO1 || J.fwd_kind == CFUN_WOP_WINO2;
    const bool dk = J.dgrad_kind == CFUN_WOP_NONE || J.dgrad_kind == CFUN_WOP_PACKT || J.dgrad_kind == CFUN_WOP_WINO1_T ||
                    J.dgrad_kind == CFUN_WOP_WINO2_T || J.dgrad_kind == CFUN_WOP_S2FOLD;
    if (!fk || !dk) return CFUN_EINVAL;
    const bool wino = J.fwd_kind == CFUN_WOP_WINO1 || J.fwd_kind == CFUN_WOP_WINO2 || J.dgrad_kind == CFUN_WOP_WINO1_T ||
                      J.dgrad_kind == CFUN_WOP_WINO2_T || J.dgrad_kind == CFUN_WOP_S2FOLD;
    if (wino && J.T != 27) return CFUN_EINVAL;
    if (total > 0x7fffffffLL) return CFUN_EINVAL;
    J.block_begin = (uint32_t)total;
    total += (int64_t)((J.Co + 15) / 16) * ((J.Ci + 15) / 16);
  }
  if (total > 0x7fffffffLL) return CFUN_EINVAL;
  *nblocks = total;
  return CFUN_OK;
}

int cfun_fold_up2_fwd(const float* w, float* wf, int32_t Co, int32_t Ci, int32_t k, int32_t cqp, cfun_stream_t stream) {
  if (!w || !wf || Co <= 0 || Ci <= 0 || (k != 3 && k != 5) || cqp < Co) return CFUN_EINVAL;
  const int64_t total = (int64_t)8 * cqp * Ci * 27;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 256 * 64) blocks = 256 * 64;
  hipLaunchKernelGGL(k_fold_up2_fwd, dim3((unsigned)blocks), dim3(256), 0, cfun_st(stream), w, wf, Co, Ci, k, cqp, total);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

int cfun_fold_up2_bwd(const float* g, float* dw, int32_t Co, int32_t Ci, int32_t k, int32_t cqp, cfun_stream_t stream) {
  if (!g || !dw || Co <= 0 || Ci <= 0 || (k != 3 && k != 5) || cqp < Co) return CFUN_EINVAL;
  const int64_t total = (int64_t)Co * Ci * k * k * k;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 256 * 64) blocks = 256 * 64;
  hipLaunchKernelGGL(k_fold_up2_bwd, dim3((unsigned)blocks), dim3(256), 0, cfun_st(stream), g, dw, Co, Ci, k, cqp, total);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

int cfun_weight_prepare(const CfunWeightJob* jobs_dev, int32_t njobs, int64_t nblocks, cfun_stream_t stream) {
  if (njobs <= 0 || nblocks <= 0) return CFUN_OK;
  if (!jobs_dev || nblocks > 0x7fffffffLL) return CFUN_EINVAL;
  hipLaunchKernelGGL(k_weight_prepare, dim3((unsigned)nblocks), dim3(kBlock), 0, cfun_st(stream), jobs_dev, njobs);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}

}  // extern "C"
