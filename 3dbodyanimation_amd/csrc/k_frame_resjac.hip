// k_frame_resjac.hip — per-frame keypoint residuals + analytic Jacobian, f64, one 4-wave workgroup per frame.
//
// Replaces, for every reprojection block of a frame at once, what the reference evaluates through
// ceres::DynamicAutoDiffCostFunction<ReprojCost[Shape]> (include/Sim3BA.h:34-88,126-227,420,581;
// include/MultiFrameBA.h:85-102): 22 dual-number passes per 2-residual block become one closed-form
// Jacobian.  The same wave also prepares the operands of the mesh kernel (pose-feature fragments in
// bf16 hi/lo, shape coefficients, 24 skinning transforms) so the two kernels share one Rodrigues pass.
//
// Work layout (256 threads = 4 CDNA4 wavefronts per frame, one per SIMD of the CU; phases separated by
// workgroup barriers, all intermediate state in LDS):
//   lanes = joints      Rodrigues R_j, dR_j/da (both branches of Ceres' AngleAxisRotatePoint)
//   lanes = (joint,e)   level-synchronous kinematic chain A_j, P_j, dP_j/dbeta, staged in LDS
//   lanes = (k,c)       W_{k,c} = A_par(k) dR_{k,c} R_k^T A_par(k)^T  (d x / d a_{k,c} = W (x - P_k))
//   wave reductions     landmark blend rows  v_p = v_t + sd.beta + pd.feat   (coalesced 512-B reads, one
//                       landmark per wave at a time)
//   lanes = columns     the dense row-major [2K][ncols] panel is written with consecutive lanes on
//                       consecutive columns (coalesced 512-B stores), keypoints dealt round-robin to waves
#include "bodyfit_device.h"

namespace bodyfit {
namespace {

__device__ inline void mul33(const double* A, const double* B, double* C) {  // C = A B
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) C[r * 3 + c] = A[r * 3] * B[c] + A[r * 3 + 1] * B[3 + c] + A[r * 3 + 2] * B[6 + c];
}
__device__ inline void mul33_bt(const double* A, const double* B, double* C) {  // C = A B^T
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c)
      C[r * 3 + c] = A[r * 3] * B[c * 3] + A[r * 3 + 1] * B[c * 3 + 1] + A[r * 3 + 2] * B[c * 3 + 2];
}
__device__ inline void mv3(const double* A, double x0, double x1, double x2, double* y) {
#pragma unroll
  for (int r = 0; r < 3; ++r) y[r] = A[r * 3] * x0 + A[r * 3 + 1] * x1 + A[r * 3 + 2] * x2;
}

// R(a) and dR/da_c.  theta^2 <= DBL_EPSILON: R = I + [a]x, dR_c = [e_c]x (the first-order branch).
__device__ void rodrigues_grad(double a0, double a1, double a2, double* R, double* dR) {
  const double th2 = a0 * a0 + a1 * a1 + a2 * a2;
  if (th2 > 2.220446049250313e-16) {
    const double th = sqrt(th2), ith = 1.0 / th;
    double st, ct;
    sincos(th, &st, &ct);
    const double sh = sin(0.5 * th);
    const double omc = 2.0 * sh * sh;
    const double w[3] = {a0 * ith, a1 * ith, a2 * ith};
    const double K[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) R[r * 3 + c] = (r == c ? ct : 0.0) + st * K[r * 3 + c] + omc * w[r] * w[c];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      double dw[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) dw[i] = ((i == k ? 1.0 : 0.0) - w[i] * w[k]) * ith;
      const double dK[9] = {0, -dw[2], dw[1], dw[2], 0, -dw[0], -dw[1], dw[0], 0};
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
          dR[k * 9 + r * 3 + c] = (r == c ? -st * w[k] : 0.0) + ct * w[k] * K[r * 3 + c] + st * dK[r * 3 + c] +
                                  st * w[k] * w[r] * w[c] + omc * (dw[r] * w[c] + w[r] * dw[c]);
    }
  } else {
    R[0] = 1; R[1] = -a2; R[2] = a1;
    R[3] = a2; R[4] = 1; R[5] = -a0;
    R[6] = -a1; R[7] = a0; R[8] = 1;
#pragma unroll
    for (int i = 0; i < 27; ++i) dR[i] = 0.0;
    dR[0 * 9 + 5] = -1; dR[0 * 9 + 7] = 1;   // [e_x]x
    dR[1 * 9 + 2] = 1;  dR[1 * 9 + 6] = -1;  // [e_y]x
    dR[2 * 9 + 1] = -1; dR[2 * 9 + 3] = 1;   // [e_z]x
  }
}

__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

constexpr int KC = 32;  // keypoints staged per chunk
constexpr int kThreads = 256;

// LDS carve (doubles)
constexpr int OFF_X = 0;                       // 88
constexpr int OFF_R = OFF_X + 88;              // 24*9
constexpr int OFF_DR = OFF_R + 216;            // 24*27
constexpr int OFF_A = OFF_DR + 648;            // 24*9
constexpr int OFF_P = OFF_A + 216;             // 24*3
constexpr int OFF_O = OFF_P + 72;              // 24*3
constexpr int OFF_JC = OFF_O + 72;             // 24*3
constexpr int OFF_W = OFF_JC + 72;             // 69*9 -> 624
constexpr int OFF_B = OFF_W + 624;             // 24*30
constexpr int OFF_FEAT = OFF_B + 720;          // 208
constexpr int OFF_CAM = OFF_FEAT + 208;        // Rr0[9], dRr0[27], pad -> 40
constexpr int OFF_KP = OFF_CAM + 40;           // KC*18
constexpr int OFF_TAB = OFF_KP + KC * 18;      // int tables (as 4-byte words): 128 ints = 64 doubles
constexpr int OFF_LM = OFF_TAB + 64;           // landmarks: nL * LM_STRIDE
constexpr int LM_VP = 0;                       // 3
constexpr int LM_Q = 3;                        // 3
constexpr int LM_A = 6;                        // 9  blended rotation
constexpr int LM_X = 15;                       // kMaxLmNnz*3 = 24
constexpr int LM_W = 39;                       // kMaxLmNnz weights
constexpr int LM_J = 47;                       // kMaxLmNnz joint ids (stored as doubles' worth of ints: 8 ints = 4 doubles)
constexpr int LM_NW = 51;                      // weight count (as double)
constexpr int LM_BETA = 52;                    // 10*3 = 30   d q / d beta
constexpr int LM_PD = 82;                      // 69*3 = 207  pose-blend Jacobian term
constexpr int LM_STRIDE = 290;
// int table layout inside OFF_TAB
constexpr int TAB_PARENT = 0;                  // 24
constexpr int TAB_ANC = 24;                    // 24
constexpr int TAB_LVOFF = 48;                  // 25
constexpr int TAB_LVJ = 73;                    // 23
constexpr int TAB_KPID = 96;                   // KC

__global__ __launch_bounds__(kThreads) void k_frame_resjac(DevModel M, DevProblem Pb, const double* __restrict__ params,
                                                      const double* __restrict__ beta, double* __restrict__ r_out,
                                                      double* __restrict__ J_out, double* __restrict__ joints_out,
                                                      MeshCoef mc, int want_jac) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int f = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int nJ = M.nJ, nS = M.nS, P = M.P, nL = M.nL;
  const int ncols = Pb.ncols;
  const int npose = 7 + 3 * (nJ - 1);
  const bool use_shape = Pb.use_shape != 0;
  double* sx = sm + OFF_X;
  double* sR = sm + OFF_R;
  double* sdR = sm + OFF_DR;
  double* sA = sm + OFF_A;
  double* sP = sm + OFF_P;
  double* sO = sm + OFF_O;
  double* sJc = sm + OFF_JC;
  double* sW = sm + OFF_W;
  double* sB = sm + OFF_B;
  double* sFeat = sm + OFF_FEAT;
  double* sCam = sm + OFF_CAM;
  double* sKp = sm + OFF_KP;
  int* sTab = reinterpret_cast<int*>(sm + OFF_TAB);
  double* sLm = sm + OFF_LM;
  int* sParent = sTab + TAB_PARENT;
  unsigned* sAnc = reinterpret_cast<unsigned*>(sTab + TAB_ANC);
  int* sLvOff = sTab + TAB_LVOFF;
  int* sLvJ = sTab + TAB_LVJ;
  int* sKpId = sTab + TAB_KPID;

  // ---- 0. small model tables into LDS (every later loop reads them from there) ----------------
  if (tid < nJ) { sParent[tid] = M.parent[tid]; sAnc[tid] = M.anc_mask[tid]; }
  if (tid <= M.nLevels) sLvOff[tid] = M.level_off[tid];
  if (tid < nJ - 1) sLvJ[tid] = M.level_joint[tid];
  if (tid < nL) {
    const int w0 = M.lm_woff[tid], nw = M.lm_woff[tid + 1] - w0;
    double* L = sLm + tid * LM_STRIDE;
    int* Lj = reinterpret_cast<int*>(L + LM_J);
    L[LM_NW] = (double)nw;
    for (int i = 0; i < kMaxLmNnz; ++i) {
      L[LM_W + i] = (i < nw) ? M.lm_ww[w0 + i] : 0.0;
      Lj[i] = (i < nw) ? M.lm_wj[w0 + i] : 0;
    }
  }
  // ---- 1. parameters -------------------------------------------------------------------------
  for (int i = tid; i < npose; i += kThreads) sx[i] = params[(size_t)f * npose + i];
  for (int i = tid; i < nS; i += kThreads) sx[npose + i] = (use_shape && beta) ? beta[(size_t)f * Pb.beta_stride + i] : 0.0;
  __syncthreads();
  const double* sbeta = sx + npose;

  // ---- 2. Rodrigues + gradient per joint (lane = joint; joint 0 = root angle-axis) -------------
  if (tid < nJ) {
    const double* aa = (tid == 0) ? (sx + 1) : (sx + 7 + 3 * (tid - 1));
    double R[9], dR[27];
    rodrigues_grad(aa[0], aa[1], aa[2], R, dR);
#pragma unroll
    for (int i = 0; i < 9; ++i) sR[tid * 9 + i] = R[i];
#pragma unroll
    for (int i = 0; i < 27; ++i) sdR[tid * 27 + i] = dR[i];
  }
  // ---- 3. chain offsets o_j(beta) (include/Sim3BA.h:142-170,179-205) and centred rest joints ----
  for (int i = tid; i < nJ * 3; i += kThreads) {
    double o = M.offset[i], jc = M.Jc0[i];
    if (use_shape) {
      for (int k = 0; k < nS; ++k) {
        o += M.dS[(size_t)i * nS + k] * sbeta[k];
        jc += M.Sc[(size_t)i * nS + k] * sbeta[k];
      }
    }
    sO[i] = (i < 3) ? 0.0 : o;
    sJc[i] = jc;
  }
  if (tid < 9) sA[tid] = (tid % 4 == 0) ? 1.0 : 0.0;
  if (tid < 3) sP[tid] = 0.0;
  for (int i = tid; i < 3 * nS; i += kThreads) sB[i] = 0.0;
  __syncthreads();
  // pose feature vec(R_j - I), j = 1..nJ-1 (row-major), zero padded to 208
  for (int i = tid; i < 208; i += kThreads) {
    double v = 0.0;
    if (i < 9 * (nJ - 1)) {
      const int e = i % 9;
      v = sR[9 + i] - ((e % 4 == 0) ? 1.0 : 0.0);
    }
    sFeat[i] = v;
  }

  // ---- 4. level-synchronous chain: A_j = A_p R_j, P_j = P_p + A_p o_j, B_j = B_p + A_p dS_j -----
  for (int lv = 0; lv < M.nLevels; ++lv) {
    const int j0 = sLvOff[lv], nj = sLvOff[lv + 1] - j0;
    for (int i = tid; i < nj * 9; i += kThreads) {
      const int j = sLvJ[j0 + i / 9], e = i % 9, r = e / 3, c = e % 3;
      const int p = sParent[j];
      sA[j * 9 + e] = sA[p * 9 + r * 3] * sR[j * 9 + c] + sA[p * 9 + r * 3 + 1] * sR[j * 9 + 3 + c] +
                      sA[p * 9 + r * 3 + 2] * sR[j * 9 + 6 + c];
    }
    for (int i = tid; i < nj * 3; i += kThreads) {
      const int j = sLvJ[j0 + i / 3], r = i % 3;
      const int p = sParent[j];
      sP[j * 3 + r] = sP[p * 3 + r] + sA[p * 9 + r * 3] * sO[j * 3] + sA[p * 9 + r * 3 + 1] * sO[j * 3 + 1] +
                      sA[p * 9 + r * 3 + 2] * sO[j * 3 + 2];
    }
    if (use_shape && want_jac) {
      for (int i = tid; i < nj * 3 * nS; i += kThreads) {
        const int j = sLvJ[j0 + i / (3 * nS)], rem = i % (3 * nS), r = rem / nS, k = rem % nS;
        const int p = sParent[j];
        const double* d = M.dS + (size_t)j * 3 * nS;
        sB[(j * 3 + r) * nS + k] = sB[(p * 3 + r) * nS + k] + sA[p * 9 + r * 3] * d[k] +
                                   sA[p * 9 + r * 3 + 1] * d[nS + k] + sA[p * 9 + r * 3 + 2] * d[2 * nS + k];
      }
    }
    __syncthreads();
  }

  // ---- 5. camera matrices: Rr0 = R_root R0, dRr0_c = dR_root,c R0 -------------------------------
  {
    const double* R0 = Pb.R0 + (size_t)f * 9;
    if (tid < 36) {
      const int mtx = tid / 9, e = tid % 9, r = e / 3, c = e % 3;
      const double* L = (mtx == 0) ? sR : (sdR + (mtx - 1) * 9);
      sCam[tid] = L[r * 3] * R0[c] + L[r * 3 + 1] * R0[3 + c] + L[r * 3 + 2] * R0[6 + c];
    }
  }
  // ---- 6. W_{k,c} = A_p (dR_{k,c} R_k^T) A_p^T  (lane = (k,c)) -------------------------------------
  if (want_jac) {
    for (int i = tid; i < 3 * (nJ - 1); i += kThreads) {
      const int k = 1 + i / 3, c = i % 3, p = sParent[k];
      double T1[9], T2[9], Wm[9];
      mul33_bt(sdR + k * 27 + c * 9, sR + k * 9, T1);
      mul33(sA + p * 9, T1, T2);
      mul33_bt(T2, sA + p * 9, Wm);
#pragma unroll
      for (int e = 0; e < 9; ++e) sW[i * 9 + e] = Wm[e];
    }
  }
  __syncthreads();
  const double s = sx[0];
  const double* Rr0 = sCam;
  const double* dRr0 = sCam + 9;

  // ---- 7. outputs for the mesh kernel and the posed joints -----------------------------------------
  if (tid < nJ) {
    const int jj = tid;
    double RA[9], t[3], q[3];
    mul33(Rr0, sA + jj * 9, RA);
    mv3(sA + jj * 9, sJc[jj * 3], sJc[jj * 3 + 1], sJc[jj * 3 + 2], q);
    mv3(Rr0, sP[jj * 3] - q[0], sP[jj * 3 + 1] - q[1], sP[jj * 3 + 2] - q[2], t);
    if (mc.skinT) {
      float* T = mc.skinT + ((size_t)f * nJ + jj) * 12;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        T[r * 4 + 0] = (float)(s * RA[r * 3 + 0]);
        T[r * 4 + 1] = (float)(s * RA[r * 3 + 1]);
        T[r * 4 + 2] = (float)(s * RA[r * 3 + 2]);
        T[r * 4 + 3] = (float)(s * t[r] + sx[4 + r]);
      }
    }
    if (joints_out) {
      mv3(Rr0, sP[jj * 3], sP[jj * 3 + 1], sP[jj * 3 + 2], t);
#pragma unroll
      for (int r = 0; r < 3; ++r) joints_out[((size_t)f * nJ + jj) * 3 + r] = s * t[r] + sx[4 + r];
    }
  }
  if (mc.featA) {
    const int ftile = f / kFTile, row = f % kFTile;
    if (wave == 1 && lane < kPoseKSteps * 4) {
      const int kstep = lane >> 2, h = (lane >> 1) & 1, hl = lane & 1;
      uint32_t pk[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        uint16_t b[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const float x = Pb.pose_blend ? (float)sFeat[kstep * 16 + 8 * h + 2 * jj + u] : 0.0f;
          const uint16_t hi = f32_to_bf16(x);
          b[u] = hl == 0 ? hi : f32_to_bf16(x - bf16_to_f32(hi));
        }
        pk[jj] = (uint32_t)b[0] | ((uint32_t)b[1] << 16);
      }
      uint4* dst = reinterpret_cast<uint4*>(mc.featA + ((((size_t)ftile * kPoseKSteps + kstep) * 2 + hl) * 64 + (h * 32 + row)) * 8);
      *dst = make_uint4(pk[0], pk[1], pk[2], pk[3]);
    }
    if (wave == 2 && lane < 2 * kShapeKSteps) {
      const int kstep = lane >> 1, h = lane & 1, k = 2 * kstep + h;
      mc.betaA[((size_t)ftile * kShapeKSteps + kstep) * 64 + h * 32 + row] = (k < nS) ? (float)sbeta[k] : 0.0f;
    }
  }

  // ---- 8. vertex landmarks: blend rows by wave reductions (3 rows in flight), then LBS per landmark ---
  if (nL > 0) {
    for (int l = wave; l < nL; l += kThreads / 64) {
      double acc[3] = {0.0, 0.0, 0.0};
      if (Pb.pose_blend && P > 0) {
        for (int i = lane; i < P; i += 64) {
          const double ft = sFeat[i];
#pragma unroll
          for (int a = 0; a < 3; ++a) acc[a] += M.lm_pd[((size_t)l * 3 + a) * P + i] * ft;
        }
      }
      if (use_shape && lane < nS) {
#pragma unroll
        for (int a = 0; a < 3; ++a) acc[a] += M.lm_sd[((size_t)l * 3 + a) * nS + lane] * sbeta[lane];
      }
#pragma unroll
      for (int a = 0; a < 3; ++a) acc[a] = wave_sum(acc[a]);
      if (lane == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) sLm[l * LM_STRIDE + LM_VP + a] = M.lm_vt[l * 3 + a] + acc[a];
      }
    }
    __syncthreads();
    if (tid < nL) {
      double* L = sLm + tid * LM_STRIDE;
      const int* Lj = reinterpret_cast<const int*>(L + LM_J);
      double q[3] = {0, 0, 0}, Ab[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      const int nw = (int)L[LM_NW];
      for (int i = 0; i < nw; ++i) {
        const int j = Lj[i];
        const double w = L[LM_W + i];
        double xj[3];
        mv3(sA + j * 9, L[LM_VP] - sJc[j * 3], L[LM_VP + 1] - sJc[j * 3 + 1], L[LM_VP + 2] - sJc[j * 3 + 2], xj);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          xj[a] += sP[j * 3 + a];
          q[a] += w * xj[a];
          L[LM_X + i * 3 + a] = xj[a];
        }
#pragma unroll
        for (int e = 0; e < 9; ++e) Ab[e] += w * sA[j * 9 + e];
      }
#pragma unroll
      for (int a = 0; a < 3; ++a) L[LM_Q + a] = q[a];
#pragma unroll
      for (int e = 0; e < 9; ++e) L[LM_A + e] = Ab[e];
    }
    __syncthreads();
    if (want_jac) {
      // pose-blend term  Ablend . (pd[:, 9(k-1):9k] . vec(dR_{k,c}))   item = (landmark, joint k):
      // 27 independent loads (72 contiguous bytes per row, consecutive lanes on consecutive k)
      const int nItems = nL * (nJ - 1);
      for (int it = tid; it < nItems; it += kThreads) {
        const int l = it / (nJ - 1), k = 1 + it % (nJ - 1);
        double pdv[27];
        if (Pb.pose_blend && P > 0) {
#pragma unroll
          for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int e = 0; e < 9; ++e) pdv[a * 9 + e] = M.lm_pd[((size_t)l * 3 + a) * P + 9 * (k - 1) + e];
        } else {
#pragma unroll
          for (int e = 0; e < 27; ++e) pdv[e] = 0.0;
        }
        const double* Ab = sLm + l * LM_STRIDE + LM_A;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const double* d = sdR + k * 27 + c * 9;
          double h[3];
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            double acc = 0;
#pragma unroll
            for (int e = 0; e < 9; ++e) acc += pdv[a * 9 + e] * d[e];
            h[a] = acc;
          }
          double t[3];
          mv3(Ab, h[0], h[1], h[2], t);
          double* o = sLm + l * LM_STRIDE + LM_PD + (3 * (k - 1) + c) * 3;
          o[0] = t[0]; o[1] = t[1]; o[2] = t[2];
        }
      }
      // shape columns  d q / d beta_k = sum_i w_i (A_j (sd_l - Sc_j) + B_j)[:, k]   item = (landmark, k)
      if (use_shape) {
        for (int it = tid; it < nL * nS; it += kThreads) {
          const int l = it / nS, k = it % nS;
          const double* L = sLm + l * LM_STRIDE;
          const int* Lj = reinterpret_cast<const int*>(L + LM_J);
          const int nw = (int)L[LM_NW];
          const double s0 = M.lm_sd[(size_t)(l * 3 + 0) * nS + k], s1 = M.lm_sd[(size_t)(l * 3 + 1) * nS + k],
                       s2 = M.lm_sd[(size_t)(l * 3 + 2) * nS + k];
          double d0 = 0, d1 = 0, d2 = 0;
          for (int i = 0; i < nw; ++i) {
            const int j = Lj[i];
            const double w = L[LM_W + i];
            double t[3];
            mv3(sA + j * 9, s0 - M.Sc[(size_t)(j * 3 + 0) * nS + k], s1 - M.Sc[(size_t)(j * 3 + 1) * nS + k],
                s2 - M.Sc[(size_t)(j * 3 + 2) * nS + k], t);
            d0 += w * (t[0] + sB[(j * 3 + 0) * nS + k]);
            d1 += w * (t[1] + sB[(j * 3 + 1) * nS + k]);
            d2 += w * (t[2] + sB[(j * 3 + 2) * nS + k]);
          }
          double* o = sLm + l * LM_STRIDE + LM_BETA + k * 3;
          o[0] = d0; o[1] = d1; o[2] = d2;
        }
      }
      __syncthreads();
    }
  }

  // ---- 9. keypoints of this frame, KC at a time ------------------------------------------------------
  const int k_begin = Pb.kp_offset[f], k_end = Pb.kp_offset[f + 1];
  for (int kc0 = k_begin; kc0 < k_end; kc0 += KC) {
    const int nk = min(KC, k_end - kc0);
    if (tid < nk) {
      const int kg = kc0 + tid;
      const int id = Pb.kp_id[kg];
      sKpId[tid] = id;
      double q[3];
      if (id < nJ) {
        if (id == 0 || sParent[id] < 0) {
          // include/Sim3BA.h:142-170 without a chain: q = offset + S_id beta (no parent term)
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            double v = M.offset[id * 3 + a];
            if (use_shape)
              for (int k = 0; k < nS; ++k) v += M.dS[(size_t)(id * 3 + a) * nS + k] * sbeta[k];
            q[a] = v;
          }
        } else {
#pragma unroll
          for (int a = 0; a < 3; ++a) q[a] = sP[id * 3 + a];
        }
      } else {
#pragma unroll
        for (int a = 0; a < 3; ++a) q[a] = sLm[(id - nJ) * LM_STRIDE + LM_Q + a];
      }
      double z[3];
      mv3(Rr0, q[0], q[1], q[2], z);                     // include/Sim3BA.h:210-216
      const double X0 = s * z[0] + sx[4], X1 = s * z[1] + sx[5], X2 = s * z[2] + sx[6];  // :217-219
      const double iz = 1.0 / X2;                        // :222-223, Z unguarded as in the reference
      r_out[2 * (size_t)kg] = Pb.fx * X0 * iz + Pb.cx - Pb.kp_uv[2 * (size_t)kg];
      r_out[2 * (size_t)kg + 1] = Pb.fy * X1 * iz + Pb.cy - Pb.kp_uv[2 * (size_t)kg + 1];
      double* kp = sKp + tid * 18;
      const double dpi[6] = {Pb.fx * iz, 0.0, -Pb.fx * X0 * iz * iz, 0.0, Pb.fy * iz, -Pb.fy * X1 * iz * iz};
#pragma unroll
      for (int a = 0; a < 3; ++a) { kp[a] = q[a]; kp[3 + a] = z[a]; }
#pragma unroll
      for (int i = 0; i < 6; ++i) kp[6 + i] = dpi[i];
#pragma unroll
      for (int rr = 0; rr < 2; ++rr)
#pragma unroll
        for (int c = 0; c < 3; ++c)
          kp[12 + rr * 3 + c] = s * (dpi[rr * 3] * Rr0[c] + dpi[rr * 3 + 1] * Rr0[3 + c] + dpi[rr * 3 + 2] * Rr0[6 + c]);
    }
    __syncthreads();
    if (want_jac) {
      for (int kk = wave; kk < nk; kk += kThreads / 64) {
        const int kg = kc0 + kk;
        const int id = sKpId[kk];
        const double* kp = sKp + kk * 18;
        const double* G = kp + 12;
        for (int col = lane; col < ncols; col += 64) {
          double j0, j1;
          if (col == 0) {
            j0 = kp[6] * kp[3] + kp[7] * kp[4] + kp[8] * kp[5];
            j1 = kp[9] * kp[3] + kp[10] * kp[4] + kp[11] * kp[5];
          } else if (col < 4) {
            double t[3];
            mv3(dRr0 + (col - 1) * 9, kp[0], kp[1], kp[2], t);
            j0 = s * (kp[6] * t[0] + kp[7] * t[1] + kp[8] * t[2]);
            j1 = s * (kp[9] * t[0] + kp[10] * t[1] + kp[11] * t[2]);
          } else if (col < 7) {
            j0 = kp[6 + (col - 4)];
            j1 = kp[9 + (col - 4)];
          } else {
            double d0 = 0.0, d1 = 0.0, d2 = 0.0;
            if (col < npose) {
              const int kc = col - 7, k = 1 + kc / 3;
              if (id < nJ) {
                if ((sAnc[id] >> k) & 1u) {
                  double t[3];
                  mv3(sW + kc * 9, kp[0] - sP[k * 3], kp[1] - sP[k * 3 + 1], kp[2] - sP[k * 3 + 2], t);
                  d0 = t[0]; d1 = t[1]; d2 = t[2];
                }
              } else {
                const int l = id - nJ;
                const double* L = sLm + l * LM_STRIDE;
                const int* Lj = reinterpret_cast<const int*>(L + LM_J);
                const int nw = (int)L[LM_NW];
                double a0 = 0, a1 = 0, a2 = 0;
                for (int i = 0; i < nw; ++i) {
                  const int j = Lj[i];
                  if (j == k || ((sAnc[j] >> k) & 1u)) {
                    const double w = L[LM_W + i];
                    a0 += w * (L[LM_X + i * 3] - sP[k * 3]);
                    a1 += w * (L[LM_X + i * 3 + 1] - sP[k * 3 + 1]);
                    a2 += w * (L[LM_X + i * 3 + 2] - sP[k * 3 + 2]);
                  }
                }
                double t[3];
                mv3(sW + kc * 9, a0, a1, a2, t);
                d0 = t[0] + L[LM_PD + kc * 3];
                d1 = t[1] + L[LM_PD + kc * 3 + 1];
                d2 = t[2] + L[LM_PD + kc * 3 + 2];
              }
            } else if (use_shape) {
              const int k = col - npose;
              if (id < nJ) {
                if (id == 0 || sParent[id] < 0) {
                  d0 = M.dS[(size_t)(id * 3 + 0) * nS + k];
                  d1 = M.dS[(size_t)(id * 3 + 1) * nS + k];
                  d2 = M.dS[(size_t)(id * 3 + 2) * nS + k];
                } else {
                  d0 = sB[(id * 3 + 0) * nS + k];
                  d1 = sB[(id * 3 + 1) * nS + k];
                  d2 = sB[(id * 3 + 2) * nS + k];
                }
              } else {
                const double* o = sLm + (id - nJ) * LM_STRIDE + LM_BETA + k * 3;
                d0 = o[0]; d1 = o[1]; d2 = o[2];
              }
            }
            j0 = G[0] * d0 + G[1] * d1 + G[2] * d2;
            j1 = G[3] * d0 + G[4] * d1 + G[5] * d2;
          }
          J_out[(size_t)(2 * kg) * ncols + col] = j0;
          J_out[(size_t)(2 * kg + 1) * ncols + col] = j1;
        }
      }
    }
    __syncthreads();
  }
}

}  // namespace

void launch_frame_resjac(const DevModel& M, const DevProblem& P, const double* d_params, const double* d_beta,
                         double* d_r, double* d_J, double* d_joints, const MeshCoef& mc, int want_jac,
                         hipStream_t s) {
  if (P.F <= 0) return;
  const size_t lds = (size_t)(OFF_LM + M.nL * LM_STRIDE) * sizeof(double);
  static size_t lds_granted = 48 * 1024;
  if (lds > lds_granted) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_frame_resjac), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    lds_granted = lds;
  }
  hipLaunchKernelGGL(k_frame_resjac, dim3(P.F), dim3(kThreads), lds, s, M, P, d_params, d_beta, d_r, d_J, d_joints, mc,
                     want_jac);
}

}  // namespace bodyfit
