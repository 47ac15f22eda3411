// bodyfit.hpp — C++17 host API mirroring the reference's entry points over the C ABI (bodyfit.h).
//
// Same names, argument order and error behaviour as the reference so a caller of
//   OptimizePoseReprojection / OptimizePoseShapeReprojection   (include/Sim3BA.h:348-358,515-525)
//   OptimizeMultiFrame                                          (include/MultiFrameBA.h:33-43)
// switches by changing the include and the namespace.  ark::AvatarModel / ark::Avatar /
// ark::GaussianMixture (external/avatar, not vendored by the reference) are replaced by thin
// owners of the device handles; Eigen types by std::array / std::vector (row-major 3x3).
// The outer loop is bodyfit_solve (Ceres-like LM); every evaluation runs on the GPU.
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "bodyfit.h"

namespace bodyfit {

struct PixelKP { int jid; double u, v; };                       // include/Sim3BA.h:9

struct Sim3Params {                                             // include/Sim3BA.h:11-19
  double data[7];  // [s, aa(3), t(3)]
  double& scale() { return data[0]; }
  double* aa_root() { return data + 1; }
  double* trans() { return data + 4; }
  const double& scale() const { return data[0]; }
  const double* aa_root() const { return data + 1; }
  const double* trans() const { return data + 4; }
};

struct FramePoseParams {                                        // include/MultiFrameBA.h:9-14
  double scale;
  double rootAA[3];
  double rootT[3];
  std::vector<std::array<double, 3>> jointAA;  // size == nJ, index 0 unused
};

using Matrix3d = std::array<double, 9>;  // row-major

inline Matrix3d Identity3() { return {1, 0, 0, 0, 1, 0, 0, 0, 1}; }
inline Matrix3d Mul(const Matrix3d& A, const Matrix3d& B) {
  Matrix3d C{};
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) C[r * 3 + c] = A[r * 3] * B[c] + A[r * 3 + 1] * B[3 + c] + A[r * 3 + 2] * B[6 + c];
  return C;
}
// Eigen::AngleAxisd(theta, axis).toRotationMatrix() as used in the write-back (include/Sim3BA.h:483-496)
inline Matrix3d AngleAxisToMatrix(const double aa[3]) {
  const double th = std::sqrt(aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2]);
  if (!(th > 1e-12)) return Identity3();
  const double x = aa[0] / th, y = aa[1] / th, z = aa[2] / th, c = std::cos(th), s = std::sin(th), t = 1 - c;
  return {t * x * x + c, t * x * y - s * z, t * x * z + s * y, t * x * y + s * z, t * y * y + c, t * y * z - s * x,
          t * x * z - s * y, t * y * z + s * x, t * z * z + c};
}
// log map, for Avatar::update() which is driven by rotation matrices
inline void MatrixToAngleAxis(const Matrix3d& R, double aa[3]) {
  const double tr = R[0] + R[4] + R[8];
  double c = std::min(1.0, std::max(-1.0, 0.5 * (tr - 1.0)));
  const double th = std::acos(c);
  const double v[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
  if (th < 1e-12) { aa[0] = 0.5 * v[0]; aa[1] = 0.5 * v[1]; aa[2] = 0.5 * v[2]; return; }
  const double s = std::sin(th);
  if (std::fabs(s) > 1e-6) {
    const double k = th / (2.0 * s);
    aa[0] = k * v[0]; aa[1] = k * v[1]; aa[2] = k * v[2];
    return;
  }
  // theta near pi: axis from the diagonal
  double ax[3] = {std::sqrt(std::max(0.0, 0.5 * (R[0] + 1))), std::sqrt(std::max(0.0, 0.5 * (R[4] + 1))),
                  std::sqrt(std::max(0.0, 0.5 * (R[8] + 1)))};
  if (R[1] + R[3] < 0) ax[1] = -ax[1];
  if (R[2] + R[6] < 0) ax[2] = -ax[2];
  aa[0] = th * ax[0]; aa[1] = th * ax[1]; aa[2] = th * ax[2];
}

inline void check(int rc) {
  if (rc != BODYFIT_OK) throw std::runtime_error(std::string("bodyfit: ") + bodyfit_last_error());
}

class GaussianMixture {  // ark::GaussianMixture stand-in (pose_prior.txt contents)
 public:
  int nComps = 0;
  GaussianMixture() = default;
  GaussianMixture(int K, int D, const double* weights, const double* means, const double* covs, int device = 0)
      : nComps(K) { check(bodyfit_gmm_create(K, D, weights, means, covs, std::sqrt(0.5), device, &h_)); }
  ~GaussianMixture() { bodyfit_gmm_destroy(h_); }
  GaussianMixture(const GaussianMixture&) = delete;
  GaussianMixture& operator=(const GaussianMixture&) = delete;
  const bodyfit_gmm* handle() const { return h_; }
 private:
  bodyfit_gmm* h_ = nullptr;
};

class AvatarModel {  // ark::AvatarModel stand-in: owns the device-resident SMPL tensors
 public:
  std::vector<int> parent;
  bool useJointShapeRegressor = true;
  AvatarModel(const bodyfit_model_desc& desc, int device = 0)
      : parent(desc.parent, desc.parent + desc.n_joints), nJ_(desc.n_joints), nS_(desc.n_shape), nV_(desc.n_verts) {
    check(bodyfit_model_create(&desc, device, &h_));
  }
  ~AvatarModel() { bodyfit_model_destroy(h_); }
  AvatarModel(const AvatarModel&) = delete;
  AvatarModel& operator=(const AvatarModel&) = delete;
  int numJoints() const { return nJ_; }
  int numShapeKeys() const { return nS_; }
  int numPoints() const { return nV_; }
  const bodyfit_model* handle() const { return h_; }
 private:
  bodyfit_model* h_ = nullptr;
  int nJ_, nS_, nV_;
};

class Avatar {  // ark::Avatar stand-in: w (shape), p (root position), r (local rotations), update()
 public:
  const AvatarModel& model;
  std::vector<double> w;          // [nS]
  std::array<double, 3> p{0, 0, 0};
  std::vector<Matrix3d> r;        // [nJ]
  std::vector<double> jointPos;   // [nJ][3]
  std::vector<float> cloud;       // [V][3]
  explicit Avatar(const AvatarModel& m)
      : model(m), w(m.numShapeKeys(), 0.0), r(m.numJoints(), Identity3()), jointPos(3 * m.numJoints(), 0.0) {}

  // SMPL forward for the current w, p, r.  No Sim3 scale (the avatar has none: include/Sim3BA.h:497-507).
  void update(bool with_cloud = true) {
    const int nJ = model.numJoints();
    bodyfit_problem_desc d{};
    const int off[2] = {0, 0};
    d.n_frames = 1; d.kp_offset = off; d.n_cols = BODYFIT_FRAME_PARAMS + model.numShapeKeys();
    d.use_shape = 1; d.pose_blend = 1; d.R0 = r[0].data(); d.want_mesh = with_cloud ? 1 : 0; d.huber_delta = 3.0;
    bodyfit_problem* pr = nullptr;
    check(bodyfit_problem_create(model.handle(), &d, &pr));
    std::vector<double> x(BODYFIT_FRAME_PARAMS, 0.0);
    x[0] = 1.0; x[4] = p[0]; x[5] = p[1]; x[6] = p[2];
    for (int j = 1; j < nJ; ++j) MatrixToAngleAxis(r[j], &x[7 + 3 * (j - 1)]);
    if (with_cloud) cloud.resize((size_t)3 * model.numPoints());
    const int rc = bodyfit_forward(pr, x.data(), w.data(), jointPos.data(), with_cloud ? cloud.data() : nullptr);
    bodyfit_problem_destroy(pr);
    check(rc);
  }

  // update() of many avatars of one model in ONE device pass (each with its own w, p, r): what the write-back loops of
  // OptimizeMultiFrame and the drivers do frame by frame (include/MultiFrameBA.h:154-174, src/main_multi_frame.cpp:146)
  static void update_batch(const std::vector<Avatar*>& avs, bool with_cloud = true) {
    const int F = (int)avs.size();
    if (F == 0) return;
    const AvatarModel& model = avs[0]->model;
    const int nJ = model.numJoints(), nS = model.numShapeKeys();
    std::vector<int> off(F + 1, 0);
    std::vector<double> R0((size_t)F * 9), x((size_t)F * BODYFIT_FRAME_PARAMS, 0.0), w((size_t)F * nS);
    for (int f = 0; f < F; ++f) {
      std::copy(avs[f]->r[0].begin(), avs[f]->r[0].end(), R0.begin() + (size_t)f * 9);
      double* xf = &x[(size_t)f * BODYFIT_FRAME_PARAMS];
      xf[0] = 1.0; xf[4] = avs[f]->p[0]; xf[5] = avs[f]->p[1]; xf[6] = avs[f]->p[2];
      for (int j = 1; j < nJ; ++j) MatrixToAngleAxis(avs[f]->r[j], &xf[7 + 3 * (j - 1)]);
      std::copy(avs[f]->w.begin(), avs[f]->w.end(), w.begin() + (size_t)f * nS);
    }
    bodyfit_problem_desc d{};
    d.n_frames = F; d.kp_offset = off.data(); d.n_cols = BODYFIT_FRAME_PARAMS + nS;
    d.use_shape = 1; d.beta_per_frame = 1; d.pose_blend = 1; d.R0 = R0.data(); d.want_mesh = with_cloud ? 1 : 0;
    d.huber_delta = 3.0;
    bodyfit_problem* pr = nullptr;
    check(bodyfit_problem_create(model.handle(), &d, &pr));
    std::vector<double> joints((size_t)F * nJ * 3);
    std::vector<float> clouds(with_cloud ? (size_t)F * model.numPoints() * 3 : 0);
    const int rc = bodyfit_forward(pr, x.data(), w.data(), joints.data(), with_cloud ? clouds.data() : nullptr);
    bodyfit_problem_destroy(pr);
    check(rc);
    for (int f = 0; f < F; ++f) {
      avs[f]->jointPos.assign(joints.begin() + (size_t)f * nJ * 3, joints.begin() + (size_t)(f + 1) * nJ * 3);
      if (with_cloud)
        avs[f]->cloud.assign(clouds.begin() + (size_t)f * model.numPoints() * 3,
                             clouds.begin() + (size_t)(f + 1) * model.numPoints() * 3);
    }
  }
};

namespace detail {

inline std::string report(const bodyfit_fit_summary& s) {
  char buf[256];
  std::snprintf(buf, sizeof buf, "bodyfit LM: iterations %d (ok %d, rejected %d), cost %.6e -> %.6e, termination %s",
                s.iterations, s.n_successful, s.n_unsuccessful, s.initial_cost, s.final_cost,
                s.termination == 0 ? "CONVERGENCE" : (s.termination == 1 ? "NO_CONVERGENCE" : "FAILURE"));
  return buf;
}

inline std::vector<PixelKP> filter(const std::vector<PixelKP>& kps, const std::vector<int>& valid) {
  if (valid.empty()) return kps;
  std::vector<PixelKP> out;
  for (const auto& kp : kps)
    if (std::find(valid.begin(), valid.end(), kp.jid) != valid.end()) out.push_back(kp);   // include/Sim3BA.h:411-414
  return out;
}

// shared body of the two single-frame entry points
inline std::pair<bool, std::string> single(const AvatarModel& model, Avatar& avatar, const std::vector<PixelKP>& kps_in,
                                           double fx, double fy, double cx, double cy,
                                           const std::vector<int>& valid_joint_ids, Sim3Params& initSim3, int max_iters,
                                           double betaPose, double betaShape, const GaussianMixture* gmm,
                                           bool shape_in_residual, bool freeze_unobserved) {
  const int nJ = model.numJoints(), nS = model.numShapeKeys();
  const std::vector<PixelKP> kps = filter(kps_in, valid_joint_ids);
  std::vector<int> ids;
  std::vector<double> uv;
  for (const auto& kp : kps) { ids.push_back(kp.jid); uv.push_back(kp.u); uv.push_back(kp.v); }
  const int off[2] = {0, (int)ids.size()};
  const bool shape_block = betaShape > 0.0 && nS > 0;          // include/Sim3BA.h:428,466 / :630-637
  bodyfit_problem_desc d{};
  d.n_frames = 1; d.kp_offset = off; d.kp_id = ids.data(); d.kp_uv = uv.data();
  d.fx = fx; d.fy = fy; d.cx = cx; d.cy = cy;
  d.R0 = avatar.r[0].data();                                   // fixed initial root orientation (:395,558)
  d.n_cols = BODYFIT_FRAME_PARAMS + (shape_block ? nS : 0);
  d.use_shape = (shape_in_residual && shape_block && model.useJointShapeRegressor) ? 1 : 0;   // :417
  d.pose_blend = 1;
  d.beta_pose = betaPose > 0.0 ? betaPose : 0.0;
  d.gmm = (gmm && gmm->nComps > 0) ? gmm->handle() : nullptr;
  d.beta_shape = shape_block ? betaShape : 0.0;
  d.huber_delta = 3.0;                                         // :407,570
  bodyfit_problem* pr = nullptr;
  check(bodyfit_problem_create(model.handle(), &d, &pr));
  std::vector<double> x(BODYFIT_FRAME_PARAMS, 0.0);            // joints start from zero (:401-404)
  x[0] = initSim3.scale();
  for (int i = 0; i < 3; ++i) { x[1 + i] = initSim3.aa_root()[i]; x[4 + i] = initSim3.trans()[i]; }
  std::vector<unsigned char> constant(BODYFIT_FRAME_PARAMS, 0);
  if (freeze_unobserved)                                       // :608-611
    for (int j : {10, 11, 22, 23})
      if (nJ > j) for (int i = 0; i < 3; ++i) constant[7 + 3 * (j - 1) + i] = 1;
  bodyfit_fit_options opt{max_iters, 0.3, 3.0, 0, 0};          // bounds :450-451,613-614
  bodyfit_fit_summary sum{};
  const int rc = bodyfit_solve(pr, x.data(), shape_block ? avatar.w.data() : nullptr, constant.data(), 1, &opt, &sum, 1);
  bodyfit_problem_destroy(pr);
  check(rc);
  // write-back (include/Sim3BA.h:481-507,649-679): r[0] <- R(rootAA) r[0]; r[j] <- R(aa_j); p <- rootT
  avatar.r[0] = Mul(AngleAxisToMatrix(&x[1]), avatar.r[0]);
  for (int j = 1; j < nJ; ++j) avatar.r[j] = AngleAxisToMatrix(&x[7 + 3 * (j - 1)]);
  avatar.p = {x[4], x[5], x[6]};
  initSim3.scale() = x[0];
  for (int i = 0; i < 3; ++i) { initSim3.aa_root()[i] = x[1 + i]; initSim3.trans()[i] = x[4 + i]; }
  return {sum.usable != 0, report(sum)};
}

}  // namespace detail

// include/Sim3BA.h:348-511
inline std::pair<bool, std::string> OptimizePoseShapeReprojection(
    const AvatarModel& model, Avatar& avatar, const std::vector<PixelKP>& kps, double fx, double fy, double cx, double cy,
    const std::vector<int>& valid_joint_ids, Sim3Params& initSim3, int max_iters = 100, double betaPose = 0.0,
    double betaShape = 0.0, const GaussianMixture* gmmPosePrior = nullptr) {
  return detail::single(model, avatar, kps, fx, fy, cx, cy, valid_joint_ids, initSim3, max_iters, betaPose, betaShape,
                        gmmPosePrior, /*shape_in_residual=*/true, /*freeze_unobserved=*/false);
}

// include/Sim3BA.h:515-683
inline std::pair<bool, std::string> OptimizePoseReprojection(
    const AvatarModel& model, Avatar& avatar, const std::vector<PixelKP>& kps, double fx, double fy, double cx, double cy,
    const std::vector<int>& valid_joint_ids, Sim3Params& initSim3, int max_iters = 100, double betaPose = 0.0,
    double betaShape = 0.0, const GaussianMixture* gmmPosePrior = nullptr) {
  return detail::single(model, avatar, kps, fx, fy, cx, cy, valid_joint_ids, initSim3, max_iters, betaPose, betaShape,
                        gmmPosePrior, /*shape_in_residual=*/false, /*freeze_unobserved=*/true);
}

// include/MultiFrameBA.h:33-177
inline std::pair<bool, std::string> OptimizeMultiFrame(
    const AvatarModel& model, const std::vector<Avatar*>& avatars, const std::vector<std::vector<PixelKP>>& kps_vec,
    double fx, double fy, double cx, double cy, const std::vector<int>& valid_ids, std::vector<FramePoseParams>& poses,
    double betaPose, double betaShape, double lambdaTemp, int max_iters = 100) {
  const int F = (int)avatars.size(), nJ = model.numJoints(), nS = model.numShapeKeys();
  std::vector<int> off(F + 1, 0), ids;
  std::vector<double> uv, R0((size_t)F * 9);
  for (int f = 0; f < F; ++f) {
    for (const auto& kp : detail::filter(kps_vec[f], valid_ids)) { ids.push_back(kp.jid); uv.push_back(kp.u); uv.push_back(kp.v); }
    off[f + 1] = (int)ids.size();
    std::copy(avatars[f]->r[0].begin(), avatars[f]->r[0].end(), R0.begin() + (size_t)f * 9);   // :87
  }
  bodyfit_problem_desc d{};
  d.n_frames = F; d.kp_offset = off.data(); d.kp_id = ids.data(); d.kp_uv = uv.data();
  d.fx = fx; d.fy = fy; d.cx = cx; d.cy = cy; d.R0 = R0.data();
  d.n_cols = BODYFIT_FRAME_PARAMS + nS;                        // the shape block is always present (:95,100)
  d.use_shape = betaShape > 0.0 ? 1 : 0;                       // jointShapeReg only when betaShape > 0 (:88)
  d.pose_blend = 1;
  d.beta_pose = (betaPose > 0.0 && nJ > 1) ? betaPose : 0.0;   // L2 fallback: gmm = nullptr (:109)
  d.beta_shape = (nS > 0 && betaShape > 0.0) ? betaShape : 0.0;   // :115-118
  d.lambda_temporal = (lambdaTemp > 0.0 && F > 1) ? lambdaTemp : 0.0;   // :121
  d.huber_delta = 3.0;                                         // :64
  bodyfit_problem* pr = nullptr;
  check(bodyfit_problem_create(model.handle(), &d, &pr));
  std::vector<double> x((size_t)F * BODYFIT_FRAME_PARAMS);
  for (int f = 0; f < F; ++f) {
    double* xf = &x[(size_t)f * BODYFIT_FRAME_PARAMS];
    xf[0] = poses[f].scale;
    for (int i = 0; i < 3; ++i) { xf[1 + i] = poses[f].rootAA[i]; xf[4 + i] = poses[f].rootT[i]; }
    for (int j = 1; j < nJ; ++j)
      for (int i = 0; i < 3; ++i) xf[7 + 3 * (j - 1) + i] = poses[f].jointAA[j][i];
  }
  double* w_block = avatars.front()->w.data();                 // shared shape block (:67)
  bodyfit_fit_options opt{max_iters, 0.3, 3.0, 0, 0};
  opt.scale_lo = -1e300; opt.scale_hi = 1e300;                 // the multi-frame problem sets no bounds
  bodyfit_fit_summary sum{};
  const int rc = bodyfit_solve(pr, x.data(), w_block, nullptr, 0, &opt, &sum, 1);
  bodyfit_problem_destroy(pr);
  check(rc);
  for (int f = 0; f < F; ++f) {                                // write-back + update() (:154-174)
    const double* xf = &x[(size_t)f * BODYFIT_FRAME_PARAMS];
    poses[f].scale = xf[0];
    for (int i = 0; i < 3; ++i) { poses[f].rootAA[i] = xf[1 + i]; poses[f].rootT[i] = xf[4 + i]; }
    for (int j = 1; j < nJ; ++j)
      for (int i = 0; i < 3; ++i) poses[f].jointAA[j][i] = xf[7 + 3 * (j - 1) + i];
    avatars[f]->r[0] = Mul(AngleAxisToMatrix(poses[f].rootAA), avatars[f]->r[0]);
    avatars[f]->p = {poses[f].rootT[0], poses[f].rootT[1], poses[f].rootT[2]};
    for (int j = 1; j < nJ; ++j) avatars[f]->r[j] = AngleAxisToMatrix(poses[f].jointAA[j].data());
  }
  Avatar::update_batch(avatars);                               // every avatar's update(), one device pass
  return {sum.usable != 0, detail::report(sum)};
}

// include/Utils.h:102-115
inline double mean_pixel_error(const std::vector<PixelKP>& kps, const Avatar& avatar, double fx, double fy, double cx,
                               double cy) {
  if (kps.empty()) return 0.0;
  std::vector<int> ids;
  std::vector<double> uv;
  for (const auto& kp : kps) { ids.push_back(kp.jid); uv.push_back(kp.u); uv.push_back(kp.v); }
  return bodyfit_mean_pixel_error((int)ids.size(), ids.data(), uv.data(), avatar.jointPos.data(), fx, fy, cx, cy);
}

// ---- mesh overlay: include/RenderSMPLMesh.h ------------------------------------------------------------------
// cv::Mat stand-in for an 8-bit 3-channel image the caller owns (data, rows, cols, step in bytes).
struct ImageView {
  unsigned char* data;
  int rows, cols;
  size_t step;
};

// A reusable overlay context for one face list and image size (the device buffers live here).
class MeshOverlay {
 public:
  MeshOverlay(const std::vector<std::array<int, 3>>& faces, int n_vertices, int width, int height, int max_frames = 1,
              int device = 0) {
    bodyfit_overlay_desc d{};
    d.device = device; d.n_vertices = n_vertices; d.n_faces = (int)faces.size();
    d.faces = faces.empty() ? nullptr : faces[0].data();
    d.width = width; d.height = height; d.max_frames = max_frames;
    check(bodyfit_overlay_create(&d, &h_));
  }
  ~MeshOverlay() { bodyfit_overlay_destroy(h_); }
  MeshOverlay(const MeshOverlay&) = delete;
  MeshOverlay& operator=(const MeshOverlay&) = delete;
  bodyfit_overlay* handle() const { return h_; }
  // one frame, host buffers; cloud: x, y, z per vertex (Avatar::cloud, or the data() of the reference's 3xN matrix)
  template <typename T>
  void render(const T* cloud, ImageView img, double fx, double fy, double cx, double cy, bool fill = true,
              bool backface_cull = true, bool wireframe = false) {
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "float or double vertices");
    check(bodyfit_overlay_render(h_, cloud, sizeof(T) == 8, 0, 1, img.data, img.step, img.step * (size_t)img.rows, fx, fy,
                                 cx, cy, fill, backface_cull, wireframe));
  }
 private:
  bodyfit_overlay* h_ = nullptr;
};

}  // namespace bodyfit

namespace smpl {
namespace render {
// include/RenderSMPLMesh.h:16-24, same argument order; `cloud` is the avatar's vertex buffer (3 values per vertex).
template <typename T>
inline void renderSMPLMesh(const std::vector<T>& cloud, const std::vector<std::array<int, 3>>& faces,
                           bodyfit::ImageView img, double fx, double fy, double cx, double cy, bool fill = true,
                           bool backface_cull = true, bool wireframe = false) {
  bodyfit::MeshOverlay ov(faces, (int)(cloud.size() / 3), img.cols, img.rows);
  ov.render(cloud.data(), img, fx, fy, cx, cy, fill, backface_cull, wireframe);
}
}  // namespace render
}  // namespace smpl
