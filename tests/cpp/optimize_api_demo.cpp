// Drives include/bodyfit.hpp exactly the way src/main_multi_frame.cpp / main_single_frame.cpp drive the
// reference API: avatars at (0,0,3) with r[0] = Ry(pi) diag(1,-1,1), zero poses, then OptimizeMultiFrame
// and OptimizePoseShapeReprojection.  Inputs come from a binary blob written by tests/test_gpu_cpp_api.py;
// results go back as raw doubles.
#include <array>
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <vector>

#include "bodyfit.hpp"

template <typename T>
static std::vector<T> rd(std::ifstream& f, size_t n) {
  std::vector<T> v(n);
  f.read(reinterpret_cast<char*>(v.data()), n * sizeof(T));
  return v;
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  std::ifstream f(argv[1], std::ios::binary);
  auto h = rd<int32_t>(f, 7);
  const int V = h[0], nJ = h[1], nS = h[2], P = h[3], nL = h[4], F = h[5], K = h[6];
  auto vt = rd<double>(f, (size_t)V * 3), sd = rd<double>(f, (size_t)V * 3 * nS), pd = rd<double>(f, (size_t)V * 3 * P),
       jr = rd<double>(f, (size_t)nJ * V), w = rd<double>(f, (size_t)V * nJ);
  auto parent = rd<int32_t>(f, nJ), lvid = rd<int32_t>(f, nL), kp_off = rd<int32_t>(f, F + 1), kp_id = rd<int32_t>(f, K);
  auto kp_uv = rd<double>(f, (size_t)K * 2), intr = rd<double>(f, 4);
  bodyfit_model_desc md{V, nJ, nS, P, vt.data(), sd.data(), pd.data(), jr.data(), w.data(), parent.data(), nL, lvid.data()};
  try {
    bodyfit::AvatarModel model(md, 0);
    const bodyfit::Matrix3d R0 = {-1, 0, 0, 0, -1, 0, 0, 0, -1};   // Ry(pi) diag(1,-1,1)
    std::vector<bodyfit::Avatar> store;
    store.reserve(F);
    std::vector<bodyfit::Avatar*> avatars;
    std::vector<bodyfit::FramePoseParams> poses;
    std::vector<std::vector<bodyfit::PixelKP>> kps(F);
    for (int i = 0; i < F; ++i) {
      store.emplace_back(model);
      store.back().p = {0, 0, 3};
      store.back().r[0] = R0;
      avatars.push_back(&store.back());
      bodyfit::FramePoseParams Pp;
      Pp.scale = 1.0;
      for (int k = 0; k < 3; ++k) { Pp.rootAA[k] = 0; Pp.rootT[k] = k == 2 ? 3.0 : 0.0; }
      Pp.jointAA.assign(nJ, {0.0, 0.0, 0.0});
      poses.push_back(Pp);
      for (int k = kp_off[i]; k < kp_off[i + 1]; ++k) kps[i].push_back({kp_id[k], kp_uv[2 * k], kp_uv[2 * k + 1]});
    }
    auto [ok, rep] = bodyfit::OptimizeMultiFrame(model, avatars, kps, intr[0], intr[1], intr[2], intr[3], {}, poses, 5.0,
                                                 25.0, 3.0, 30);
    std::cout << "multi: " << (ok ? "OK " : "FAIL ") << rep << "\n";
    std::ofstream o(argv[2], std::ios::binary);
    for (int i = 0; i < F; ++i) {
      std::vector<double> x(76);
      x[0] = poses[i].scale;
      for (int k = 0; k < 3; ++k) { x[1 + k] = poses[i].rootAA[k]; x[4 + k] = poses[i].rootT[k]; }
      for (int j = 1; j < nJ; ++j)
        for (int k = 0; k < 3; ++k) x[7 + 3 * (j - 1) + k] = poses[i].jointAA[j][k];
      o.write(reinterpret_cast<const char*>(x.data()), 76 * sizeof(double));
    }
    o.write(reinterpret_cast<const char*>(avatars.front()->w.data()), nS * sizeof(double));
    std::vector<bodyfit::PixelKP> joints_only;   // mean_pixel_error indexes jointPos by jid (include/Utils.h:109)
    for (const auto& kp : kps[0])
      if (kp.jid < nJ) joints_only.push_back(kp);
    const double px = bodyfit::mean_pixel_error(joints_only, *avatars[0], intr[0], intr[1], intr[2], intr[3]);
    o.write(reinterpret_cast<const char*>(&px), sizeof(double));
    o.write(reinterpret_cast<const char*>(avatars[0]->jointPos.data()), 3 * nJ * sizeof(double));
    // single-frame entry point on frame 0 with a fresh avatar
    bodyfit::Avatar single(model);
    single.p = {0, 0, 3};
    single.r[0] = R0;
    bodyfit::Sim3Params s3{};
    s3.scale() = 1.0;
    s3.trans()[2] = 3.0;
    auto [ok2, rep2] = bodyfit::OptimizePoseShapeReprojection(model, single, kps[0], intr[0], intr[1], intr[2], intr[3], {},
                                                              s3, 40, 20.0, 30.0, nullptr);
    std::cout << "single: " << (ok2 ? "OK " : "FAIL ") << rep2 << "\n";
    o.write(reinterpret_cast<const char*>(s3.data), 7 * sizeof(double));
    o.write(reinterpret_cast<const char*>(single.w.data()), nS * sizeof(double));
    if (argc >= 5) {
      // the overlay of the fitted avatar, as src/main_single_frame.cpp:273-275 draws it (here on a 640x360 image)
      std::ifstream ff(argv[3], std::ios::binary);
      auto nf = rd<int32_t>(ff, 1);
      auto fraw = rd<int32_t>(ff, (size_t)nf[0] * 3);
      std::vector<std::array<int, 3>> faces(nf[0]);
      for (int i = 0; i < nf[0]; ++i) faces[i] = {fraw[3 * i], fraw[3 * i + 1], fraw[3 * i + 2]};
      single.update();
      const int W = 640, H = 360;
      std::vector<unsigned char> img((size_t)W * H * 3, 17);
      smpl::render::renderSMPLMesh(single.cloud, faces, bodyfit::ImageView{img.data(), H, W, (size_t)W * 3}, intr[0] / 3,
                                   intr[1] / 3, intr[2] / 3, intr[3] / 3, /*fill=*/true, /*backface_cull=*/true,
                                   /*wireframe=*/false);
      std::ofstream oi(argv[4], std::ios::binary);
      oi.write(reinterpret_cast<const char*>(single.cloud.data()), single.cloud.size() * sizeof(float));
      oi.write(reinterpret_cast<const char*>(img.data()), img.size());
    }
    return (ok && ok2) ? 0 : 1;
  } catch (const std::exception& e) {
    std::cerr << e.what() << "\n";
    return 3;
  }
}
