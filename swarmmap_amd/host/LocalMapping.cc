// LocalMapping.cc — see LocalMapping.h.
#include "LocalMapping.h"

#include <stdexcept>
#include <string>

namespace ORB_SLAM2 {

namespace {
void check(int rc, const char* what) {
    if (rc != SO_OK) throw std::runtime_error(std::string(what) + ": " + so_last_error());
}
so_tri_keyframe view(const TriangulationKeyFrame& k) {
    so_tri_keyframe t{};
    for (int i = 0; i < 12; i++) t.Tcw[i] = k.Tcw[i];
    t.fx = k.fx; t.fy = k.fy; t.cx = k.cx; t.cy = k.cy;
    t.invfx = 1.0f / k.fx;
    t.invfy = 1.0f / k.fy;
    t.scale_factors = k.mvScaleFactors.data();
    t.level_sigma2 = k.mvLevelSigma2.data();
    t.nlevels = (int32_t)k.mvScaleFactors.size();
    return t;
}
}  // namespace

LocalMappingOps::LocalMappingOps(int device) { check(so_matcher_create(device, &handle_), "so_matcher_create"); }
LocalMappingOps::~LocalMappingOps() { so_matcher_destroy(handle_); }

int LocalMappingOps::TriangulateMatches(const TriangulationKeyFrame& current, const std::vector<TriangulationKeyFrame>& neighbours,
                                        float ratioFactor, const TriangulationMatches& mt, std::vector<uint8_t>& ok,
                                        std::vector<float>& x3D) {
    const so_tri_keyframe k1 = view(current);
    std::vector<so_tri_keyframe> k2;
    for (const TriangulationKeyFrame& k : neighbours) k2.push_back(view(k));
    const int n = mt.size();
    ok.assign((size_t)n, 0);
    x3D.assign(3 * (size_t)n, 0.f);
    check(so_triangulate_matches(handle_, &k1, (int32_t)k2.size(), k2.data(), ratioFactor, n, mt.neighbour.data(), mt.xy1.data(),
                                 mt.octave1.data(), mt.xy2.data(), mt.octave2.data(), ok.data(), x3D.data()),
          "so_triangulate_matches");
    int nnew = 0;
    for (uint8_t f : ok) nnew += f;
    return nnew;
}

int LocalMappingOps::CreateNewPoints(const TriangulationKeyFrame& current, const std::vector<TriangulationKeyFrame>& neighbours,
                                     float ratioFactor, const TriangulationMatches& mt, std::vector<uint8_t>& ok, std::vector<float>& x3D,
                                     std::vector<float>& normal, std::vector<float>& maxDistance, std::vector<float>& minDistance) {
    const so_tri_keyframe k1 = view(current);
    std::vector<so_tri_keyframe> k2;
    for (const TriangulationKeyFrame& k : neighbours) k2.push_back(view(k));
    const int n = mt.size();
    ok.assign((size_t)n, 0);
    x3D.assign(3 * (size_t)n, 0.f);
    normal.assign(3 * (size_t)n, 0.f);
    maxDistance.assign((size_t)n, 0.f);
    minDistance.assign((size_t)n, 0.f);
    check(so_triangulate_new_points(handle_, &k1, (int32_t)k2.size(), k2.data(), ratioFactor, n, mt.neighbour.data(), mt.xy1.data(),
                                    mt.octave1.data(), mt.xy2.data(), mt.octave2.data(), ok.data(), x3D.data(), normal.data(),
                                    maxDistance.data(), minDistance.data()),
          "so_triangulate_new_points");
    int nnew = 0;
    for (uint8_t f : ok) nnew += f;
    return nnew;
}

void LocalMappingOps::UpdateNormalAndDepth(const std::vector<int32_t>& offsets, const std::vector<float>& obsOw,
                                           const std::vector<float>& Xw, const std::vector<float>& refOw,
                                           const std::vector<float>& refLevelScale, const std::vector<float>& refLastScale,
                                           std::vector<float>& normal, std::vector<float>& maxDistance, std::vector<float>& minDistance) {
    const int n = (int)offsets.size() - 1;
    if (n <= 0) return;
    check(so_update_normal_and_depth(handle_, n, offsets.data(), obsOw.data(), Xw.data(), refOw.data(), refLevelScale.data(),
                                     refLastScale.data(), normal.data(), maxDistance.data(), minDistance.data()),
          "so_update_normal_and_depth");
}

}  // namespace ORB_SLAM2
