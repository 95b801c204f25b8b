// adapter_smoke.cpp — drives every method of the C++ adapter classes (the reference's signatures, swarmmap_amd/host/)
// end to end on the GPU.  Inputs are raw arrays written by tests/test_cpp_adapters_gpu.py into a directory; results
// are printed as "name count fnv-hash" lines (or plain numbers) that the test compares with the CPU ORACLE run on the
// same inputs.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../swarmmap_amd/host/ORBextractor.h"
#include "../../swarmmap_amd/host/ORBmatcher.h"
#include "../../swarmmap_amd/host/Frame.h"
#include "../../swarmmap_amd/host/Optimizer.h"
#include "../../swarmmap_amd/host/LocalMapping.h"

static uint64_t fnv(const void* p, size_t n, uint64_t h = 1469598103934665603ull) {
    const uint8_t* b = (const uint8_t*)p;
    for (size_t i = 0; i < n; i++) h = (h ^ b[i]) * 1099511628211ull;
    return h;
}

static std::string g_dir;
template <typename T>
static std::vector<T> rd(const char* name) {
    const std::string path = g_dir + "/" + name + ".bin";
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { fprintf(stderr, "missing %s\n", path.c_str()); exit(4); }
    fseek(f, 0, SEEK_END);
    const long bytes = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<T> v((size_t)bytes / sizeof(T));
    if (bytes > 0 && fread(v.data(), 1, (size_t)bytes, f) != (size_t)bytes) exit(5);
    fclose(f);
    return v;
}
template <typename T>
static void out(const char* name, const std::vector<T>& v) {
    printf("%s %zu %016llx\n", name, v.size(), (unsigned long long)fnv(v.data(), v.size() * sizeof(T)));
}

struct FrameArrays {
    std::vector<float> x, y, angle, sf;
    std::vector<int32_t> octave;
    std::vector<uint8_t> desc, excluded;
    std::vector<float> bounds;
    so_frame_view view(bool with_excluded) const {
        so_frame_view F{};
        F.n = (int32_t)x.size(); F.x = x.data(); F.y = y.data(); F.octave = octave.data(); F.angle = angle.data();
        F.desc = desc.data(); F.excluded = with_excluded && !excluded.empty() ? excluded.data() : nullptr;
        F.min_x = bounds[0]; F.max_x = bounds[1]; F.min_y = bounds[2]; F.max_y = bounds[3];
        F.grid_inv_w = 64.0f / (bounds[1] - bounds[0]); F.grid_inv_h = 48.0f / (bounds[3] - bounds[2]);
        F.scale_factors = sf.data(); F.nlevels = (int32_t)sf.size();
        return F;
    }
};
static FrameArrays frame(const std::string& p) {
    FrameArrays f;
    f.x = rd<float>((p + "_x").c_str()); f.y = rd<float>((p + "_y").c_str()); f.angle = rd<float>((p + "_angle").c_str());
    f.octave = rd<int32_t>((p + "_octave").c_str()); f.desc = rd<uint8_t>((p + "_desc").c_str());
    f.excluded = rd<uint8_t>((p + "_excluded").c_str()); f.bounds = rd<float>((p + "_bounds").c_str());
    f.sf = rd<float>("scale_factors");
    return f;
}
static ORB_SLAM2::ORBmatcher::WindowQueries queries(const std::string& p) {
    ORB_SLAM2::ORBmatcher::WindowQueries q;
    q.valid = rd<uint8_t>((p + "_valid").c_str()); q.u = rd<float>((p + "_u").c_str()); q.v = rd<float>((p + "_v").c_str());
    q.radius = rd<float>((p + "_radius").c_str()); q.pred_level = rd<int32_t>((p + "_pred_level").c_str());
    q.min_level = rd<int32_t>((p + "_min_level").c_str()); q.max_level = rd<int32_t>((p + "_max_level").c_str());
    q.desc = rd<uint8_t>((p + "_desc").c_str()); q.angle = rd<float>((p + "_angle").c_str());
    return q;
}
struct FeatVec {
    std::vector<int32_t> node, off, idx;
    so_featvec view() const { return so_featvec{(int32_t)node.size(), node.data(), off.data(), idx.data()}; }
};
static FeatVec featvec(const std::string& p) {
    FeatVec f;
    f.node = rd<int32_t>((p + "_node").c_str()); f.off = rd<int32_t>((p + "_off").c_str()); f.idx = rd<int32_t>((p + "_idx").c_str());
    return f;
}

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    g_dir = argv[1];
    const int w = atoi(argv[2]), h = atoi(argv[3]);
    // ---- ORBextractor::operator() -----------------------------------------------------------------------------
    std::vector<uint8_t> img = rd<uint8_t>("image");
    ORB_SLAM2::ORBextractor ex(1000, 1.2f, 8, 20, 7);
    std::vector<swarmorb::KeyPoint> kps;
    swarmorb::Descriptors desc;
    swarmorb::ImageView view{img.data(), w, h, w}, mask;
    ex(view, mask, kps, desc);
    ex(view, mask, kps, desc);  // the context is reused frame after frame
    printf("levels %d scale7 %.6f\n", ex.GetLevels(), ex.GetScaleFactors()[7]);
    out("keypoints", kps);
    out("descriptors", desc.data);
    printf("distance %d\n", ORB_SLAM2::ORBmatcher::DescriptorDistance(desc.ptr(0), desc.ptr(1)));
    // ---- Frame post-processing (EuRoC lens) -------------------------------------------------------------------
    const float K[4] = {458.654f, 457.296f, 367.215f, 248.375f};
    ORB_SLAM2::Frame fr(K, {-0.28340811f, 0.07395907f, 0.00019359f, 1.76187114e-05f});
    std::vector<swarmorb::KeyPoint> keysUn;
    fr.UndistortAndAssign(kps, w, h, keysUn);
    std::vector<float> un(2 * keysUn.size());
    for (size_t i = 0; i < keysUn.size(); i++) { un[2 * i] = keysUn[i].pt.x; un[2 * i + 1] = keysUn[i].pt.y; }
    out("undistorted", un);
    printf("bounds %.9g %.9g %.9g %.9g\n", fr.mnMinX, fr.mnMaxX, fr.mnMinY, fr.mnMaxY);
    out("grid", fr.GridItems());
    // ---- ORBmatcher: the ten routines -------------------------------------------------------------------------
    const FrameArrays F1 = frame("m1f"), F2 = frame("m2f"), I1 = frame("i1"), I2 = frame("i2"), KF = frame("kf");
    {   // SearchByProjection(Frame&, vpMapPoints, th)
        ORB_SLAM2::ORBmatcher m(0.8f, true);
        ORB_SLAM2::MapPointViews mp;
        mp.in_view = rd<uint8_t>("m1_in_view"); mp.proj_x = rd<float>("m1_proj_x"); mp.proj_y = rd<float>("m1_proj_y");
        mp.view_cos = rd<float>("m1_view_cos"); mp.pred_level = rd<int32_t>("m1_pred_level"); mp.desc = rd<uint8_t>("m1_desc");
        mp.has_obs = rd<uint8_t>("m1_has_obs");
        std::vector<int32_t> k;
        const int n = m.SearchByProjection(F1.view(true), mp, 1.0f, k);
        printf("m1_n %d\n", n); out("m1", k);
    }
    {   // SearchByProjection(CurrentFrame, LastFrame, th, mono)
        ORB_SLAM2::ORBmatcher m(0.9f, true);
        ORB_SLAM2::LastFrameViews L;
        L.valid = rd<uint8_t>("m2_valid"); L.u = rd<float>("m2_u"); L.v = rd<float>("m2_v"); L.angle = rd<float>("m2_angle");
        L.octave = rd<int32_t>("m2_octave"); L.desc = rd<uint8_t>("m2_desc"); L.has_obs = rd<uint8_t>("m2_has_obs");
        std::vector<int32_t> k;
        const int n = m.SearchByProjection(F2.view(true), L, 15.0f, k);
        printf("m2_n %d\n", n); out("m2", k);
    }
    {   // SearchForInitialization
        ORB_SLAM2::ORBmatcher m(0.9f, true);
        std::vector<float> prev = rd<float>("i_prev");
        std::vector<int32_t> m12;
        const int n = m.SearchForInitialization(I1.view(false), I2.view(false), prev, m12, 100);
        printf("m4_n %d\n", n); out("m4", m12); out("m4_prev", prev);
    }
    const std::vector<float> b1x = rd<float>("b1_x"), b1y = rd<float>("b1_y"), b1a = rd<float>("b1_angle");
    const std::vector<uint8_t> b1d = rd<uint8_t>("b1_desc"), b1v = rd<uint8_t>("b1_valid"), b1f = rd<uint8_t>("b1_free");
    const std::vector<float> b2x = rd<float>("b2_x"), b2y = rd<float>("b2_y"), b2a = rd<float>("b2_angle");
    const std::vector<int32_t> b2o = rd<int32_t>("b2_octave");
    const std::vector<uint8_t> b2d = rd<uint8_t>("b2_desc"), b2v = rd<uint8_t>("b2_valid"), b2f = rd<uint8_t>("b2_free");
    const FeatVec fv1 = featvec("fv1"), fv2 = featvec("fv2");
    for (int variant = 0; variant < 2; variant++) {  // SearchByBoW(KF, F) and (KF, KF)
        ORB_SLAM2::ORBmatcher m(0.7f, true);
        std::vector<int32_t> m2, m1;
        const int n = m.SearchByBoW(variant, (int)b1a.size(), b1d.data(), b1a.data(), b1v.data(), fv1.view(), (int)b2a.size(),
                                    b2d.data(), b2a.data(), b2v.data(), fv2.view(), m2, m1);
        printf("m3_%d_n %d\n", variant, n);
        out(variant ? "m3_1_of2" : "m3_0_of2", m2); out(variant ? "m3_1_of1" : "m3_0_of1", m1);
    }
    {   // SearchForTriangulation
        ORB_SLAM2::ORBmatcher m(0.6f, true);
        ORB_SLAM2::ORBmatcher::KeyFrameFeatures k1, k2;
        k1.n = (int)b1x.size(); k1.x = b1x.data(); k1.y = b1y.data(); k1.angle = b1a.data(); k1.desc = b1d.data(); k1.free_ = b1f.data();
        k2.n = (int)b2x.size(); k2.x = b2x.data(); k2.y = b2y.data(); k2.angle = b2a.data(); k2.octave = b2o.data();
        k2.desc = b2d.data(); k2.free_ = b2f.data();
        const std::vector<float> F12 = rd<float>("F12"), sf = rd<float>("scale_factors"), ls = rd<float>("level_sigma2");
        std::vector<std::pair<size_t, size_t>> pairs;
        const int n = m.SearchForTriangulation(k1, fv1.view(), k2, fv2.view(), F12.data(), 900.0f, 240.0f, sf, ls, pairs);
        std::vector<int32_t> m12(b1x.size(), -1);
        for (auto& pr : pairs) m12[pr.first] = (int32_t)pr.second;
        printf("m5_n %d\n", n); out("m5", m12);
    }
    const ORB_SLAM2::ORBmatcher::WindowQueries q = queries("q"), q21 = queries("q21");
    const std::vector<float> inv = rd<float>("inv_sigma2");
    {   // Fuse x 2, SearchBySim3
        ORB_SLAM2::ORBmatcher m(0.6f, true);
        std::vector<int32_t> bi, bd;
        const int n1 = m.Fuse(KF.view(false), q, inv, bi, bd);
        printf("fuse_gate_n %d\n", n1); out("fuse_gate_idx", bi); out("fuse_gate_dist", bd);
        const int n2 = m.Fuse(KF.view(false), q, bi, bd);
        printf("fuse_scw_n %d\n", n2); out("fuse_scw_idx", bi); out("fuse_scw_dist", bd);
        std::vector<int32_t> m12;
        const int n3 = m.SearchBySim3(I1.view(false), KF.view(false), q, q21, m12);
        printf("sim3_n %d\n", n3); out("sim3", m12);
    }
    {   // the two sequential SearchByProjection overloads
        ORB_SLAM2::ORBmatcher m(0.75f, true);
        std::vector<int32_t> k;
        const int n1 = m.SearchByProjection(KF.view(true), q, k);
        printf("greedy_kf_n %d\n", n1); out("greedy_kf", k);
        const int n2 = m.SearchByProjection(KF.view(true), q, 64, k);
        printf("greedy_f_n %d\n", n2); out("greedy_f", k);
        const std::vector<int32_t> off = rd<int32_t>("dd_off");
        const std::vector<uint8_t> dd = rd<uint8_t>("dd_desc");
        out("distinctive", m.ComputeDistinctiveDescriptors(off, dd));
    }
    {   // the same five routines with the projection on the device (map points + pose in, bindings out)
        typedef ORB_SLAM2::ORBmatcher M;
        auto points = [](const std::string& p) {
            M::MapPointFields f;
            f.Xw = rd<float>((p + "_Xw").c_str()); f.normal = rd<float>((p + "_normal").c_str());
            f.max_dist = rd<float>((p + "_max_dist").c_str()); f.min_dist = rd<float>((p + "_min_dist").c_str());
            f.desc = rd<uint8_t>((p + "_desc").c_str()); f.valid = rd<uint8_t>((p + "_valid").c_str());
            return f;
        };
        const std::vector<float> cal4 = rd<float>("pj_cam"), lsf = rd<float>("pj_lsf");
        M::Calibration cal{cal4[0], cal4[1], cal4[2], cal4[3], lsf[0], inv};
        const FrameArrays PK = frame("pjf");
        M::MapPointFields mp = points("pj");
        mp.angle = rd<float>("pj_angle");
        M::Pose T; M::Sim3 Sc;
        const std::vector<float> Tv = rd<float>("pj_Tcw"), Sv = rd<float>("pj_Scw");
        for (int i = 0; i < 12; i++) { T.m[i] = Tv[i]; Sc.m[i] = Sv[i]; }
        ORB_SLAM2::ORBmatcher m(0.9f, true);
        std::vector<int32_t> bi, bd, k;
        const int n1 = m.Fuse(PK.view(false), cal, T, mp, 3.0f, bi, bd);
        printf("pfuse_n %d\n", n1); out("pfuse_idx", bi); out("pfuse_dist", bd);
        const int n2 = m.Fuse(PK.view(false), cal, Sc, mp, 4.0f, bi, bd);
        printf("pfuse_scw_n %d\n", n2); out("pfuse_scw_idx", bi); out("pfuse_scw_dist", bd);
        const int n3 = m.SearchByProjection(PK.view(true), cal, Sc, mp, 10, k);
        printf("pgreedy_kf_n %d\n", n3); out("pgreedy_kf", k);
        const int n4 = m.SearchByProjection(PK.view(true), cal, T, mp, 10.0f, 100, k);
        printf("pgreedy_f_n %d\n", n4); out("pgreedy_f", k);
        const FrameArrays S1 = frame("s1f"), S2 = frame("s2f");
        const M::MapPointFields mp1 = points("s1"), mp2 = points("s2");
        const std::vector<float> T1 = rd<float>("s_T1w"), T2 = rd<float>("s_T2w"), R12 = rd<float>("s_R12"), t12 = rd<float>("s_t12"),
                                 s12 = rd<float>("s_s12");
        M::Pose P1, P2;
        for (int i = 0; i < 12; i++) { P1.m[i] = T1[i]; P2.m[i] = T2[i]; }
        std::vector<int32_t> m12;
        const int n5 = m.SearchBySim3(S1.view(false), cal, P1, S2.view(false), cal, P2, mp1, mp2, s12[0], R12.data(), t12.data(),
                                      7.5f, m12);
        printf("psim3_n %d\n", n5); out("psim3", m12);
    }
    {   // LocalMapping's per-point loops: CreateNewMapPoints' triangulation, MapPoint::UpdateNormalAndDepth
        ORB_SLAM2::LocalMappingOps ops;
        auto kf = [](const std::string& p) {
            ORB_SLAM2::TriangulationKeyFrame k;
            const std::vector<float> T = rd<float>((p + "_Tcw").c_str()), K4 = rd<float>("tr_K");
            for (int i = 0; i < 12; i++) k.Tcw[i] = T[i];
            k.fx = K4[0]; k.fy = K4[1]; k.cx = K4[2]; k.cy = K4[3];
            k.mvScaleFactors = rd<float>("scale_factors"); k.mvLevelSigma2 = rd<float>("level_sigma2");
            return k;
        };
        ORB_SLAM2::TriangulationMatches mt;
        mt.xy1 = rd<float>("tr_xy1"); mt.xy2 = rd<float>("tr_xy2"); mt.octave1 = rd<int32_t>("tr_o1"); mt.octave2 = rd<int32_t>("tr_o2");
        mt.neighbour.assign(mt.octave1.size(), 0);
        std::vector<uint8_t> ok;
        std::vector<float> x3D;
        const int nnew = ops.TriangulateMatches(kf("tr1"), {kf("tr2")}, rd<float>("tr_ratio")[0], mt, ok, x3D);
        for (size_t k = 0; k < ok.size(); k++)
            if (!ok[k]) x3D[3 * k] = x3D[3 * k + 1] = x3D[3 * k + 2] = 0.f;
        printf("tri_n %d\n", nnew); out("tri_ok", ok); out("tri_x3d", x3D);
        {   // the same with the new points' normal / distance range in one launch
            std::vector<uint8_t> ok2;
            std::vector<float> x2, nn, mxn, mnn;
            const int nnew2 = ops.CreateNewPoints(kf("tr1"), {kf("tr2")}, rd<float>("tr_ratio")[0], mt, ok2, x2, nn, mxn, mnn);
            for (size_t k = 0; k < ok2.size(); k++)
                if (!ok2[k]) x2[3 * k] = x2[3 * k + 1] = x2[3 * k + 2] = 0.f;
            printf("new_n %d\n", nnew2); out("new_ok", ok2); out("new_x3d", x2); out("new_normal", nn); out("new_max", mxn); out("new_min", mnn);
        }
        std::vector<float> nrm = rd<float>("nd_normal"), mx = rd<float>("nd_max"), mn = rd<float>("nd_min");
        ops.UpdateNormalAndDepth(rd<int32_t>("nd_off"), rd<float>("nd_obs"), rd<float>("nd_Xw"), rd<float>("nd_ref"), rd<float>("nd_ls"),
                                 rd<float>("nd_ll"), nrm, mx, mn);
        out("nd_normal_out", nrm); out("nd_max_out", mx); out("nd_min_out", mn);
    }
    // ---- Optimizer ----------------------------------------------------------------------------------------------
    ORB_SLAM2::Optimizer& optimizer = ORB_SLAM2::Optimizer::ThreadInstance();  // static call sites, as in the reference
    {
        ORB_SLAM2::BAWindow win;
        win.Tcw = rd<float>("ba_Tcw"); win.fixed = rd<uint8_t>("ba_fixed"); win.intr = rd<float>("ba_intr");
        win.Xw = rd<float>("ba_Xw"); win.edge_kf = rd<int32_t>("ba_edge_pose"); win.edge_mp = rd<int32_t>("ba_edge_point");
        win.obs = rd<float>("ba_obs"); win.inv_sigma2 = rd<float>("ba_inv_sigma2");
        ORB_SLAM2::BAResult res;
        bool stop = false;
        optimizer.LocalBundleAdjustment(win, &stop, res);
        printf("ba %.9e %.9e %d %d %d\n", res.info.chi2_initial, res.info.chi2_final, res.info.iterations_stage1,
               res.info.iterations_stage2, res.info.n_outliers);
        FILE* f = fopen((g_dir + "/ba_out.bin").c_str(), "wb");
        fwrite(res.Tcw.data(), 4, res.Tcw.size(), f); fwrite(res.Xw.data(), 4, res.Xw.size(), f);
        fclose(f);
        out("ba_outlier", res.edge_outlier);
    }
    {
        std::vector<float> Tcw = rd<float>("po_Tcw");
        const std::vector<float> X = rd<float>("po_Xw"), ob = rd<float>("po_obs"), iw = rd<float>("po_inv_sigma2");
        std::vector<uint8_t> outl;
        const int inl = optimizer.PoseOptimization(Tcw.data(), K, X, ob, iw, outl);
        printf("pose %d %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g\n", inl, Tcw[0], Tcw[1], Tcw[2], Tcw[3],
               Tcw[4], Tcw[5], Tcw[6], Tcw[7], Tcw[8], Tcw[9], Tcw[10], Tcw[11]);
        out("pose_outlier", outl);
    }
    return 0;
}
