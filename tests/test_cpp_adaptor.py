"""CPU: the C++ adaptor header compiles against the C ABI and links with the library."""
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent

SRC = r'''
#include "snake_hip.hpp"
#include <map>
int main(int argc, char**) {
    if (argc > 100) {  // never executed here (no GPU); exercises every template / inline path at compile time
        snake_hip::ORBExtractor ext(1000, 1.2f, 4, 20, 7, 2);
        std::vector<snake_hip::KeyPointF> k; std::vector<snake_hip::DescriptorORB> d;
        ext.Detect(nullptr, 752, 480, 752, k, d);
        snake_hip::BruteForceMatcher m; m.matchKnn2_omp(d, d, 4); m.filterMatches(60, 0.8f);
        snake_hip::Preprocess pp; std::vector<snk_kp64> r; snk_rectification rect{}; pp.Rectify(rect, k, r);
        std::vector<float> rp, dp; pp.StereoMatching(r, d, r, d, 47.9, {1.f, 1.2f}, true, rp, dp);
        snake_hip::FrameView fv; snake_hip::SnakeORBMatcher om; snk_grid_bounds gb{0, 0, 752, 480}; om.CreateGrid(fv, gb);
        snk_camera cam{}; double pose[7] = {0, 0, 0, 1, 0, 0, 0}; std::vector<int32_t> match; std::vector<uint8_t> vis;
        std::vector<snk_lm_coarse> lc; om.SearchByProjectionFrameFrame2(fv, cam, pose, lc, 15.f, 75, 0, {1.f, 1.2f}, match);
        std::vector<snk_lm_fine> lf; om.SearchByProjection2(fv, cam, pose, lf, 5.f, 0.8f, {1.f, 1.2f}, match, vis);
        om.SearchByProjectionFrameToKeyframe(fv, cam, pose, {}, {}, {}, 15.f, 100, match);
        snake_hip::MappingORBMatcher mm; std::vector<snk_fusion_point> fp; std::vector<std::pair<int, int>> fc;
        mm.Fuse(fv, cam, pose, {}, fp, fc, 4.f, 2.f, 50, {1.f, 1.2f}); double E[9] = {}; double g[4] = {};
        mm.SearchForTriangulationProject(g, 2, 2, pose, pose, cam, {}, {}, {}, {}, fv, {}, E, fc, 4.f, 50);
        std::map<unsigned, std::vector<unsigned>> bowmap{{3u, {0u, 1u}}};
        auto bow = snake_hip::MappingORBMatcher::BowFeatureVector::from(bowmap);
        mm.SearchForTriangulation2(cam, E, {}, {}, {}, bow, {}, {}, {}, bow, fc, 4.f, 50);
        mm.SearchForTriangulationBF(cam, E, {}, {}, {}, {}, {}, {}, fc, 50);
        snake_hip::DeferredMapper dm; std::vector<snk_relink_query> rq; dm.RelinkSearch(fv, cam, pose, rq, match, match);
        snake_hip::PoseRefinement pr(1.0); std::vector<std::array<double, 3>> wps; std::vector<snk_pose_obs> po;
        pr.optimizePoseRobust(wps, po, vis, pose, cam); std::vector<snk_pose_problem> pb; pr.optimizeBatch(pb, cam);
        snake_hip::Scene sc; snake_hip::BARec ba; ba.create(sc); ba.initAndSolve(); ba.residualsSquared();
    }
    return snk_device_count() >= 0 ? 0 : 1;
}
'''


def test_adaptor_compiles_and_links(tmp_path):
    src = tmp_path / "adaptor_check.cpp"
    src.write_text(SRC)
    lib = ROOT / "snake_slam_amd" / "lib"
    assert (lib / "libsnake_hip.so").exists()
    cmd = ["g++", "-std=c++17", "-Wall", "-Werror", f"-I{ROOT / 'include'}", f"-I{ROOT / 'snake_slam_amd' / 'cpp'}", str(src),
           f"-L{lib}", "-lsnake_hip", "-L/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib",
           "-o", str(tmp_path / "adaptor_check")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # the binary runs without a GPU: snk_device_count() reports 0 devices instead of failing
    r = subprocess.run([str(tmp_path / "adaptor_check")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_reference_signature_shims_compile_against_mock_snake_types(tmp_path):
    """snake_hip_reference.hpp -- SearchByProjectionFrameFrame2(Frame&, const LocalMap<CoarseTrackingPoint>&, ...) and the other
    reference signatures -- instantiated with mock structs that carry exactly the member names of Snake/Map/{Frame,Features,
    LocalMap}.h and of the Saiga::Scene MakeLocalScene fills (tests/cpp/reference_shims_driver.cpp).  Compiled and linked here;
    executed on the golden inputs by tests/test_cpp_reference_shims_gpu.py."""
    lib = ROOT / "snake_slam_amd" / "lib"
    cmd = ["g++", "-std=c++17", "-Wall", "-Werror", f"-I{ROOT / 'include'}", f"-I{ROOT / 'snake_slam_amd' / 'cpp'}",
           str(ROOT / "tests" / "cpp" / "reference_shims_driver.cpp"), f"-L{lib}", "-lsnake_hip", "-L/opt/rocm/lib", "-lamdhip64",
           f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(tmp_path / "ref_driver")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # without inputs / a GPU it must fail cleanly (exception text, status 1), not crash
    r = subprocess.run([str(tmp_path / "ref_driver"), str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 1 and "reference_shims_driver:" in r.stderr
