"""GPU: HIP ORB extractor vs the CPU oracle through the C ABI — bit-exact keypoints and
descriptors (integer pipeline + explicitly ordered float), with stage-by-stage checks so a
mismatch points at the kernel that caused it."""
import numpy as np
import pytest

from helpers import SEED

pytestmark = pytest.mark.gpu


HARRIS = {"on": False}  # set by the `harris` fixture: stage_check then ranks the oracle's distribution like "orb.response" = 1


@pytest.fixture
def harris(orc):
    """snk_set_definition("orb.response", 1) on both sides (the Harris response of the FAST corners ranks the points of a quadtree
    node and is the keypoints' response), restored afterwards."""
    from snake_slam_amd import _lib

    _lib.set_definition("orb.response", 1)
    orc.set_definition("orb.response", 1)
    HARRIS["on"] = True
    try:
        yield
    finally:
        HARRIS["on"] = False
        _lib.set_definition("orb.response", 0)
        orc.set_definition("orb.response", 0)


def stage_check(ext, orc, img, p):
    """Compare pyramid, per-cell candidates and per-level selections with the oracle."""
    from snake_slam_amd import orb as O

    levels, L = orc.pyramid(p, img)
    for l in range(L.n_levels):
        info = ext.debug_fetch(O.DEBUG_LEVEL_INFO, 0, l, np.int32)
        w, h, pitch, ncols, nrows, wcell, hcell, nfeat = [int(v) for v in info]
        assert (w, h, nfeat) == (L.w[l], L.h[l], L.nfeat[l]), f"level {l} geometry"
        if l >= 1:
            pyr = ext.debug_fetch(O.DEBUG_PYRAMID, 0, l, np.uint8).reshape(h, pitch)[:, :w]
            assert np.array_equal(pyr, levels[l]), f"pyramid level {l} differs"
        cand = orc.candidates(levels[l], p.ini_th, p.min_th, 8192)       # after the level budget
        cand_cells = orc.candidates(levels[l], p.ini_th, p.min_th, 60000)  # only the 64-per-cell rule
        want = {}
        for c in cand_cells:
            want.setdefault(int(c["cell"]), set()).add((int(c["x"]), int(c["y"]), int(c["score"])))
        cnt = ext.debug_fetch(O.DEBUG_CELL_COUNTS, 0, l, np.uint16)
        cc = ext.debug_fetch(O.DEBUG_CELL_CANDIDATES, 0, l, np.uint32).reshape(-1, 64)
        for cell in range(ncols * nrows):
            ci, cj = divmod(cell, ncols)
            x0, y0 = 19 + cj * wcell, 19 + ci * hcell
            got = set()
            for k in cc[cell, : min(int(cnt[cell]), 64)]:
                k = int(k)
                got.add((x0 + 63 - (k & 63), y0 + 63 - ((k >> 6) & 63), k >> 12))
            assert got == want.get(cell, set()), f"level {l} cell {cell}: candidates differ"
        if HARRIS["on"]:
            rank = np.array([orc.harris_rank(orc.harris_response(levels[l], int(c["x"]), int(c["y"]))) for c in cand], np.uint32)
            sel = orc.distribute_ranked(cand, rank, w, h, L.nfeat[l])
        else:
            sel = orc.distribute(cand, w, h, L.nfeat[l])
        n_sel = int(ext.debug_fetch(O.DEBUG_SELECTED_COUNT, 0, l, np.int32)[0])
        got_sel = ext.debug_fetch(O.DEBUG_SELECTED, 0, l, np.uint32)[:n_sel]
        want_sel = [(int(cand[i]["x"]) | (int(cand[i]["y"]) << 16)) for i in sel]
        assert [int(v) for v in got_sel] == want_sel, f"level {l}: selection differs"


def check_image(orc, img, nfeatures=1000, n_levels=4, scale=1.2, ini=20, mn=7, stages=True):
    from snake_slam_amd.orb import ORBExtractor

    ext = ORBExtractor(nfeatures, scale, n_levels, ini, mn)
    try:
        kps, desc = ext.Detect(img)
        p = orc.orb_params(nfeatures, scale, n_levels, ini, mn)
        if stages:
            stage_check(ext, orc, np.ascontiguousarray(img), p)
        wk, wd = orc.orb_detect(p, img)
        assert len(kps) == len(wk)
        for f in ("octave", "x", "y", "size", "response", "angle"):
            assert np.array_equal(kps[f], wk[f]), f"keypoint field {f} differs"
        assert np.array_equal(desc, wd), "descriptors differ"
        return len(kps)
    finally:
        ext.close()


def test_euroc_frame_parity(orc):
    from snake_slam_amd import synth

    left, right = synth.stereo_frame(0)
    assert check_image(orc, left) >= 1000
    assert check_image(orc, right) >= 1000


def test_kitti_shape_parity(orc):
    from snake_slam_amd import synth

    left, _ = synth.stereo_frame(1, 1241, 376, n_rects=600)
    assert check_image(orc, left, 2000, 7) >= 1500


@pytest.mark.parametrize("shape", [(480, 752), (376, 1241), (200, 120), (97, 131), (70, 75), (40, 45)])
def test_noise_images_hit_the_candidate_budget(orc, shape):
    """Uniform noise: hundreds of corners per cell -> exercises the 64-per-cell cap, the level
    budget reduction and heavy score ties."""
    rng = np.random.default_rng(SEED + shape[0])
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    check_image(orc, img, 1000, 4)


def test_low_contrast_uses_min_threshold(orc):
    rng = np.random.default_rng(9)
    img = (100 + rng.integers(0, 3, (300, 400))).astype(np.uint8)
    for _ in range(60):
        y, x = int(rng.integers(20, 260)), int(rng.integers(20, 360))
        img[y:y + int(rng.integers(6, 30)), x:x + int(rng.integers(6, 30))] += np.uint8(rng.integers(9, 19))
    n = check_image(orc, img, 500, 3)
    assert n > 20


def test_flat_and_tiny_images(orc):
    assert check_image(orc, np.full((480, 752), 128, np.uint8)) == 0
    assert check_image(orc, np.zeros((40, 45), np.uint8), 100, 4) == 0


def test_pitch_and_params_variants(orc):
    from snake_slam_amd import synth

    left, _ = synth.stereo_frame(2, 640, 400, n_rects=300)
    padded = np.zeros((400, 700), np.uint8)
    padded[:, :640] = left
    check_image(orc, padded[:, :640], 1400, 3, 1.2, 20, 3)      # reference configs/saiga.ini
    check_image(orc, left, 300, 8, 1.3, 30, 10, stages=False)
    check_image(orc, left, 50, 1, 1.2, 20, 7)


@pytest.mark.parametrize("scale,levels", [(1.05, 6), (1.5, 4), (2.0, 4), (2.5, 3), (3.0, 3)])
def test_scale_factor_range(orc, scale, levels):
    """The streaming level pass makes the next pyramid level in-stream for scale <= 2 and falls
    back to the stand-alone resize above; both must reproduce the oracle's pyramid bit for bit."""
    from snake_slam_amd import synth

    left, _ = synth.stereo_frame(5, 752, 480, n_rects=300)
    check_image(orc, left, 800, levels, scale, 20, 7)


@pytest.mark.parametrize("shape", [(120, 1920), (333, 1023), (65, 249), (129, 245)])
def test_strip_and_band_boundaries(orc, shape):
    """Widths / heights around the 244-column strip and 64-row band sizes of the level pass."""
    rng = np.random.default_rng(SEED + shape[1])
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    img[::7, :] //= 2
    check_image(orc, img, 500, 4, 1.2, 20, 7)


def test_narrow_and_deep_pyramids(orc):
    """Levels much narrower than one wavefront strip (w < 126: the halo lanes far right of the image must
    not index outside the row) and levels below 8 x 8 (down-scaled by the stand-alone kernel, never blurred)."""
    rng = np.random.default_rng(SEED + 77)
    for shape, levels in [((200, 120), 4), ((60, 100), 16), ((300, 50), 8), ((9, 400), 3)]:
        img = rng.integers(0, 256, shape, dtype=np.uint8)
        check_image(orc, img, 300, levels, 1.2, 20, 7)


def test_random_shapes_and_parameters(orc):
    """Small fuzz over image shape / levels / scale / thresholds (seeded)."""
    from snake_slam_amd import synth

    rng = np.random.default_rng(SEED + 99)
    for k in range(10):
        w, h = int(rng.integers(40, 420)), int(rng.integers(40, 320))
        levels, scale = int(rng.integers(1, 9)), float(rng.choice([1.1, 1.2, 1.3, 1.5, 2.0]))
        if k % 2:
            img = rng.integers(0, 256, (h, w), dtype=np.uint8)
        else:
            img, _ = synth.stereo_frame(200 + k, w, h, n_rects=60)
        check_image(orc, img, int(rng.integers(50, 1500)), levels, scale, int(rng.integers(10, 40)), int(rng.integers(3, 10)),
                    stages=(k < 4))


def test_large_image(orc):
    """A 2000 x 1500 image with 5000 features and 8 levels: 9 strips x 24 bands at level 0, ~3300 FAST cells,
    levels above the 2048-candidate LDS carve of the distribution kernel."""
    from snake_slam_amd import synth

    img, _ = synth.stereo_frame(77, 2000, 1500, n_rects=2500)
    n = check_image(orc, img, 5000, 8, 1.2, 20, 7, stages=False)
    assert n >= 5000


def test_batch_dev_matches_single(orc):
    import torch
    from snake_slam_amd import synth
    from snake_slam_amd.orb import ORBExtractor, KEYPOINT_DTYPE

    B = 5
    imgs = [synth.stereo_frame(i)[i % 2] for i in range(B)]
    imgs[3] = np.full((480, 752), 7, np.uint8)  # an empty frame inside the batch
    pitch = 768
    host = np.zeros((B, 480, pitch), np.uint8)
    for i, im in enumerate(imgs):
        host[i, :, :752] = im
    ext = ORBExtractor(1000, 1.2, 4, 20, 7)
    cap = ext.configure(752, 480, B)
    dev = torch.device("cuda:0")
    d_img = torch.from_numpy(host).to(dev)
    d_kps = torch.zeros((B, cap, 24), dtype=torch.uint8, device=dev)
    d_desc = torch.zeros((B, cap, 4), dtype=torch.int64, device=dev)
    d_n = torch.zeros(B, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ext.detect_batch_dev(d_img, d_kps, d_desc, d_n)
    ext.sync()
    n = d_n.cpu().numpy()
    kps = d_kps.cpu().numpy().view(KEYPOINT_DTYPE).reshape(B, cap)
    desc = d_desc.cpu().numpy().view(np.uint64)
    p = orc.orb_params()
    for i in range(B):
        wk, wd = orc.orb_detect(p, imgs[i])
        assert n[i] == len(wk), f"image {i}"
        assert np.array_equal(kps[i, : n[i]], wk.astype(KEYPOINT_DTYPE)) and np.array_equal(desc[i, : n[i]], wd)
    ext.close()


def test_split_batch_two_streams(orc):
    """With set_chains(2) batches of >= 8 images run as two half-batch launch chains on two streams: every image must come
    out as if it had been extracted alone, the stage timers count one entry per chain, and work queued on
    the caller's stream afterwards sees both halves."""
    import torch
    from snake_slam_amd import synth
    from snake_slam_amd.orb import ORBExtractor, KEYPOINT_DTYPE

    B, W, H = 11, 320, 240
    imgs = [synth.stereo_frame(300 + i, W, H, n_rects=80)[0] for i in range(B)]
    host = np.stack(imgs)
    ext = ORBExtractor(400, 1.2, 4, 20, 7)
    cap = ext.configure(W, H, B)
    dev = torch.device("cuda:0")
    d_img = torch.from_numpy(host).to(dev)
    d_kps = torch.zeros((B, cap, 24), dtype=torch.uint8, device=dev)
    d_desc = torch.zeros((B, cap, 4), dtype=torch.int64, device=dev)
    d_n = torch.zeros(B, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ext.set_chains(2)
    ext.set_profiling(True)
    ext.detect_batch_dev(d_img, d_kps, d_desc, d_n)
    ms, chains = ext.stage_times()
    assert chains == 2 and all(m >= 0 for m in ms)
    ext.set_profiling(False)
    ext.sync()
    n = d_n.cpu().numpy()
    kps = d_kps.cpu().numpy().view(KEYPOINT_DTYPE).reshape(B, cap)
    desc = d_desc.cpu().numpy().view(np.uint64)
    p = orc.orb_params(400, 1.2, 4, 20, 7)
    for i in range(B):
        wk, wd = orc.orb_detect(p, imgs[i])
        assert n[i] == len(wk) and n[i] > 100, f"image {i}"
        assert np.array_equal(kps[i, : n[i]], wk.astype(KEYPOINT_DTYPE)) and np.array_equal(desc[i, : n[i]], wd)
    ext.close()


def test_staggered_schedule(orc):
    """snk_orb_set_stagger(3): the batch in three ranges, front halves (level passes, FAST) back to back on the handle's stream, the
    back half (distribution, descriptors) of range p on a second stream beside the front half of range p + 1 -- every image as if
    extracted alone, the stage timers count one entry per range, work queued on the caller's stream afterwards sees everything."""
    import torch
    from snake_slam_amd import synth
    from snake_slam_amd.orb import ORBExtractor, KEYPOINT_DTYPE

    B, W, H = 13, 320, 240
    imgs = [synth.stereo_frame(500 + i, W, H, n_rects=80)[0] for i in range(B)]
    ext = ORBExtractor(400, 1.2, 4, 20, 7)
    cap = ext.configure(W, H, B)
    dev = torch.device("cuda:0")
    d_img = torch.from_numpy(np.stack(imgs)).to(dev)
    d_kps = torch.zeros((B, cap, 24), dtype=torch.uint8, device=dev)
    d_desc = torch.zeros((B, cap, 4), dtype=torch.int64, device=dev)
    d_n = torch.zeros(B, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ext.set_stagger(3)
    ext.set_profiling(True)
    for _ in range(2):  # twice: the second call reuses the events of the first
        ext.detect_batch_dev(d_img, d_kps, d_desc, d_n)
    ms, parts = ext.stage_times()
    assert parts == 6 and all(m >= 0 for m in ms)
    ext.set_profiling(False)
    ext.sync()
    n = d_n.cpu().numpy()
    kps = d_kps.cpu().numpy().view(KEYPOINT_DTYPE).reshape(B, cap)
    desc = d_desc.cpu().numpy().view(np.uint64)
    p = orc.orb_params(400, 1.2, 4, 20, 7)
    for i in range(B):
        wk, wd = orc.orb_detect(p, imgs[i])
        assert n[i] == len(wk) and n[i] > 100, f"image {i}"
        assert np.array_equal(kps[i, : n[i]], wk.astype(KEYPOINT_DTYPE)) and np.array_equal(desc[i, : n[i]], wd)
    ext.set_stagger(0)
    ext.close()


def test_xcd_mapped_batch(orc):
    """Batches of >= 16 images use the XCD-aware 1-D grids (image b on XCD b % 8); 17 is not a multiple of 8,
    so the padded part of the grid must fall out cleanly."""
    import torch
    from snake_slam_amd import synth
    from snake_slam_amd.orb import ORBExtractor, KEYPOINT_DTYPE

    B, W, H = 17, 200, 160
    imgs = [synth.stereo_frame(400 + i, W, H, n_rects=50)[0] for i in range(B)]
    ext = ORBExtractor(300, 1.2, 3, 20, 7)
    cap = ext.configure(W, H, B)
    dev = torch.device("cuda:0")
    d_img = torch.from_numpy(np.stack(imgs)).to(dev)
    d_kps = torch.zeros((B, cap, 24), dtype=torch.uint8, device=dev)
    d_desc = torch.zeros((B, cap, 4), dtype=torch.int64, device=dev)
    d_n = torch.zeros(B, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    for chains in (1, 2):
        ext.set_chains(chains)
        d_n.zero_()
        ext.detect_batch_dev(d_img, d_kps, d_desc, d_n)
        ext.sync()
        n = d_n.cpu().numpy()
        kps = d_kps.cpu().numpy().view(KEYPOINT_DTYPE).reshape(B, cap)
        desc = d_desc.cpu().numpy().view(np.uint64)
        p = orc.orb_params(300, 1.2, 3, 20, 7)
        for i in range(B):
            wk, wd = orc.orb_detect(p, imgs[i])
            assert n[i] == len(wk) and n[i] > 50, f"image {i} chains {chains}"
            assert np.array_equal(kps[i, : n[i]], wk.astype(KEYPOINT_DTYPE)) and np.array_equal(desc[i, : n[i]], wd)
    ext.close()


def test_unaligned_device_images_take_the_byte_path(orc):
    """Odd base address and odd pitch: the aligned dword loaders must fall back to byte loads."""
    import torch
    from snake_slam_amd import synth
    from snake_slam_amd.orb import ORBExtractor, KEYPOINT_DTYPE

    B, H, W, P = 2, 240, 320, 323
    imgs = [synth.stereo_frame(10 + i, W, H, n_rects=120)[0] for i in range(B)]
    host = np.zeros((B, H, P), np.uint8)
    for i, im in enumerate(imgs):
        host[i, :, :W] = im
    dev = torch.device("cuda:0")
    flat = torch.zeros(B * H * P + 16, dtype=torch.uint8, device=dev)
    view = flat[1:1 + B * H * P].view(B, H, P)
    view.copy_(torch.from_numpy(host))
    assert view.data_ptr() % 4 != 0
    ext = ORBExtractor(500, 1.2, 4, 20, 7)
    cap = ext.configure(W, H, B)
    d_kps = torch.zeros((B, cap, 24), dtype=torch.uint8, device=dev)
    d_desc = torch.zeros((B, cap, 4), dtype=torch.int64, device=dev)
    d_n = torch.zeros(B, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ext.detect_batch_dev(view, d_kps, d_desc, d_n)
    ext.sync()
    n = d_n.cpu().numpy()
    kps = d_kps.cpu().numpy().view(KEYPOINT_DTYPE).reshape(B, cap)
    desc = d_desc.cpu().numpy().view(np.uint64)
    p = orc.orb_params(500, 1.2, 4, 20, 7)
    for i in range(B):
        wk, wd = orc.orb_detect(p, imgs[i])
        assert n[i] == len(wk) and len(wk) > 100
        assert np.array_equal(kps[i, : n[i]], wk.astype(KEYPOINT_DTYPE)) and np.array_equal(desc[i, : n[i]], wd)
    ext.close()


@pytest.mark.parametrize("workload", ["euroc", "kitti"])
def test_candidate_budget_is_inactive_on_every_bench_frame(workload):
    """the per-cell top-64 / level_cap (8192) candidate budget is this repository's definition, not
    ORB-SLAM2's; the claim that it has no effect on ordinary images is checked here on every synthetic frame bench.py
    uses (one distinct stereo pair per frame of the batch; the first 64 of each workload are checked here): no FAST cell holds more than 64 candidates and no level more than 8192,
    so the budget never truncates and the result is the un-budgeted algorithm's."""
    from snake_slam_amd import orb as O
    from snake_slam_amd import synth

    if workload == "euroc":
        w, h, prm = 752, 480, (1000, 1.2, 4, 20, 7)
    else:
        w, h, prm = 1241, 376, (2000, 1.2, 7, 20, 7)
    ext = O.ORBExtractor(*prm)
    worst_cell, worst_level = 0, 0
    for pair in synth.stereo_frames(range(64), w, h):
        for img in pair:
            ext.Detect(img)
            for l in range(prm[2]):
                cnt = ext.debug_fetch(O.DEBUG_CELL_COUNTS, 0, l, np.uint16)
                worst_cell = max(worst_cell, int(cnt.max()))
                worst_level = max(worst_level, int(cnt.astype(np.int64).sum()))
    ext.close()
    assert 0 < worst_cell <= 64, worst_cell
    assert worst_level <= 8192, worst_level


@pytest.mark.skipif(__import__("os").environ.get("SNK_ORB_NO_RECURSE") == "1", reason="child run")
def test_fast_survivor_list_spill_path_on_ordinary_images():
    """fast_kernel keeps a bounded list of quick-test survivors per cell (half of the cell's pixels) and, when a cell
    has more, scores the list and starts it again; non-maximum suppression then walks the score map.  Noise images take that
    path by themselves; here the whole ORB parity file runs again in a child process with the list forced down to 128 entries
    (SNK_ORB_FAST_SURV_CAP), so that ordinary images -- every stage compared with the oracle -- take it as well."""
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, "-m", "pytest", str(root / "tests" / "test_orb_gpu.py"), "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider",
                        "-k", "not budget_is_inactive"],
                       env=dict(os.environ, SNK_ORB_FAST_SURV_CAP="128", SNK_ORB_NO_RECURSE="1"), capture_output=True, text=True, cwd=str(root), timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-1000:])
    assert " passed" in r.stdout


@pytest.mark.skipif(__import__("os").environ.get("SNK_ORB_NO_RECURSE") == "1", reason="child run")
def test_fast_kernel_loop_form_on_ordinary_images():
    """Big launches run fast_kernel with several cells per wavefront (the loop form; measurements in profiles/NOTES.md); the parity file runs again
    in a child process with THREE cells per wavefront forced for every launch size (SNK_ORB_FAST_CPW; 3 does not divide the cell
    counts, so the last wavefronts of an image run out of cells mid-loop)."""
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, "-m", "pytest", str(root / "tests" / "test_orb_gpu.py"), "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider",
                        "-k", "not budget_is_inactive"],
                       env=dict(os.environ, SNK_ORB_FAST_CPW="3", SNK_ORB_NO_RECURSE="1"), capture_output=True, text=True, cwd=str(root), timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-1000:])
    assert " passed" in r.stdout


def test_levels_scaled_down_to_nothing(orc):
    """Small images with many levels and a large scale factor: deep levels round to zero width and / or height (found by
    tools/fuzz_orb.py: a zero grid dimension in one case, an integer division by the zero strip count in another).  Such levels
    have no pixels and no cells; the levels above them are unaffected."""
    rng = np.random.default_rng(SEED + 1234)
    for shape, levels, scale in [((40, 1300), 8, 2.5), ((60, 45), 8, 2.5), ((45, 70), 8, 2.0), ((800, 48), 7, 2.5)]:
        img = rng.integers(0, 256, shape, dtype=np.uint8)
        check_image(orc, img, 400, levels, scale, 20, 7, stages=False)


def test_harris_response_parity(orc, harris):
    """north_star's "Harris score" as the definition switch "orb.response" = 1: BASELINE configs 2 and 3 at full size, bit for bit
    (keypoints, the float response, descriptors), stage by stage; and the switch changes the result (it is not a no-op)."""
    from snake_slam_amd import _lib, synth
    from snake_slam_amd.orb import ORBExtractor

    left, right = synth.stereo_frame(0)
    assert check_image(orc, left) >= 1000
    assert check_image(orc, right) >= 1000
    kitti, _ = synth.stereo_frame(1, 1241, 376, n_rects=600)
    assert check_image(orc, kitti, 2000, 7) >= 1500
    ext = ORBExtractor(1000, 1.2, 4, 20, 7)
    try:
        k1, _ = ext.Detect(left)
        _lib.set_definition("orb.response", 0)
        k0, _ = ext.Detect(left)
        _lib.set_definition("orb.response", 1)
    finally:
        ext.close()
    assert len(k0) == len(k1)
    same = set(zip(k0["x"], k0["y"], k0["octave"])) & set(zip(k1["x"], k1["y"], k1["octave"]))
    assert 0.3 * len(k0) < len(same) < len(k0), "Harris ranking must move some keypoints, not all"
    assert np.all(k0["response"] == np.round(k0["response"])) and not np.all(k1["response"] == np.round(k1["response"]))


@pytest.mark.parametrize("shape", [(480, 752), (200, 120), (97, 131), (40, 45)])
def test_harris_on_noise_images(orc, harris, shape):
    """Uniform noise under "orb.response" = 1: the 64-per-cell cap and the level budget (both on the FAST score) feed the Harris
    ranking; negative responses (edges) occur and must order correctly."""
    rng = np.random.default_rng(SEED + 5 * shape[0])
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    check_image(orc, img, 1000, 4)


def test_harris_random_shapes_unaligned_and_large(orc, harris):
    """Random shapes / parameters; a level-0 view whose rows are not 4-byte aligned (byte loads in harris_kernel); a 2000 x 1500
    image whose levels exceed the 2048-candidate LDS carve (ranks of the full-budget launch live in global scratch)."""
    from snake_slam_amd import synth

    rng = np.random.default_rng(SEED + 123)
    for k in range(8):
        w, h = int(rng.integers(40, 420)), int(rng.integers(40, 320))
        levels, scale = int(rng.integers(1, 9)), float(rng.choice([1.1, 1.2, 1.3, 1.5, 2.0]))
        img = rng.integers(0, 256, (h, w), dtype=np.uint8) if k % 2 else synth.stereo_frame(300 + k, w, h, n_rects=60)[0]
        check_image(orc, img, int(rng.integers(50, 1500)), levels, scale, int(rng.integers(10, 40)), int(rng.integers(3, 10)), stages=(k < 3))
    big, _ = synth.stereo_frame(77, 2000, 1500, n_rects=2500)
    assert check_image(orc, big, 5000, 8, 1.2, 20, 7, stages=False) >= 5000
    noise = rng.integers(0, 256, (700, 900), dtype=np.uint8)  # > 2048 candidates on level 0: distribute_large_kernel
    check_image(orc, noise, 3000, 3, 1.2, 20, 7, stages=False)


def test_harris_batch_dev_and_frontend(orc, harris):
    """The device-resident batch entry point (XCD-mapped, >= 16 images) and the one-call front-end (hipGraph keyed by the
    definition: flipping the switch between frames must rebuild it) under "orb.response" = 1."""
    import torch
    from snake_slam_amd import _lib, synth
    from snake_slam_amd.orb import ORBExtractor, KEYPOINT_DTYPE

    B, W, H = 18, 320, 240
    imgs = [synth.stereo_frame(i, W, H, n_rects=150)[i % 2] for i in range(B)]
    ext = ORBExtractor(300, 1.2, 3, 20, 7)
    try:
        cap = ext.configure(W, H, B)
        dev = torch.device("cuda:0")
        d_img = torch.from_numpy(np.stack(imgs)).to(dev)
        d_kps = torch.zeros((B, cap, 24), dtype=torch.uint8, device=dev)
        d_desc = torch.zeros((B, cap, 4), dtype=torch.int64, device=dev)
        d_n = torch.zeros(B, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        ext.detect_batch_dev(d_img, d_kps, d_desc, d_n)
        ext.sync()
        n = d_n.cpu().numpy()
        kps = d_kps.cpu().numpy().view(KEYPOINT_DTYPE).reshape(B, cap)
        desc = d_desc.cpu().numpy().view(np.uint64)
        p = orc.orb_params(300, 1.2, 3, 20, 7)
        for i in range(B):
            wk, wd = orc.orb_detect(p, imgs[i])
            assert n[i] == len(wk), f"image {i}"
            assert np.array_equal(kps[i, : n[i]], wk.astype(KEYPOINT_DTYPE)) and np.array_equal(desc[i, : n[i]], wd), f"image {i}"
    finally:
        ext.close()

    # the one-call front-end: frames 1-3 under Harris (uncaptured, captured, replayed), then the switch flips back and forth
    from snake_slam_amd.frontend import Frontend
    from snake_slam_amd.matcher import Rectification

    left, right = synth.stereo_frame(3, 752, 480)
    e_k, e_d = (458.654, 457.296, 367.215, 248.375), (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0, 0.0, 0.0, 0.0)
    rect = Rectification.make(e_k, e_d)
    fe = Frontend((1000, 1.2, 4, 20, 7), rect, rect, (-120.0, -60.0, 880.0, 540.0), 47.9)
    p = orc.orb_params(1000, 1.2, 4, 20, 7)
    try:
        for mode in (1, 1, 1, 0, 0, 1, 1):
            _lib.set_definition("orb.response", mode)
            orc.set_definition("orb.response", mode)
            fr = fe.Process(left, right)
            wk, _ = orc.orb_detect(p, left)
            wr, wdr = orc.orb_detect(p, right)
            assert fr["N"] == len(wk) and fr["n_right"] == len(wr), mode
            # the left keypoints come back in grid order: compare as multisets of (x, y, octave, response)
            k = fr["keypoints"]
            got = sorted(zip(k["x"], k["y"], k["octave"], k["response"]))
            want = sorted(zip(wk["x"], wk["y"], wk["octave"], wk["response"]))
            assert got == want, mode
            assert np.array_equal(fr["keypoints_right"], wr.astype(KEYPOINT_DTYPE)) and np.array_equal(fr["descriptors_right"], wdr), mode
    finally:
        fe.close()

