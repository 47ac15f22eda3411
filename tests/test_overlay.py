"""Mesh overlay (include/RenderSMPLMesh.h:16-110).

CPU tests pin the C restatement's own properties (there are no reference golden images: parity unpinned); the
-m gpu tests compare the device path with the restatement through the C ABI, pixel for pixel (integer work: exact).
"""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import random_params

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "overlay_v1.json")


@pytest.fixture(scope="module")
def ovo():
    from oracle import overlay
    overlay.lib()
    return overlay


@pytest.fixture(scope="module")
def faces(synth, model):
    return synth.make_faces(model)


def posed_clouds(synth, model, F, seed, depth=3.0, shift=(0.0, 0.0)):
    rng = np.random.default_rng(seed)
    x = random_params(rng, F)
    x[:, 0] = 1.0
    x[:, 4:7] = np.array([shift[0], shift[1], depth]) + rng.normal(scale=0.05, size=(F, 3))
    R0 = -np.eye(3)
    return np.stack([synth.forward_numpy(model, x[f], rng.normal(size=10) * 0.5, R0)[1] for f in range(F)])


def oracle_images(ovo, clouds, faces, bg, intr, fill=True, cull=True, wire=False):
    out = bg.copy()
    for f in range(clouds.shape[0]):
        ovo.render(clouds[f], faces, out[f], *intr, fill=fill, backface_cull=cull, wireframe=wire)
    return out


# ---------------------------------------------------------------------------------------------------- CPU
def test_tables_shape(ovo):
    filt, slope = ovo.tables()
    assert np.array_equal(filt[:32], filt[:32][::-1])          # the centre tap is symmetric
    assert np.all(np.diff(filt[32:].astype(int)) <= 0)         # the side taps fall off
    assert np.all(np.diff(slope.astype(int)) >= 0) and slope[0] == 181   # 256 / sqrt(2) .. 256


def test_triangle_interior_and_translation(ovo):
    img = np.zeros((64, 80, 3), np.uint8)
    ovo.fill_triangle(img, [10, 8, 60, 20, 25, 50], 200)
    assert (img[..., 0] == 200).sum() > 800                    # area 1120 px less the soft rim
    assert img[25, 30, 0] == 200 and img[5, 5, 0] == 0
    assert np.array_equal(img[..., 0], img[..., 1]) and np.array_equal(img[..., 0], img[..., 2])
    img2 = np.zeros((64, 80, 3), np.uint8)
    ovo.fill_triangle(img2, [10 + 7, 8 + 3, 60 + 7, 20 + 3, 25 + 7, 50 + 3], 200)
    assert np.array_equal(img2[3:, 7:], img[:-3, :-7])         # integer scan conversion: shifts with its corners


def test_triangle_degenerate_and_offscreen(ovo):
    img = np.full((32, 32, 3), 9, np.uint8)
    ovo.fill_triangle(img, [-50, -50, -10, -40, -30, -5], 255)        # entirely outside
    ovo.fill_triangle(img, [100, 5, 140, 9, 120, 60], 255)
    assert np.all(img == 9)
    ovo.fill_triangle(img, [4, 4, 4, 4, 4, 4], 255)                   # a point: only the AA dot, no span
    assert img[4, 4, 0] > 9 and (img[..., 0] == 255).sum() <= 1
    big = np.zeros((32, 32, 3), np.uint8)
    ovo.fill_triangle(big, [-500, -400, 900, -300, 20, 1200], 77)     # covers the whole image
    assert np.all(big == 77)


def test_painter_order_and_cull(ovo):
    # two triangles over the same pixels: the nearer one (smaller mean z) is drawn last and wins its interior
    cloud = np.array([[-1, -1, 4.0], [1, -1, 4.0], [0, 1, 4.0], [-1, -1, 2.0], [0, 1, 2.0], [1, -1, 2.0]])
    faces = np.array([[0, 2, 1], [3, 4, 5]], np.int32)               # both wound to face the camera (n.z < 0)
    intr = (100.0, 100.0, 64.0, 64.0)
    face, depth, pts, gray = ovo.drawlist(cloud, faces, *intr)
    assert list(face) == [0, 1] and depth[0] > depth[1]
    img = np.zeros((128, 128, 3), np.uint8)
    ovo.render(cloud, faces, img, *intr)
    assert img[64, 64, 0] == gray[1]
    # reversed winding is culled; with culling off it is drawn with shade clamped to 0
    face_c, *_ = ovo.drawlist(cloud, faces[:, ::-1].copy(), *intr)
    assert len(face_c) == 0
    face_n, _, _, gray_n = ovo.drawlist(cloud, faces[:, ::-1].copy(), *intr, backface_cull=False)
    assert len(face_n) == 2 and np.all(gray_n == 0)
    # a vertex at or behind the camera drops the face (RenderSMPLMesh.h:42,52)
    cloud2 = cloud.copy(); cloud2[0, 2] = 1e-7
    assert list(ovo.drawlist(cloud2, faces, *intr)[0]) == [1]


def test_wireframe_outline(ovo):
    cloud = np.array([[-1, -1, 4.0], [1, -1, 4.0], [0, 1, 4.0]])
    faces = np.array([[0, 2, 1]], np.int32)
    intr = (40.0, 40.0, 12.0, 12.0)
    img = np.zeros((24, 24, 3), np.uint8)
    ovo.render(cloud, faces, img, *intr, fill=False, wireframe=True)
    assert 0 < img.max() <= 40 and img[8, 12, 0] == 0          # outline only, never brighter than its colour
    both = np.zeros((24, 24, 3), np.uint8)
    ovo.render(cloud, faces, both, *intr, fill=True, wireframe=True)
    filled = np.zeros((24, 24, 3), np.uint8)
    ovo.render(cloud, faces, filled, *intr)
    assert both[8, 12, 0] == filled[8, 12, 0] and (both != filled).any()   # interior untouched, rim darkened


def test_drawlist_order_is_stable(ovo):
    rng = np.random.default_rng(3)
    cloud = rng.uniform(-1, 1, (60, 3)); cloud[:, 2] = np.round(rng.uniform(2, 3, 60), 1)   # many equal depths
    faces = rng.integers(0, 60, (400, 3)).astype(np.int32)
    face, depth, _, _ = ovo.drawlist(cloud, faces, 50.0, 50.0, 32.0, 32.0, backface_cull=False)
    assert np.all(np.diff(depth) <= 0)
    same = np.diff(depth) == 0
    assert same.any() and np.all(np.diff(face)[same] > 0)


def test_golden_overlay(ovo, synth, model, faces):
    """The committed fixture (made by tests/golden/make_overlay_golden.py with this restatement) still reproduces."""
    g = json.load(open(GOLDEN))
    clouds = posed_clouds(synth, model, 1, g["seed"]).astype(np.float32)
    img = np.zeros((g["height"], g["width"], 3), np.uint8)
    ovo.render(clouds[0], faces, img, *synth.camera_intrinsics(g["width"], g["height"]))
    assert int((img[..., 0] > 0).sum()) == g["covered"]
    assert hashlib.sha256(img.tobytes()).hexdigest() == g["sha256"]


# ---------------------------------------------------------------------------------------------------- GPU
def gpu_render(api, faces, clouds, bg, intr, **kw):
    F, H, W = bg.shape[:3]
    ov = api.Overlay(faces, clouds.shape[1], W, H, max_frames=F)
    out = bg.copy()
    ov.render(clouds, out, intr, **kw)
    return ov, out


def assert_same(got, want):
    if not np.array_equal(got, want):
        bad = np.argwhere(np.any(got != want, axis=-1))
        pytest.fail(f"{len(bad)} pixels differ; first (frame,row,col) {bad[:5].tolist()} got "
                    f"{got[tuple(bad[0])].tolist()} want {want[tuple(bad[0])].tolist()}")


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_gpu_drawlist_matches(api, ovo, synth, model, faces, dtype):
    clouds = posed_clouds(synth, model, 3, 11).astype(dtype)
    intr = synth.camera_intrinsics(1920, 1080)
    ov, _ = gpu_render(api, faces, clouds, np.zeros((3, 1080, 1920, 3), np.uint8), intr)
    for f in range(3):
        face, pts, gray = ov.drawlist(f)
        oface, _, opts, ogray = ovo.drawlist(clouds[f].astype(np.float64), faces, *intr)
        assert np.array_equal(face, oface) and np.array_equal(pts, opts) and np.array_equal(gray, ogray)


@pytest.mark.gpu
@pytest.mark.parametrize("W,H", [(1920, 1080), (480, 270), (333, 217)])
def test_gpu_image_matches(api, ovo, synth, model, faces, W, H):
    F = 3
    clouds = posed_clouds(synth, model, F, 5).astype(np.float32)
    intr = synth.camera_intrinsics(W, H)
    bg = np.random.default_rng(1).integers(0, 256, (F, H, W, 3), dtype=np.uint8)   # a video frame underneath
    _, got = gpu_render(api, faces, clouds, bg, intr)
    want = oracle_images(ovo, clouds.astype(np.float64), faces, bg, intr)
    assert (got != bg).any()
    assert_same(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("depth,shift", [(3.0, (1.6, 0.0)), (3.0, (-1.7, 0.9)), (3.0, (0.0, -1.2)), (0.9, (0.0, 0.0)),
                                         (0.35, (0.1, 0.2)), (40.0, (0.0, 0.0)), (150.0, (3.0, 2.0))])
def test_gpu_borders_closeups_and_specks(api, ovo, synth, model, faces, depth, shift):
    """Bodies leaving the image on every side, close-ups (triangles over many tiles, vertices behind the camera) and
    far bodies (every triangle in a handful of tiles: tile lists longer than one LDS pass)."""
    W, H = 640, 360
    clouds = posed_clouds(synth, model, 2, 7, depth=depth, shift=shift)
    intr = synth.camera_intrinsics(W, H)
    bg = np.full((2, H, W, 3), 30, np.uint8)
    _, got = gpu_render(api, faces, clouds, bg, intr)
    assert_same(got, oracle_images(ovo, clouds, faces, bg, intr))


@pytest.mark.gpu
def test_gpu_flags_and_errors(api, ovo, synth, model, faces):
    W, H = 480, 270
    clouds = posed_clouds(synth, model, 1, 2)
    intr = synth.camera_intrinsics(W, H)
    bg = np.zeros((1, H, W, 3), np.uint8)
    _, got = gpu_render(api, faces, clouds, bg, intr, backface_cull=False)
    assert_same(got, oracle_images(ovo, clouds, faces, bg, intr, cull=False))
    ov, got = gpu_render(api, faces, clouds, bg, intr, fill=False)
    assert np.array_equal(got, bg)                                  # nothing is drawn without fill (wireframe is off)
    assert len(ov.drawlist(0)[0]) == len(ovo.drawlist(clouds[0], faces, *intr)[0])
    # the wireframe branch (RenderSMPLMesh.h:106-109): gray-40 outlines after each fill, and outlines alone
    _, got = gpu_render(api, faces, clouds, bg, intr, wireframe=True)
    assert_same(got, oracle_images(ovo, clouds, faces, bg, intr, wire=True))
    _, got = gpu_render(api, faces, clouds, bg, intr, fill=False, wireframe=True, backface_cull=False)
    assert_same(got, oracle_images(ovo, clouds, faces, bg, intr, fill=False, cull=False, wire=True))
    with pytest.raises(api.BodyfitError):
        ov.render(np.concatenate([clouds, clouds]), np.zeros((2, H, W, 3), np.uint8), intr)   # > max_frames
    with pytest.raises(api.BodyfitError):
        api.Overlay(np.array([[0, 1, 99999]], np.int32), model.n_verts, W, H)
    # the one-frame form with the reference's argument order
    img = np.zeros((H, W, 3), np.uint8)
    api.renderSMPLMesh(clouds[0].T, faces, img, *intr)
    assert_same(img[None], oracle_images(ovo, clouds, faces, bg, intr))


@pytest.mark.gpu
@pytest.mark.parametrize("n_faces,n_verts,W,H,seed", [(1, 3, 40, 30, 0), (300, 80, 64, 48, 1), (5000, 900, 200, 120, 2),
                                                      (8192, 3000, 97, 61, 3), (8193, 3000, 128, 128, 4),
                                                      (20000, 500, 160, 90, 5)])
def test_gpu_random_triangle_soup(api, ovo, n_faces, n_verts, W, H, seed):
    """Random triangles of every size and shape (slivers, repeated vertices, equal depths, vertices behind the camera),
    including face counts around the 8192-face sort chunk."""
    rng = np.random.default_rng(seed)
    cloud = rng.uniform(-1.5, 1.5, (n_verts, 3))
    cloud[:, 2] = np.round(rng.uniform(-0.2, 4.0, n_verts), 2)       # some behind the camera, many depth ties
    fc = rng.integers(0, n_verts, (n_faces, 3)).astype(np.int32)
    fc[::7, 1] = fc[::7, 0]                                          # degenerate: repeated vertex
    intr = (0.9 * W, 0.9 * W, W / 2, H / 2)
    bg = rng.integers(0, 256, (1, H, W, 3), dtype=np.uint8)
    wire = bool(seed & 1)
    ov, got = gpu_render(api, fc, cloud[None], bg, intr, backface_cull=False, wireframe=wire)
    face, pts, gray = ov.drawlist(0)
    oface, _, opts, ogray = ovo.drawlist(cloud, fc, *intr, backface_cull=False)
    assert np.array_equal(face, oface) and np.array_equal(pts, opts) and np.array_equal(gray, ogray)
    assert_same(got, oracle_images(ovo, cloud[None], fc, bg, intr, cull=False, wire=wire))


@pytest.mark.gpu
def test_gpu_overlay_of_a_writeback(api, ovo, synth, model, faces, gpu_model):
    """The resident float cloud of bodyfit_writeback_batch goes straight into the overlay (no host round trip)."""
    import torch
    F, W, H = 4, 960, 540
    seq = synth.make_sequence(model, F, seed=4)
    intr = synth.camera_intrinsics(W, H)
    p = api.Problem.from_sequence(gpu_model, seq, n_cols=86, use_shape=True, want_mesh=True)
    x = random_params(np.random.default_rng(0), F); x[:, 0] = 1.0
    wb = p.writeback(x, np.zeros(10), want_cloud=True)
    v = p.views()
    ov = api.Overlay(faces, model.n_verts, W, H, max_frames=F)
    imgs = torch.zeros((F, H, W, 3), dtype=torch.uint8, device="cuda")
    ov.render_device(v.cloud, False, v.cloud_frame_stride, F, imgs.data_ptr(), intr)
    torch.cuda.synchronize()
    want = oracle_images(ovo, wb["cloud"].astype(np.float64), faces, np.zeros((F, H, W, 3), np.uint8), intr)
    assert_same(imgs.cpu().numpy(), want)
    t = ov.last_timing()
    assert all(t[k] >= 0 for k in ("faces", "order", "binning", "tiles"))
