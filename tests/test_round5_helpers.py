"""CPU tests of the round-5 test infrastructure and data generators: the sampled-stream helpers of the configs[3] / configs[4] parity checks
(oracle/refcheck.py), the open-path pair list, the cell occupancy figure and the kinfu-like fragment generator (synth.kinfu_fragment) -- the
latter on a stand-in volume backed by the CPU oracle (oracle/tsdf_oracle.c + the numpy restatement of er_tsdf_extract_surface), so that the
generator's own logic (dense grid assembly, gradient normals, NaN rule, cube bounds, thinning) is covered where no GPU is."""
import ctypes
import os

import numpy as np

from elasticreconstruction_amd import formats, synth
from oracle import refcheck


def test_sampled_frames_cover_the_job_from_its_first_to_its_last_fragment():
    for n, interval, runs, rl in ((10000, 50, 40, 10), (5000, 50, 20, 10), (200, 50, 2, 50), (150, 50, 3, 7)):
        ids = refcheck.sampled_frames(n, interval, runs, rl)
        assert len(ids) == runs * rl and np.all(np.diff(ids) > 0) and ids[0] == 0 and ids[-1] == n - 1
        frag = ids // interval
        for r in range(runs):                                  # every run stays inside one fragment
            assert len(set(frag[r * rl:(r + 1) * rl])) == 1
        assert frag[0] == 0 and frag[-1] == n // interval - 1


def test_unit_coordinates_invert_hash_key_at_negative_coordinates():
    # Integrate/TSDFVolume.cpp:50-53 + TSDFVolume.h:62-64: unit index = (voxel + 256 * 64) / 64, key = x * 512 * 512 + y * 512 + z
    rng = np.random.default_rng(3)
    vox = rng.integers(-300 * 64 + 1, 250 * 64, size=(500, 3))
    idx = (vox + 256 * 64) // 64
    idx = idx[((idx >= 0) & (idx < 512)).all(axis=1)]
    keys = idx[:, 0] * 512 * 512 + idx[:, 1] * 512 + idx[:, 2]
    c = refcheck.unit_coordinates(keys)
    assert np.array_equal(c, idx - 256) and (c < 0).any() and (c > 7).any()


def test_chain_pair_list_never_wraps_around():
    frs = [(None, None, synth.look_at((1.5, 1.5, 1.5), (np.cos(0.1 * i), 0, np.sin(0.1 * i)))) for i in range(25)]
    pl = synth.chain_pair_list(frs, 50, 2.0, 0.02, 700)
    assert len(pl) == 50 and all(a < b <= a + 3 for a, b, _ in pl)
    assert [p[:2] for p in pl[:3]] == [(0, 1), (1, 2), (2, 3)] and pl[24][:2] == (0, 2) and pl[47][:2] == (0, 3)
    assert {q for a, b, _ in pl for q in (a, b)} == set(range(25))
    for a, b, T in pl[:5]:                                    # guess = ground truth o small perturbation
        gt = np.linalg.inv(frs[a][2]) @ frs[b][2]
        assert 0 < np.abs(np.linalg.inv(gt) @ T - np.eye(4)).max() < 0.06


def test_cell_occupancy_counts_points_per_occupied_cell():
    x = np.array([[0, 0, 0]] * 5 + [[0.1, 0, 0]] * 2 + [[0.1, 0.2, 0.3]], np.float32)
    assert synth.cell_occupancy(x, 0.03) == (5, 8 / 3.0, 3)


def test_write_integrate_files_round_trip(tmp_path):
    sc = synth.make_scenario(8, interval=4, warp=True, render_frames=[0, 7])
    assert sc["depth"].shape[0] == 2 and list(sc["rendered"]) == [0, 7] and sc["traj"].shape[0] == 8
    p = refcheck.write_integrate_files(sc, str(tmp_path))
    pose, seg = formats.load_log(p[0]), formats.load_log(p[1])
    assert len(pose) == 2 + 1 and len(seg) == 8 + 4                      # one extra fragment of entries (IntegrateApp.cpp:200-203)
    assert np.abs(pose[1].T - sc["pose"][1]).max() < 1e-7 and np.abs(seg[5].T - sc["seg"][5]).max() < 1e-7
    assert formats.load_ctr(p[2], 2, 8).shape == (2, 729, 3)


class _OracleBackedVolume:
    """Stand-in for elasticreconstruction_amd.tsdf.TSDFVolume on machines without a GPU: the CPU oracle integrates, the numpy restatement of
    er_tsdf_extract_surface (tests/test_tsdf_gpu.py) extracts."""

    def __init__(self, cols=640, rows=480, cam=None, max_units=0, device=0):
        from oracle.pyoracle import OracleVolume
        self.o = OracleVolume(cols, rows, cam)

    def IntegrateFrames(self, depth, seg, warp, device_ptr=None):
        n, px = len(seg), 640 * 480
        d = np.frombuffer((ctypes.c_uint16 * (n * px)).from_address(device_ptr), np.uint16).reshape(n, px)
        for f in range(n):
            self.o.Integrate(d[f], seg[f])

    def unit_keys(self):
        return self.o.unit_keys()

    def read_unit(self, k):
        return self.o.read_unit(k)

    def extract_surface(self):
        from test_tsdf_gpu import _surface_oracle
        return _surface_oracle({int(k): self.o.read_unit(k) for k in self.o.unit_keys()})

    def close(self):
        pass


def test_kinfu_fragment_generator_on_a_cpu_backed_volume(monkeypatch):
    from elasticreconstruction_amd import tsdf
    monkeypatch.setattr(tsdf, "TSDFVolume", _OracleBackedVolume)
    x, n, F, st = synth.kinfu_fragment(3, 50, target_points=30000, frames=3, noise_mm=2.0, device="cpu")
    assert x.shape == (30000, 3) and st["zero_crossings"] >= st["inside_cube"] > 30000 and st["kept"] == 30000
    assert (x >= 0).all() and (x < 3.0).all()                            # PointCloud::GetCoordinate's cube, far face excluded
    nan = np.isnan(n).any(axis=1)
    assert 0.005 < nan.mean() < 0.3 and np.isnan(n[nan]).all()           # a NaN normal is NaN in every component (CorresApp.cpp:93 tests normal_x)
    ok = ~nan
    assert np.abs(np.linalg.norm(n[ok], axis=1) - 1).max() < 1e-5
    # the points lie on the scene, the normals agree with the analytic ones (walls: into the room, sphere: outward)
    w = x[ok].astype(np.float64) @ F[:3, :3].T + F[:3, 3]
    nw = n[ok].astype(np.float64) @ F[:3, :3].T
    dw = np.minimum(np.abs(w - synth.ROOM_LO), np.abs(w - synth.ROOM_HI))
    ds = np.abs(np.linalg.norm(w - np.asarray(synth.SPHERE_C), axis=1) - synth.SPHERE_R)
    assert np.mean(np.minimum(dw.min(axis=1), ds) < 2 * 3.0 / 512) > 0.99
    sph = ds < dw.min(axis=1)
    ns = w[sph] - np.asarray(synth.SPHERE_C)
    ns /= np.linalg.norm(ns, axis=1, keepdims=True)
    assert (nw[sph] * ns).sum(axis=1).mean() > 0.95
    ax = dw.argmin(axis=1)
    rows = np.arange(len(w))
    sgn = np.where(np.abs(w[rows, ax] - synth.ROOM_LO) < np.abs(w[rows, ax] - synth.ROOM_HI), 1.0, -1.0)
    assert (nw[rows, ax] * sgn)[~sph].mean() > 0.9
    # thinned ~ 1 / z^2 of the first camera: nearer on average than the same crossings thinned uniformly
    xu, _, _, _ = synth.kinfu_fragment(3, 50, target_points=30000, frames=3, noise_mm=2.0, density="tsdf", device="cpu")
    assert np.mean(x[:, 2]) < np.mean(xu[:, 2]) - 0.1, (np.mean(x[:, 2]), np.mean(xu[:, 2]))
    # the file order of the extraction is kept, and the world pose puts the first camera at basepose
    W = synth.kinfu_camera_path(3, 50, 3)
    assert np.abs(F @ synth.basepose() - W[0]).max() < 1e-12
