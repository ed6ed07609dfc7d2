"""Deterministic synthetic inputs for the parity tests and bench.py.

Scene: the inside of an axis-aligned box room plus a sphere, ray cast analytically into 16-bit
millimetre depth images (0 = no return), the way a Kinect stream would look to Integrate; and
seeded surfel fragments with analytic normals for BuildCorrespondence.  No file or network input.

The renderer is written with torch ops only so that bench.py can generate thousands of frames
directly in HBM (device="cuda") while the CPU tests use the same code on small counts.  This is
data plumbing, not part of the measured path.
"""
import math

import numpy as np
import torch

SEED = 20150722
ROOM_LO, ROOM_HI = 0.01, 2.99            # keeps every surface inside 8x8x8 volume units ("512^3")
SPHERE_C, SPHERE_R = (1.5, 1.5, 1.5), 0.4
CAM = (525.0, 525.0, 319.5, 239.5)       # reference defaults, TSDFVolumeUnit.h:69


def look_at(eye, forward, up=(0.0, 1.0, 0.0)):
    """world_T_camera (4x4 float64) for a camera at `eye` looking along `forward` (camera +z), +y down-ish."""
    f = np.asarray(forward, np.float64)
    f = f / np.linalg.norm(f)
    u = np.asarray(up, np.float64)
    r = np.cross(u, f)
    r = r / np.linalg.norm(r)
    d = np.cross(f, r)
    T = np.eye(4)
    T[:3, 0], T[:3, 1], T[:3, 2], T[:3, 3] = r, d, f, np.asarray(eye, np.float64)
    return T


def circle_trajectory(n, radius=0.6, center=(1.5, 1.5, 1.5), revolutions=1.0, radius_drift=0.0):
    """n camera poses on a circle around the room centre looking outward with a slow pitch wobble
    (config 2 of BASELINE.json; config 4 adds a radius drift)."""
    out = np.empty((n, 4, 4), np.float64)
    c = np.asarray(center, np.float64)
    for i in range(n):
        th = 2.0 * math.pi * revolutions * i / max(n, 1)
        r = radius + radius_drift * i / max(n, 1)
        eye = c + r * np.array([math.cos(th), 0.05 * math.sin(2.0 * th), math.sin(th)])
        fwd = np.array([math.cos(th), 0.15 * math.sin(3.0 * th), math.sin(th)])
        out[i] = look_at(eye, fwd)
    return out


def render_depth(poses, cols=640, rows=480, cam=CAM, lo=ROOM_LO, hi=ROOM_HI, sphere=True, max_depth=4.0,
                 device="cpu", chunk=64):
    """Ray cast the scene for every world_T_camera in `poses` [n,4,4].  Returns uint16 [n, rows*cols]
    (torch tensor on `device`): round(z*1000), 0 where there is no hit closer than max_depth."""
    fx, fy, cx, cy = cam
    P = torch.as_tensor(np.asarray(poses, np.float64), dtype=torch.float64, device=device).reshape(-1, 4, 4)
    n = P.shape[0]
    u = torch.arange(cols, dtype=torch.float64, device=device)
    v = torch.arange(rows, dtype=torch.float64, device=device)
    dx = ((u - cx) / fx).repeat(rows)                      # row-major pixel order
    dy = ((v - cy) / fy).repeat_interleave(cols)
    out = torch.empty((n, rows * cols), dtype=torch.int32, device=device)
    inf = float("inf")
    # Only elementwise IEEE float64 ops in a fixed order (no matmul / library reductions), so the images
    # are bit-reproducible across machines and devices; tests/golden digests depend on that.
    for s in range(0, n, chunk):
        R = P[s:s + chunk, :3, :3]
        o = P[s:s + chunk, :3, 3]                                       # [b,3]
        d = [(R[:, a, 0:1] * dx[None, :] + R[:, a, 1:2] * dy[None, :]) + R[:, a, 2:3] for a in range(3)]   # camera z = 1 -> t == depth
        t = torch.full_like(d[0], inf)
        for a in range(3):
            da = d[a]
            wall = torch.where(da > 0, torch.full_like(da, hi), torch.full_like(da, lo))
            ta = (wall - o[:, a:a + 1]) / da
            ta = torch.where(da.abs() < 1e-12, torch.full_like(ta, inf), ta)
            t = torch.minimum(t, torch.where(ta > 0, ta, torch.full_like(ta, inf)))
        if sphere:
            oc = [o[:, a:a + 1] - SPHERE_C[a] for a in range(3)]        # [b,1] each
            A = (d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]
            B = 2.0 * ((d[0] * oc[0] + d[1] * oc[1]) + d[2] * oc[2])
            Cc = ((oc[0] * oc[0] + oc[1] * oc[1]) + oc[2] * oc[2]) - SPHERE_R * SPHERE_R
            disc = B * B - 4.0 * A * Cc
            sq = torch.sqrt(disc.clamp(min=0))
            ts = (-B - sq) / (2.0 * A)
            ts = torch.where((disc > 0) & (ts > 1e-6), ts, torch.full_like(ts, inf))
            t = torch.minimum(t, ts)
        mm = torch.floor(t * 1000.0 + 0.5)
        mm = torch.where(torch.isfinite(mm) & (t <= max_depth) & (mm < 65535.5), mm, torch.zeros_like(mm))
        out[s:s + chunk] = mm.to(torch.int32)
    return out.to(torch.uint16) if hasattr(torch, "uint16") else out.to(torch.int16)


def to_numpy_u16(t):
    """torch uint16/int16 tensor -> numpy uint16 (host)."""
    a = t.cpu()
    if a.dtype == torch.int16:
        return a.numpy().view(np.uint16)
    return a.view(torch.int16).numpy().view(np.uint16)


# ----------------------------------------------------------------------------------------------
# Fragment bookkeeping of the elastic pipeline (IntegrateApp.cpp:64-78, 242-243; the kinfu fragment
# convention: every fragment's first camera sits at basepose = (L/2, L/2, -0.3) of its own L^3 cube,
# BuildCorrespondence/CorresApp.cpp:45-48).
def basepose(length=3.0):
    B = np.eye(4)
    B[0, 3], B[1, 3], B[2, 3] = length / 2.0, length / 2.0, -0.3
    return B


def split_trajectory(world_T_cam, interval=50, length=3.0):
    """world_T_cam [n,4,4] -> (pose [n/interval], seg [n]) with world_T_cam[i*interval+j] == pose[i] @ seg[...]."""
    n = world_T_cam.shape[0]
    num = n // interval
    B = basepose(length)
    pose = np.empty((num, 4, 4))
    seg = np.empty((num * interval, 4, 4))
    for i in range(num):
        pose[i] = world_T_cam[i * interval] @ np.linalg.inv(B)
        pinv = np.linalg.inv(pose[i])
        for j in range(interval):
            seg[i * interval + j] = pinv @ world_T_cam[i * interval + j]
    return pose, seg


def control_grids(pose, resolution=8, length=3.0, amplitude=0.005, seed=SEED):
    """One control lattice per fragment: the undeformed lattice pose[0]^-1 * pose[i] * (i,j,k)*length/res
    (FragmentOptimizer/OptApp.cpp:691-703 defines the regular lattice; IntegrateApp.cpp:243 fixes the frame)
    plus a smooth seeded sinusoidal deformation of `amplitude` metres.  float32 [num, (res+1)^3, 3],
    vertex order i + j*(res+1) + k*(res+1)^2 (ControlGrid.h:41-43)."""
    rng = np.random.RandomState(seed)
    num = pose.shape[0]
    n1 = resolution + 1
    ul = length / resolution
    k, j, i = np.meshgrid(np.arange(n1), np.arange(n1), np.arange(n1), indexing="ij")
    verts = np.stack([i.reshape(-1), j.reshape(-1), k.reshape(-1)], axis=1).astype(np.float64) * ul   # index = i + j*n1 + k*n1^2
    p0inv = np.linalg.inv(pose[0])
    out = np.empty((num, n1 ** 3, 3), np.float32)
    for g in range(num):
        M = p0inv @ pose[g]
        w = verts @ M[:3, :3].T + M[:3, 3]
        ph = rng.uniform(0, 2 * math.pi, size=3)
        fr = rng.uniform(1.0, 2.5, size=3)
        defo = amplitude * np.stack([np.sin(fr[0] * verts[:, 1] + ph[0]), np.sin(fr[1] * verts[:, 2] + ph[1]),
                                     np.sin(fr[2] * verts[:, 0] + ph[2])], axis=1)
        out[g] = (w + defo).astype(np.float32)
    return out


# ----------------------------------------------------------------------------------------------
def sample_fragment(world_T_frag, n_points, seed, length=3.0, lo=ROOM_LO, hi=ROOM_HI, sphere=True):
    """n_points surfels sampled uniformly (seeded) on the scene surfaces, expressed in the fragment's
    own cube frame and kept only if inside [0,length]^3.  Returns (xyz float32 [m,3], normals float32 [m,3])."""
    rng = np.random.RandomState(seed)
    side = hi - lo
    areas = [side * side] * 6 + ([4 * math.pi * SPHERE_R ** 2] if sphere else [])
    p = np.array(areas) / sum(areas)
    which = rng.choice(len(areas), size=n_points, p=p)
    a = rng.uniform(lo, hi, size=n_points)
    b = rng.uniform(lo, hi, size=n_points)
    pts = np.empty((n_points, 3))
    nrm = np.zeros((n_points, 3))
    for f in range(6):
        m = which == f
        ax, sgn = f // 2, f % 2
        o = [x for x in range(3) if x != ax]
        pts[m, ax] = hi if sgn else lo
        pts[m, o[0]] = a[m]
        pts[m, o[1]] = b[m]
        nrm[m, ax] = -1.0 if sgn else 1.0                                 # pointing into the room
    if sphere:
        m = which == 6
        v = rng.normal(size=(int(m.sum()), 3))
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        pts[m] = np.asarray(SPHERE_C) + SPHERE_R * v
        nrm[m] = v
    F = np.linalg.inv(world_T_frag)
    q = pts @ F[:3, :3].T + F[:3, 3]
    qn = nrm @ F[:3, :3].T
    keep = np.all((q >= 0.0) & (q <= length), axis=1)
    return q[keep].astype(np.float32), qn[keep].astype(np.float32)


def _sample_fragment_torch(world_T_frag, n_points, seed, length, device):
    """sample_fragment with torch ops (any device): same distribution, its own random stream.  Used where hundreds of
    thousands of surfels per fragment are needed many times over (bench.py, the configs[2]-size GPU test)."""
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    side = ROOM_HI - ROOM_LO
    areas = torch.tensor([side * side] * 6 + [4 * math.pi * SPHERE_R ** 2], dtype=torch.float64)
    edges = (torch.cumsum(areas, 0) / areas.sum()).to(device)
    r = torch.rand(n_points, generator=g, device=device, dtype=torch.float64)
    face = torch.bucketize(r, edges[:-1], right=True)                       # 0..5 walls, 6 sphere
    ab = torch.rand((n_points, 2), generator=g, device=device, dtype=torch.float64) * side + ROOM_LO
    v = torch.randn((n_points, 3), generator=g, device=device, dtype=torch.float64)
    v = v / v.norm(dim=1, keepdim=True)
    ax, sgn = face // 2, face % 2
    wall = torch.where(sgn == 1, torch.full_like(r, ROOM_HI), torch.full_like(r, ROOM_LO))
    pts = torch.empty((n_points, 3), device=device, dtype=torch.float64)
    nrm = torch.zeros((n_points, 3), device=device, dtype=torch.float64)
    for a in range(3):
        o = [x for x in range(3) if x != a]
        m = (ax == a) & (face < 6)
        pts[:, a] = torch.where(m, wall, pts[:, a])
        nrm[:, a] = torch.where(m, torch.where(sgn == 1, -torch.ones_like(r), torch.ones_like(r)), nrm[:, a])
        pts[:, o[0]] = torch.where(m, ab[:, 0], pts[:, o[0]])
        pts[:, o[1]] = torch.where(m, ab[:, 1], pts[:, o[1]])
    sp = (face == 6)[:, None]
    c = torch.tensor(SPHERE_C, device=device, dtype=torch.float64)
    pts = torch.where(sp, c + SPHERE_R * v, pts)
    nrm = torch.where(sp, v, nrm)
    F = torch.as_tensor(np.linalg.inv(world_T_frag), device=device, dtype=torch.float64)
    q = pts @ F[:3, :3].T + F[:3, 3]
    qn = nrm @ F[:3, :3].T
    keep = ((q >= 0.0) & (q <= length)).all(dim=1)
    return q[keep].to(torch.float32).cpu().numpy(), qn[keep].to(torch.float32).cpu().numpy()


def fragment_set(num, target_points=250000, seed=SEED, length=3.0, radius=0.0, device="cpu"):
    """`num` DISTINCT fragments of target_points surfels each (configs[2] / configs[4] shape), independently sampled.
    radius == 0 (default): fragment i's length^3 cube is centred on the room centre and turned by 2 pi i / num about the
    vertical axis, so every fragment holds floor, ceiling, the sphere and a differently clipped part of all four walls: any
    two fragments overlap and every pair constrains all six degrees of freedom.  (Fragments that see nothing but one wall
    plus floor and ceiling slide along the wall under point-to-plane ICP -- the reference's just as well -- and a parity
    test on such a pair compares two arbitrary answers.)
    radius > 0: fragment i is what a level camera at that distance from the centre, looking outward at angle 2 pi i / num,
    sees inside its own cube (frame = camera pose x basepose^-1, the kinfu convention of CorresApp.cpp:45-48): neighbours
    overlap, opposite fragments do not (the all-pairs scene of configs[4] with its pre-check rejects).
    Returns [(xyz float32 [m,3], normals float32 [m,3], world_T_frag float64 4x4)]."""
    out = []
    Binv = np.linalg.inv(basepose(length))
    for i in range(num):
        th = 2.0 * math.pi * i / num
        d = np.array([math.cos(th), 0.0, math.sin(th)])
        if radius > 0:
            F = look_at(np.array([1.5, 1.5, 1.5]) + radius * d, d) @ Binv
        else:
            C = np.eye(4)
            C[:3, 3] = -length / 2.0
            F = look_at(np.array([1.5, 1.5, 1.5]), d) @ C
        x, n = _sample_fragment_torch(F, (5 if radius > 0 else 2) * target_points, seed + 7919 * i + 1, length, device)
        out.append((x[:target_points], n[:target_points], F))                   # (samples are i.i.d.: a prefix is a sample)
    return out


def perturbation(seed, max_rot_deg=2.0, max_trans=0.02):
    """Small seeded rigid perturbation (4x4 float64)."""
    rng = np.random.RandomState(seed)
    ax = rng.normal(size=3)
    ax /= np.linalg.norm(ax)
    ang = math.radians(max_rot_deg) * rng.uniform(0.3, 1.0)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = np.eye(3) + math.sin(ang) * K + (1 - math.cos(ang)) * (K @ K)
    t = rng.normal(size=3)
    t = t / np.linalg.norm(t) * max_trans * rng.uniform(0.3, 1.0)
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, t
    return T


# ----------------------------------------------------------------------------------------------
def make_scenario(n_frames, interval=50, warp=True, resolution=8, length=3.0, amplitude=0.005, revolutions=None,
                  frame_offset=0, total_frames=None, radius_drift=0.0, device="cpu", seed=SEED, room=(ROOM_LO, ROOM_HI),
                  render_frames=None):
    """Everything one Integrate run needs (BASELINE.json configs 1/2/4), all in memory:
      depth  uint16 [n, 480*640] torch tensor on `device`
      traj   float64 [n,4,4]  world_T_camera as the reference composes it: pose[i] * seg[i*interval+j]
      pose   float64 [n/interval,4,4], seg float64 [n,4,4]
      grids  float32 [n/interval, (res+1)^3, 3] (None when warp is False)
    frame_offset/total_frames select a window of a longer trajectory (multi-GPU frame split).
    room = (lo, hi) of the box room; the default keeps every surface inside 8x8x8 volume units ("512^3"), a larger room
    (config 4) makes the hashed unit grid grow past 512 units.
    render_frames = sorted 0-based frame indices: only those frames are ray cast (depth [len(render_frames), 480*640], in that
    order; sc["rendered"] names them) while traj / pose / seg / grids still describe ALL n_frames -- a sampled stream of a long
    path (the configs[3] / configs[4] parity checks integrate a few hundred frames of a 10 000-frame job)."""
    from .tsdf import mat4_mul
    total = total_frames if total_frames is not None else n_frames
    revs = revolutions if revolutions is not None else max(1.0, total / 3000.0)
    full = circle_trajectory(total, revolutions=revs, radius_drift=radius_drift)
    w = full[frame_offset:frame_offset + n_frames]
    num = n_frames // interval
    assert num * interval == n_frames, "n_frames must be a multiple of interval"
    pose, seg = split_trajectory(w, interval, length)
    traj = np.empty_like(w)
    for i in range(num):
        for j in range(interval):
            traj[i * interval + j] = mat4_mul(pose[i], seg[i * interval + j])
    grids = control_grids(pose, resolution, length, amplitude, seed) if warp else None
    if render_frames is None:
        depth, rendered = render_depth(w, lo=room[0], hi=room[1], device=device), None
    else:
        rendered = np.asarray(render_frames, np.int64)
        depth = render_depth(w[rendered], lo=room[0], hi=room[1], device=device)
    return dict(depth=depth, traj=traj, pose=pose, seg=seg, grids=grids, interval=interval, resolution=resolution,
                length=length, n=n_frames, rendered=rendered)


def warp_arrays(sc, lo=0, hi=None):
    """The er_warp arrays for frames [lo, hi) of a scenario (IntegrateApp.cpp:242-243,251)."""
    from .tsdf import reproject_matrix
    hi = sc["n"] if hi is None else hi
    gi = np.array([f // sc["interval"] for f in range(lo, hi)], np.int32)
    madj = np.stack([reproject_matrix(sc["traj"][f], sc["traj"][0], sc["seg"][0]) for f in range(lo, hi)])
    return dict(ctr=sc["grids"], resolution=sc["resolution"], length=np.float32(sc["length"]), grid_index=gi,
                seg=sc["seg"][lo:hi].copy(), madj=madj)


# ---- pair lists over a fragment_set (configs[2] shape) ----------------------------------------------------------------
def pair_list(frs, n_pairs, rot, trans, seed0):
    """Pair k = fragment a = k mod F with its 1st / 2nd / 3rd neighbour b, guess = ground truth o perturbation(seed0 + k)."""
    n_frag = len(frs)
    out = []
    for k in range(n_pairs):
        a = k % n_frag
        b = (a + 1 + (k // n_frag) % 3) % n_frag
        out.append((a, b, np.linalg.inv(frs[a][2]) @ frs[b][2] @ perturbation(seed0 + k, rot, trans)))
    return out


def chain_pair_list(frs, n_pairs, rot, trans, seed0):
    """Pairs over fragments that follow each other along an OPEN path (kinfu_fragment_set over part of a circle): every fragment with its next
    neighbour, then with the one after, then the third, ... -- no wrap-around from the last fragment to the first.  Same guesses as pair_list."""
    n_frag, out, step = len(frs), [], 1
    while len(out) < n_pairs and step < n_frag:
        for a in range(n_frag - step):
            if len(out) < n_pairs:
                out.append((a, a + step, np.linalg.inv(frs[a][2]) @ frs[a + step][2] @ perturbation(seed0 + len(out), rot, trans)))
        step += 1
    return out


def hard_pair_list(frs, n_pairs):
    """The HARD list of bench.py's icp.hard_set and of tests/test_icp_gpu.py: guesses up to 6 deg / 6 cm off the ground truth -- three
    times the configs[2] perturbation -- so that PCL's 20-iteration budget, the transform criterion and the iteration limit are all
    reached (BuildCorrespondence/CorresApp.cpp:295-306)."""
    return pair_list(frs, n_pairs, 6.0, 0.06, 1700)


def config2_pair_list(frs, n_pairs):
    """configs[2]'s list: guesses <= 2 deg / 2 cm off the ground truth."""
    return pair_list(frs, n_pairs, 2.0, 0.02, 700)


# ---- fragments that look like fragments: what cloud_bin_<i>.pcd is in the real pipeline ----------------------------------
def kinfu_camera_path(i, num, frames=50, radius=1.1, arc_deg=8.0, pitch_deg=15.0):
    """world_T_camera [frames,4,4] of fragment i of num: a hand-held sweep -- the camera stands `radius` from the room centre at angle
    2 pi i / num, walks a short arc around it looking INWARD and tilts up and down once (so that the fragment holds the sphere, the far
    wall, parts of both side walls and of floor and ceiling: every pair of overlapping fragments constrains all six degrees of freedom)."""
    out = np.empty((frames, 4, 4), np.float64)
    c = np.array([1.5, 1.5, 1.5])
    for j in range(frames):
        th = 2.0 * math.pi * i / num + math.radians(arc_deg) * j / frames
        d = np.array([math.cos(th), 0.0, math.sin(th)])
        pitch = math.tan(math.radians(pitch_deg)) * math.sin(2.0 * math.pi * j / frames + 0.7 * i)
        out[j] = look_at(c + radius * d, np.array([-d[0], pitch, -d[2]]))
    return out


def kinfu_fragment(i, num, target_points=250000, frames=50, noise_mm=0.0, density="inv_z2", seed=SEED, length=3.0, device=0):
    """One fragment the way the pipeline makes them (README.txt:41-60; BuildCorrespondence/CorresApp.cpp:82-99 reads the result): `frames`
    consecutive depth images of a hand-held sweep (kinfu_camera_path; optional Gaussian depth noise of noise_mm, seeded) are integrated
    into a TSDF volume in the fragment's own cube frame -- first camera at basepose, the kinfu convention -- by THIS library's Integrate
    path, the zero crossings of the volume are extracted (er_tsdf_extract_surface: points on the voxel lattice's edges, 5.9 mm apart), and
    every point gets the normalised TSDF gradient (central differences at its nearest voxel) as its normal -- NaN where a neighbour voxel
    was never observed, as at the border of what the sweep saw.  Points outside the cube [0, length) are dropped (PointCloud::LoadFromPCDFile stops
    there); density = "inv_z2" thins the survivors with probability ~ 1 / z^2 of the first camera (what one depth image of the sweep
    gives: dense close to the camera, > 10 x sparser on the far wall), "tsdf" keeps the lattice density.  NOT part of the measured path:
    a generator of realistic inputs that happens to need a GPU.
    Returns (xyz float32 [m,3], normals float32 [m,3] WITH NaN rows, world_T_frag float64 4x4, stats dict)."""
    from . import tsdf as _tsdf
    dev = device if isinstance(device, str) else "cuda:%d" % device
    W = kinfu_camera_path(i, num, frames)
    F = W[0] @ np.linalg.inv(basepose(length))
    Finv = np.linalg.inv(F)
    seg = np.stack([Finv @ W[j] for j in range(frames)])
    depth = render_depth(W, device=dev)
    if noise_mm > 0:
        g = torch.Generator(device=dev)
        g.manual_seed(int(seed) + 104729 * i + 17)
        d32 = depth.view(torch.int16).to(torch.int32) & 0xffff
        noisy = torch.round(d32.to(torch.float32) + noise_mm * torch.randn(d32.shape, generator=g, device=dev)).clamp(1, 65535).to(torch.int32)
        d16 = torch.where(d32 > 0, noisy, d32).to(torch.int16)            # (<= 4000 mm: no wrap)
        depth = d16 if depth.dtype == torch.int16 else d16.view(depth.dtype)
    if dev != "cpu":
        torch.cuda.synchronize()
    # camera file of the run: the reference's intrinsics with integration_trunc_ = 4 m (TSDFVolumeUnit.h:66-69; ScaleDepth drops rays longer than
    # that, TSDFVolume.cpp:28-30) -- with the default 2.5 m a sweep across a 3 m room loses its far wall and a fragment shrinks to ~40 k points
    cam = np.array([CAM[0], CAM[1], CAM[2], CAM[3], 2.5, 4.0], np.float32)
    vol = _tsdf.TSDFVolume(640, 480, cam, max_units=1024, device=0 if isinstance(device, str) else device)
    vol.IntegrateFrames(None, seg, None, device_ptr=depth.data_ptr())
    pts = vol.extract_surface()                                           # [n, 4] = x y z axis, metres, fragment frame
    ul = length / 512.0
    S = torch.zeros((512, 512, 512), dtype=torch.float32, device=dev)
    Wt = torch.zeros((512, 512, 512), dtype=torch.bool, device=dev)
    for key in vol.unit_keys():
        key = int(key)
        ux, uy, uz = (key >> 18) - 256, ((key >> 9) & 511) - 256, (key & 511) - 256
        if not (0 <= ux < 8 and 0 <= uy < 8 and 0 <= uz < 8):
            continue                                                      # outside the fragment's cube: dropped below anyway
        s, w = vol.read_unit(key)
        sl = (slice(ux * 64, ux * 64 + 64), slice(uy * 64, uy * 64 + 64), slice(uz * 64, uz * 64 + 64))
        S[sl] = torch.from_numpy(s.reshape(64, 64, 64)).to(dev)
        Wt[sl] = torch.from_numpy(w.reshape(64, 64, 64) != 0).to(dev)
    vol.close()
    P = torch.from_numpy(pts[:, :3].copy()).to(dev)
    inside = ((P >= 0.0) & (P < length)).all(dim=1)           # (PointCloud::GetCoordinate, PointCloud.h:101-110: floor(p / unit) must stay below the resolution --
    v = torch.round(P.to(torch.float64) / ul).to(torch.int64)    #  a zero crossing ON the cube's far face, voxel 512, is already outside)
    ok = inside & ((v >= 1) & (v <= 510)).all(dim=1)
    vc = v.clamp(1, 510)
    g3, seen = [], Wt[vc[:, 0], vc[:, 1], vc[:, 2]]
    for a in range(3):
        e = torch.zeros(3, dtype=torch.int64, device=dev)
        e[a] = 1
        hi, lo = vc + e, vc - e
        g3.append(S[hi[:, 0], hi[:, 1], hi[:, 2]] - S[lo[:, 0], lo[:, 1], lo[:, 2]])
        seen = seen & Wt[hi[:, 0], hi[:, 1], hi[:, 2]] & Wt[lo[:, 0], lo[:, 1], lo[:, 2]]
    G = torch.stack(g3, dim=1)
    nrm = G.norm(dim=1, keepdim=True)
    N = torch.where((seen & ok)[:, None] & (nrm > 0), G / nrm.clamp(min=1e-30), torch.full_like(G, float("nan")))
    keep = torch.nonzero(inside).reshape(-1)
    raw = int(keep.numel())
    if density == "inv_z2" and raw > target_points:
        zc = (P[keep, 2] + 0.3).clamp(min=0.3)                            # depth from the fragment's first camera (basepose: z = -0.3)
        gen = torch.Generator(device=dev)
        gen.manual_seed(int(seed) + 7919 * i + 3)
        pick = torch.multinomial((1.0 / (zc * zc)).to(torch.float32), target_points, replacement=False, generator=gen)
        keep = keep[torch.sort(pick).values]                              # (the file keeps the extraction's order)
    elif raw > target_points:
        gen = torch.Generator(device=dev)
        gen.manual_seed(int(seed) + 7919 * i + 3)
        keep = keep[torch.sort(torch.randperm(raw, generator=gen, device=dev)[:target_points]).values]
    xyz = P[keep].cpu().numpy().astype(np.float32)
    nr = N[keep].cpu().numpy().astype(np.float32)
    nan = np.isnan(nr).any(axis=1)
    stats = dict(zero_crossings=int(pts.shape[0]), inside_cube=raw, kept=int(xyz.shape[0]), nan_normals=int(nan.sum()),
                 nan_fraction=float(nan.mean()) if len(nan) else 0.0)
    del S, Wt
    return xyz, nr, F, stats


def kinfu_fragment_set(num, target_points=250000, noise_mm=0.0, density="inv_z2", seed=SEED, device=0, stride=1, total=None):
    """`num` fragments kinfu_fragment(i * stride, total or num * stride) with the NaN-normal points removed the way CCorresApp::LoadData does
    (CorresApp.cpp:88-97: `if ( !pcl_isnan( normal_x ) ) push_back`), in fragment_set's format [(xyz, normals, world_T_frag)] + a stats list."""
    out, stats = [], []
    for k in range(num):
        x, n, F, st = kinfu_fragment(k * stride, total or num * stride, target_points, noise_mm=noise_mm, density=density, seed=seed, device=device)
        ok = ~np.isnan(n).any(axis=1)
        out.append((np.ascontiguousarray(x[ok]), np.ascontiguousarray(n[ok]), F))
        st["points_after_nan_filter"] = int(ok.sum())
        stats.append(st)
    return out, stats


def cell_occupancy(xyz, cell=0.03 * 1.001):
    """Points per OCCUPIED cell of the uniform search grid er_cloud_create builds over a cloud (cell edge 1.001 x reg_dist): (max, mean, cells)."""
    q = np.floor((xyz - xyz.min(axis=0)) / np.float32(cell)).astype(np.int64)
    key = (q[:, 2] * (q[:, 1].max() + 1) + q[:, 1]) * (q[:, 0].max() + 1) + q[:, 0]
    cnt = np.unique(key, return_counts=True)[1]
    return int(cnt.max()), float(cnt.mean()), int(cnt.size)
