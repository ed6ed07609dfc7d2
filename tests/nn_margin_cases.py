"""Adversarial queries for the pruning margin of the GPU nearest-neighbour search (csrc/er_icp.hip: nn_block, grid_slack): the true nearest neighbour
sits just behind a cell face, straight along the x axis, and a competitor in the query's OWN cell is farther by a fraction of a micrometre.
With the absolute margin of rounds 1-4 (1e-12 m^2) the float32 face distance of such a query exceeds the pruning bound and the neighbour's cell
is skipped; the margin sized from the grid's extent keeps it.  Everything here is float32 arithmetic restated with numpy."""
import numpy as np

f32 = np.float32


def cell_of(v, org, cell):
    return np.floor(f32(f32(v - org) / cell))


def grid_slack(dim, cell):
    """csrc/er_icp.hip: grid_slack."""
    D = 2.5e-7 * (max(dim) + 2) * float(cell) + 4e-9
    return f32(1.3e5 * D * D)


def build(n_inst=40, seed=3, grid_cell=0.03):
    """-> target points [m,3], source points [n_inst,3], index of the true nearest target point per source point, per-instance records
    (xlo^2 as the kernel computes it, float32 squared distance of the competitor, of the true neighbour), grid (org, cell, dim)."""
    cell = f32(f32(grid_cell) * f32(1.001))
    lo = np.array([-1.7, -0.5, -0.5], np.float32)
    hi = np.array([1.4, 0.5, 0.5], np.float32)
    org = lo
    dim = [int(np.floor(f32(hi[a] - lo[a]) / cell)) + 1 for a in range(3)]
    rng = np.random.default_rng(seed)
    tgt, src, expect, rec = [lo.copy(), hi.copy()], [], [], []
    inst = tries = 0
    while inst < n_inst and tries < 200000:
        tries += 1
        k = int(rng.integers(55, 100))
        qx = f32(float(org[0]) + k * float(cell) + float(rng.uniform(0.3e-3, 3e-3)))
        ux = f32(f32(qx - org[0]) / cell)
        if int(np.floor(ux)) != k:
            continue
        xlo = f32(f32(ux - np.floor(ux)) * cell)                 # nn_block: xlo = (ux - cx) * g.cell
        px = f32(float(org[0]) + k * float(cell))                # the largest float that the grid build puts into column k - 1
        for _ in range(8):
            if int(cell_of(px, org[0], cell)) >= k:
                px = np.nextafter(px, f32(-10))
        if int(cell_of(px, org[0], cell)) != k - 1:
            continue
        dx = f32(qx - px)
        dstar = f32(dx * dx)
        lhs = f32(xlo * xlo)
        cy, cz = 2 + 3 * (inst % 10), 2 + 3 * (inst // 10)       # every instance in (y, z) cells of its own, the query in the middle
        qy = f32(float(org[1]) + (cy + 0.5) * float(cell))
        qz = f32(float(org[2]) + (cz + 0.5) * float(cell))
        if int(cell_of(qy, org[1], cell)) != cy or int(cell_of(qz, org[2], cell)) != cz:
            continue
        y0 = f32(qy + f32(np.sqrt(dstar)))
        found = None
        for step in range(-6, 7):                                # a competitor (qx, y', qz): float32 distance just above the true neighbour's
            yy = y0
            for _ in range(abs(step)):
                yy = np.nextafter(yy, f32(10) if step > 0 else f32(-10))
            dy = f32(qy - yy)
            d = f32(f32(f32(0) + f32(dy * dy)) + f32(0))
            if d > dstar and lhs > f32(f32(d * f32(1.0001)) + f32(1e-12)) and int(cell_of(yy, org[1], cell)) == cy:
                found = (yy, d)
                break
        if found is None:
            continue
        yy, d = found
        tgt.append(np.array([px, qy, qz], np.float32))
        expect.append(len(tgt) - 1)
        tgt.append(np.array([qx, yy, qz], np.float32))
        src.append(np.array([qx, qy, qz], np.float32))
        rec.append((float(lhs), float(d), float(dstar)))
        inst += 1
    return np.array(tgt, np.float32), np.array(src, np.float32), np.array(expect), np.array(rec), (org, cell, dim)
