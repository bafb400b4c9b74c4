"""Point-cloud augmentation helpers with the reference's interface and random-number consumption
(`minsu3d/util/transform.py`: jitter :6-13, flip :16-26, roty :28-36, roty_batch :38-52, rotz :54-62, elastic :65-84,
crop :87-98).  The elastic distortion is restated as explicit zero-padded 3-tap box blurs + trilinear sampling of the
noise grid (what scipy.ndimage.convolve / RegularGridInterpolator compute for these arguments); the same two steps run
on the GPU in csrc/augment.hip."""
import numpy as np


def jitter(intensity=0.1):
    return np.eye(3) + np.random.randn(3, 3) * intensity


def flip(axis=0, random=False):
    m = np.eye(3)
    m[axis][axis] *= -1 if not random else np.random.randint(0, 2) * 2 - 1
    return m


def roty(t):
    c, s = np.cos(t), np.sin(t)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def roty_batch(t):
    out = np.zeros(tuple(t.shape) + (3, 3))
    c, s = np.cos(t), np.sin(t)
    out[..., 0, 0], out[..., 0, 2], out[..., 1, 1], out[..., 2, 0], out[..., 2, 2] = c, s, 1, -s, c
    return out


def rotz(t):
    c, s = np.cos(t), np.sin(t)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])


def elastic_noise(x, gran):
    """the three float32 noise grids the reference draws for `elastic(x, gran, .)` (same RNG calls, same shapes)"""
    bb = (np.abs(x).max(0) // gran + 3).astype(np.int32)
    return [np.random.randn(bb[0], bb[1], bb[2]).astype(np.float32) for _ in range(3)]


def blur_noise(noise):
    """two rounds of (1/3, 1/3, 1/3) box blurs along x, y, z with zero padding, float32 storage after every pass"""
    w = np.float64(np.float32(1.0) / np.float32(3.0))
    out = noise
    for _ in range(2):
        for axis in range(3):
            pad = [(0, 0)] * 3
            pad[axis] = (1, 1)
            p = np.pad(out.astype(np.float64), pad)
            sl = [slice(None)] * 3
            acc = 0.0
            for k in (2, 1, 0):                      # convolution order: kernel reversed (symmetric here)
                sl[axis] = slice(k, k + out.shape[axis])
                acc = acc + w * p[tuple(sl)]
            out = acc.astype(np.float32)
    return out


def trilinear(grid, gran, x):
    """sample `grid` (node i of axis a at (i - (n_a-1)/2) * 2*gran... i.e. linspace(-(n-1)g, (n-1)g, n)) at points x;
    0 outside the grid"""
    x = np.asarray(x, np.float64)
    n = np.array(grid.shape)
    lo = -(n - 1) * gran
    step = 2.0 * gran                                 # linspace(-(n-1)g, (n-1)g, n) has spacing 2g
    t = (x - lo) / step
    inside = np.all((t >= 0) & (t <= n - 1), axis=1)
    i0 = np.clip(np.floor(t).astype(np.int64), 0, n - 2)
    f = t - i0
    out = np.zeros(x.shape[0])
    g = grid.astype(np.float64)
    for dx in (0, 1):
        for dy in (0, 1):
            for dz in (0, 1):
                wgt = (f[:, 0] if dx else 1 - f[:, 0]) * (f[:, 1] if dy else 1 - f[:, 1]) * (f[:, 2] if dz else 1 - f[:, 2])
                out += wgt * g[i0[:, 0] + dx, i0[:, 1] + dy, i0[:, 2] + dz]
    return np.where(inside, out, 0.0)


def elastic(x, gran, mag):
    """x + mag * smooth random displacement field (reference :65-84)"""
    noise = [blur_noise(n) for n in elastic_noise(x, gran)]
    return x + np.hstack([trilinear(n, gran, x)[:, None] for n in noise]) * mag


def crop(pc, max_num_point, scale):
    """shrink a random window until at most max_num_point points fall inside (reference :87-98)"""
    pc_offset = pc.copy()
    valid = pc_offset.min(1) >= 0
    window = np.full(3, scale, dtype=np.uint16)
    extent = pc.max(0) - pc.min(0)
    while np.count_nonzero(valid) > max_num_point:
        offset = np.clip(window - extent + 0.001, None, 0) * np.random.rand(3)
        pc_offset = pc + offset
        valid = np.logical_and(pc_offset.min(1) >= 0, np.all(pc_offset < window, axis=1))
        window[:2] -= 32
    return pc_offset, valid
