"""A double-precision rasteriser in numpy, written from the reference's pipeline description alone.

    clip = P (V (M p))                                      src/shader.slang:25
    colour attribute = (1, 1, 10 z_object)                  src/shader.slang:27
    triangle k = indices 3k .. 3k+2                         src/invert.cpp:1307-1310
    framebuffer x = (x_ndc + 1) w / 2, y = (y_ndc + 1) h / 2, depth = z_ndc    viewport (0, 0, w, h, 0..1), :779-791
    cull BACK, front face CLOCKWISE (y down)                src/invert.cpp:1331-1337
    one sample at the pixel centre                          src/invert.cpp:1375-1378
    depth test LESS against a 1.0 clear, write on pass      src/invert.cpp:793, 1380-1385 (earlier primitive wins ties)
    colour interpolated perspective-correctly, stored as B8G8R8A8_SRGB   src/shader.slang:31-36, src/invert.cpp:519

It shares no code with oracle/ or renderer_amd/: the tests hold the ray-casting oracle against it.
"""
import numpy as np


def srgb8(x):
    """linear -> 8-bit sRGB, float64 (IEC 61966-2-1 OETF, round to nearest)."""
    x = np.clip(np.asarray(x, np.float64), 0.0, 1.0)
    s = np.where(x <= 0.0031308, 12.92 * x, 1.055 * np.power(np.maximum(x, 1e-300), 1.0 / 2.4) - 0.055)
    return np.floor(255.0 * s + 0.5).astype(np.int64)


def rasterise(xyz, faces, model, view, proj, w, h, light=None):
    """xyz float [nv, 3], faces int [nf, 3], matrices column-major 16-vectors (as GLM stores them).

    Returns primid uint32 [h, w] (0xFFFFFFFF: nothing), depth float64 [h, w] (1.0: clear), blue float64 [h, w]
    (the interpolated 10 z, before clamping; with `light`, a unit vector, times max(0, n . light)), lam float64
    [h, w] (that Lambert factor, 1 without `light`), nfront."""
    M, V, P = [np.asarray(m, np.float64).reshape(4, 4).T for m in (model, view, proj)]  # column-major -> row-major
    p = np.concatenate([np.asarray(xyz, np.float64), np.ones((len(xyz), 1))], axis=1)
    clip = (P @ (V @ (M @ p.T))).T
    wc = clip[:, 3]
    fx = (clip[:, 0] / wc + 1.0) * w / 2.0
    fy = (clip[:, 1] / wc + 1.0) * h / 2.0
    fz = clip[:, 2] / wc
    attr = 10.0 * np.asarray(xyz, np.float64)[:, 2]
    primid = np.full((h, w), 0xFFFFFFFF, np.uint32)
    depth = np.ones((h, w), np.float64)
    blue = np.zeros((h, w), np.float64)
    lam = np.ones((h, w), np.float64)
    nfront = 0
    for k, (a, b, c) in enumerate(np.asarray(faces, np.int64)):
        x0, y0, x1, y1, x2, y2 = fx[a], fy[a], fx[b], fy[b], fx[c], fy[c]
        area = (x1 - x0) * (y2 - y0) - (x2 - x0) * (y1 - y0)
        if not area > 0.0:  # clockwise on the y-down screen is front-facing; the rest is culled
            continue
        nfront += 1
        px0, px1 = max(int(np.floor(min(x0, x1, x2) - 0.5)), 0), min(int(np.ceil(max(x0, x1, x2) - 0.5)), w - 1)
        py0, py1 = max(int(np.floor(min(y0, y1, y2) - 0.5)), 0), min(int(np.ceil(max(y0, y1, y2) - 0.5)), h - 1)
        if px0 > px1 or py0 > py1:
            continue
        X, Y = np.meshgrid(np.arange(px0, px1 + 1) + 0.5, np.arange(py0, py1 + 1) + 0.5)
        l0 = ((x1 - X) * (y2 - Y) - (x2 - X) * (y1 - Y)) / area
        l1 = ((x2 - X) * (y0 - Y) - (x0 - X) * (y2 - Y)) / area
        l2 = 1.0 - l0 - l1
        inside = (l0 >= 0.0) & (l1 >= 0.0) & (l2 >= 0.0)
        if not inside.any():
            continue
        z = l0 * fz[a] + l1 * fz[b] + l2 * fz[c]  # linear in screen space
        sub_d = depth[py0:py1 + 1, px0:px1 + 1]
        win = inside & (z < sub_d) & (z >= 0.0)
        if not win.any():
            continue
        iw = l0 / wc[a] + l1 / wc[b] + l2 / wc[c]
        col = (l0 * attr[a] / wc[a] + l1 * attr[b] / wc[b] + l2 * attr[c] / wc[c]) / iw  # perspective-correct
        L = 1.0
        if light is not None:
            n = np.cross(np.asarray(xyz[b], np.float64) - xyz[a], np.asarray(xyz[c], np.float64) - xyz[a])
            nn = np.linalg.norm(n)
            L = max(0.0, float(n @ np.asarray(light, np.float64)) / nn) if nn > 0 else 0.0
        sub_d[win] = z[win]
        primid[py0:py1 + 1, px0:px1 + 1][win] = k
        blue[py0:py1 + 1, px0:px1 + 1][win] = col[win]
        lam[py0:py1 + 1, px0:px1 + 1][win] = L
    return primid, depth, blue, lam, nfront
