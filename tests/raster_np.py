"""A double-precision rasteriser in numpy, written from the reference's pipeline description alone.

    clip = P (V (M p))                                      src/shader.slang:25
    colour attribute = (1, 1, 10 z_object)                  src/shader.slang:27
    triangle k = indices 3k .. 3k+2                         src/invert.cpp:1307-1310
    framebuffer x = (x_ndc + 1) w / 2, y = (y_ndc + 1) h / 2, depth = z_ndc    viewport (0, 0, w, h, 0..1), :779-791
    cull BACK, front face CLOCKWISE (y down)                src/invert.cpp:1331-1337
    one sample at the pixel centre                          src/invert.cpp:1375-1378
    depth test LESS against a 1.0 clear, write on pass      src/invert.cpp:793, 1380-1385 (earlier primitive wins ties)
    colour interpolated perspective-correctly, stored as B8G8R8A8_SRGB   src/shader.slang:31-36, src/invert.cpp:519

Two further rules of a Vulkan rasteriser can be switched on, to size how far an implementation's fixed-function
stage may move pixels away from the ray-cast definition (the reference leaves both to the driver):

    top_left=True       a sample exactly on an edge is covered only if the edge is a top or a left edge (Vulkan 1.4
                        specification, "Basic Polygon Rasterization"); without it edges are inclusive, as in the
                        Moller-Trumbore test
    subpixel_bits=n     vertex positions are snapped to a fixed-point grid of 2^-n pixel before set-up
                        (VkPhysicalDeviceLimits::subPixelPrecisionBits; the specification requires at least 4, current
                        desktop GPUs report 8); round half to even; edge functions are then exact integers

It shares no code with oracle/ or renderer_amd/: the tests hold the ray-casting oracle against it.
"""
import numpy as np


def srgb8(x):
    """linear -> 8-bit sRGB, float64 (IEC 61966-2-1 OETF, round to nearest)."""
    x = np.clip(np.asarray(x, np.float64), 0.0, 1.0)
    s = np.where(x <= 0.0031308, 12.92 * x, 1.055 * np.power(np.maximum(x, 1e-300), 1.0 / 2.4) - 0.055)
    return np.floor(255.0 * s + 0.5).astype(np.int64)


def rasterise(xyz, faces, model, view, proj, w, h, light=None, top_left=False, subpixel_bits=None):
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
    S = None
    if subpixel_bits is not None:
        S = 1 << int(subpixel_bits)
        ix, iy = np.rint(fx * S).astype(np.int64), np.rint(fy * S).astype(np.int64)  # fixed-point positions
        fx, fy = ix / float(S), iy / float(S)  # what set-up and interpolation see from here on
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
        if S is None:
            e12 = (x1 - X) * (y2 - Y) - (x2 - X) * (y1 - Y)  # edge v1->v2 (weight of v0), positive inside
            e20 = (x2 - X) * (y0 - Y) - (x0 - X) * (y2 - Y)  # edge v2->v0 (weight of v1)
            l0, l1 = e12 / area, e20 / area
            l2 = 1.0 - l0 - l1
            e01 = l2 * area
        else:  # exact: positions and sample points in units of 1 / S pixel (sample = pixel centre, S is even)
            XI = (np.arange(px0, px1 + 1, dtype=np.int64) * S + S // 2)[None, :]
            YI = (np.arange(py0, py1 + 1, dtype=np.int64) * S + S // 2)[:, None]
            e12 = (ix[b] - XI) * (iy[c] - YI) - (ix[c] - XI) * (iy[b] - YI)
            e20 = (ix[c] - XI) * (iy[a] - YI) - (ix[a] - XI) * (iy[c] - YI)
            e01 = (ix[a] - XI) * (iy[b] - YI) - (ix[b] - XI) * (iy[a] - YI)
            area_i = (ix[b] - ix[a]) * (iy[c] - iy[a]) - (ix[c] - ix[a]) * (iy[b] - iy[a])
            if area_i <= 0:  # the snapped triangle is degenerate or flipped: nothing is drawn
                continue
            l0, l1, l2 = e12 / float(area_i), e20 / float(area_i), e01 / float(area_i)
        if top_left:
            # clockwise on the y-down screen: an edge that runs upwards is a left edge, a horizontal one that runs
            # to the right a top edge; a sample exactly on any other edge belongs to the neighbour
            def owns(xa, ya, xb, yb):
                return (yb < ya) or (yb == ya and xb > xa)
            if S is None:
                own12, own20, own01 = owns(x1, y1, x2, y2), owns(x2, y2, x0, y0), owns(x0, y0, x1, y1)
            else:
                own12, own20, own01 = (owns(ix[b], iy[b], ix[c], iy[c]), owns(ix[c], iy[c], ix[a], iy[a]),
                                       owns(ix[a], iy[a], ix[b], iy[b]))
            inside = (((e12 > 0) | ((e12 == 0) & own12)) & ((e20 > 0) | ((e20 == 0) & own20))
                      & ((e01 > 0) | ((e01 == 0) & own01)))
        else:
            inside = (e12 >= 0) & (e20 >= 0) & (e01 >= 0)
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
