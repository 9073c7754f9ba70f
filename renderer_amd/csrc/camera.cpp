// Camera: the uniform buffer and push constant the reference fills every frame
// (updateUniformBuffer, src/invert.cpp:621-642), and the reduction of a camera to
// the per-frame constants the render kernels consume.
//
// GLM is not a dependency here; the three calls the reference makes are written
// out from GLM's documented formulas (right-handed, GLM_FORCE_DEPTH_ZERO_TO_ONE,
// src/invert.cpp:8).  sin/cos/tan are evaluated in double and rounded once, so the
// matrices do not depend on the host libm's float routines.
#include <cmath>
#include <cstring>

#include "internal.h"

namespace inv {
namespace {

struct Vec3 {
    float x, y, z;
};

struct Mat4 { // column-major: c[col][row], GLM's layout
    float c[4][4];
    static Mat4 identity() {
        Mat4 m;
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) m.c[i][j] = (i == j) ? 1.0f : 0.0f;
        return m;
    }
    void store(float* out) const { std::memcpy(out, c, sizeof c); }
    static Mat4 load(const float* in) {
        Mat4 m;
        std::memcpy(m.c, in, sizeof m.c);
        return m;
    }
};

float sum3(float a, float b, float c) { return a + b + c; } // glm::dot: tmp.x + tmp.y + tmp.z
float dot3(Vec3 a, Vec3 b) { return sum3(a.x * b.x, a.y * b.y, a.z * b.z); }
Vec3 cross3(Vec3 a, Vec3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
Vec3 unit3(Vec3 v) { // glm::normalize = v * inversesqrt(dot(v, v))
    float s = 1.0f / std::sqrt(dot3(v, v));
    return {v.x * s, v.y * s, v.z * s};
}
float to_radians(float degrees) { return degrees * 0.01745329251994329576923690768489f; }

Mat4 look_at_rh(Vec3 eye, Vec3 center, Vec3 up) {
    Vec3 fwd = unit3({center.x - eye.x, center.y - eye.y, center.z - eye.z});
    Vec3 side = unit3(cross3(fwd, up));
    Vec3 upv = cross3(side, fwd);
    Mat4 m = Mat4::identity();
    m.c[0][0] = side.x, m.c[1][0] = side.y, m.c[2][0] = side.z;
    m.c[0][1] = upv.x, m.c[1][1] = upv.y, m.c[2][1] = upv.z;
    m.c[0][2] = -fwd.x, m.c[1][2] = -fwd.y, m.c[2][2] = -fwd.z;
    m.c[3][0] = -dot3(side, eye);
    m.c[3][1] = -dot3(upv, eye);
    m.c[3][2] = dot3(fwd, eye);
    return m;
}

Mat4 rotate(const Mat4& m, float angle, Vec3 axis_in) {
    const float co = (float)std::cos((double)angle);
    const float si = (float)std::sin((double)angle);
    Vec3 a = unit3(axis_in);
    Vec3 t = {(1.0f - co) * a.x, (1.0f - co) * a.y, (1.0f - co) * a.z};
    const float rot[3][3] = {
        {co + t.x * a.x, t.x * a.y + si * a.z, t.x * a.z - si * a.y},
        {t.y * a.x - si * a.z, co + t.y * a.y, t.y * a.z + si * a.x},
        {t.z * a.x + si * a.y, t.z * a.y - si * a.x, co + t.z * a.z},
    };
    Mat4 out;
    for (int col = 0; col < 3; col++)
        for (int row = 0; row < 4; row++)
            out.c[col][row] = m.c[0][row] * rot[col][0] + m.c[1][row] * rot[col][1] + m.c[2][row] * rot[col][2];
    for (int row = 0; row < 4; row++) out.c[3][row] = m.c[3][row];
    return out;
}

Mat4 perspective_rh_zo(float fovy, float aspect, float z_near, float z_far) {
    const float th = (float)std::tan((double)(fovy / 2.0f));
    Mat4 m;
    std::memset(m.c, 0, sizeof m.c);
    m.c[0][0] = 1.0f / (aspect * th);
    m.c[1][1] = 1.0f / th;
    m.c[2][2] = z_far / (z_near - z_far);
    m.c[2][3] = -1.0f;
    m.c[3][2] = -(z_far * z_near) / (z_far - z_near);
    return m;
}

// 4x4 inverse by Gauss-Jordan elimination with partial pivoting, in double.
bool invert(const Mat4& m, double inv[4][4] /* [row][col] */) {
    double w[4][8];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) {
            w[r][c] = (double)m.c[c][r];
            w[r][4 + c] = r == c ? 1.0 : 0.0;
        }
    for (int col = 0; col < 4; col++) {
        int piv = col;
        for (int r = col + 1; r < 4; r++)
            if (std::fabs(w[r][col]) > std::fabs(w[piv][col])) piv = r;
        if (w[piv][col] == 0.0) return false;
        if (piv != col)
            for (int c = 0; c < 8; c++) {
                double tmp = w[piv][c];
                w[piv][c] = w[col][c];
                w[col][c] = tmp;
            }
        const double d = w[col][col];
        for (int c = 0; c < 8; c++) w[col][c] = w[col][c] / d;
        for (int r = 0; r < 4; r++) {
            if (r == col) continue;
            const double f = w[r][col];
            if (f == 0.0) continue;
            for (int c = 0; c < 8; c++) w[r][c] = w[r][c] - f * w[col][c];
        }
    }
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) inv[r][c] = w[r][4 + c];
    return true;
}

} // namespace

void reference_camera(uint32_t width, uint32_t height, inv_camera* cam) {
    const Vec3 position = {0.0f, 0.0f, 1.0f}; // src/invert.cpp:630
    const Vec3 focus = {0.0f, 0.0f, 0.0f};    // src/invert.cpp:631
    const Vec3 up = {0.0f, 1.0f, 0.0f};       // src/invert.cpp:632
    cam->cam_angle[0] = focus.x - position.x; // src/invert.cpp:633
    cam->cam_angle[1] = focus.y - position.y;
    cam->cam_angle[2] = focus.z - position.z;
    Mat4::identity().store(cam->model);                                                    // src/invert.cpp:636
    Mat4 view = look_at_rh(position, focus, up);                                           // src/invert.cpp:637
    view = rotate(view, to_radians(180.0f), {0.0f, 0.0f, 1.0f});                           // src/invert.cpp:638
    view.store(cam->view);
    perspective_rh_zo(to_radians(45.0f), (float)width / (float)height, 0.1f, 10.0f).store(cam->proj); // :639
}

// The reference measures `elapsed` every frame and never uses it (src/invert.cpp:623-626).  This is the use the
// structure invites: the model matrix turns about the up axis at 90 degrees per second, everything else as in
// reference_camera.  elapsed = 0 gives exactly the reference camera (rotate by 0 is the identity matrix).
void turntable_camera(uint32_t width, uint32_t height, float elapsed_seconds, inv_camera* cam) {
    reference_camera(width, height, cam);
    const float angle = elapsed_seconds * to_radians(90.0f);
    rotate(Mat4::identity(), angle, {0.0f, 1.0f, 0.0f}).store(cam->model);
}

int frame_from_camera(const inv_camera& cam, float max_abs_coord, FrameK* f) {
    const Mat4 P = Mat4::load(cam.proj);
    // Rays are generated from the diagonal terms only, so the projection must have GLM's perspective shape.
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++) {
            bool used = (c == r && c < 3) || (c == 2 && r == 3) || (c == 3 && r == 2);
            if (!used && P.c[c][r] != 0.0f)
                return fail(INV_ERR_UNSUPPORTED, "camera: proj must be a GLM-style perspective matrix (off-diagonal term set)");
        }
    if (P.c[2][3] != -1.0f || !(P.c[0][0] > 0.0f) || !(P.c[1][1] > 0.0f))
        return fail(INV_ERR_UNSUPPORTED, "camera: proj must be a GLM-style perspective matrix (w_clip = -z_view)");
    const Mat4 V = Mat4::load(cam.view), M = Mat4::load(cam.model);
    Mat4 VM; // glm operator*: column c of the product = V * column c of M, summed left to right
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++)
            VM.c[c][r] = ((V.c[0][r] * M.c[c][0] + V.c[1][r] * M.c[c][1]) + V.c[2][r] * M.c[c][2]) + V.c[3][r] * M.c[c][3];
    double inv[4][4];
    if (!invert(VM, inv)) return fail(INV_ERR_ARG, "camera: view*model is singular");
    f->ox = (float)inv[0][3], f->oy = (float)inv[1][3], f->oz = (float)inv[2][3];
    f->d0x = (float)inv[0][0], f->d0y = (float)inv[1][0], f->d0z = (float)inv[2][0];
    f->d1x = (float)inv[0][1], f->d1y = (float)inv[1][1], f->d1z = (float)inv[2][1];
    f->d2x = (float)inv[0][2], f->d2y = (float)inv[1][2], f->d2z = (float)inv[2][2];
    f->p00 = P.c[0][0], f->p11 = P.c[1][1], f->p22 = P.c[2][2], f->p32 = P.c[3][2];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 4; c++) f->mv[r * 4 + c] = VM.c[c][r];
    f->inv_w2 = 2.0f / f->wf, f->inv_h2 = 2.0f / f->hf;
    f->inv_p00 = 1.0f / f->p00, f->inv_p11 = 1.0f / f->p11;
    float reach = std::fmax(std::fabs(f->ox), std::fmax(std::fabs(f->oy), std::fabs(f->oz)));
    reach = std::fmax(reach, max_abs_coord);
    f->pad = reach * 0x1p-16f;
    // Light direction for the Lambert term: -cam_angle, carried from world to object space with the
    // transpose of the model 3x3 (exact for rigid model matrices; the reference uses the identity).
    const float cx = cam.cam_angle[0], cy = cam.cam_angle[1], cz = cam.cam_angle[2];
    const float lx = (M.c[0][0] * cx + M.c[0][1] * cy) + M.c[0][2] * cz;
    const float ly = (M.c[1][0] * cx + M.c[1][1] * cy) + M.c[1][2] * cz;
    const float lz = (M.c[2][0] * cx + M.c[2][1] * cy) + M.c[2][2] * cz;
    const float ll = (lx * lx + ly * ly) + lz * lz;
    if (ll > 0.0f) {
        const float s = 1.0f / std::sqrt(ll);
        f->lx = -(lx * s), f->ly = -(ly * s), f->lz = -(lz * s);
    } else {
        f->lx = f->ly = f->lz = 0.0f;
    }
    return INV_OK;
}

// Pixel rectangle outside which no ray can hit the scene: the scene bounds grown by four guard pads, projected in
// double precision, plus a two-pixel margin.  A box corner at or behind the eye plane gives the whole frame.
void hit_rect(const FrameK& f, const float bbox[6], uint32_t rect[4]) {
    rect[0] = 0, rect[1] = 0, rect[2] = f.w, rect[3] = f.h;
    const double g = 4.0 * (double)f.pad;
    double xmin = 1e300, xmax = -1e300, ymin = 1e300, ymax = -1e300;
    int behind = 0; // corners at or behind the eye plane
    for (int corner = 0; corner < 8; corner++) {
        const double p[3] = {(corner & 1) ? (double)bbox[3] + g : (double)bbox[0] - g,
                             (corner & 2) ? (double)bbox[4] + g : (double)bbox[1] - g,
                             (corner & 4) ? (double)bbox[5] + g : (double)bbox[2] - g};
        double v[3];
        for (int r = 0; r < 3; r++)
            v[r] = (double)f.mv[r * 4 + 0] * p[0] + (double)f.mv[r * 4 + 1] * p[1] + (double)f.mv[r * 4 + 2] * p[2] +
                   (double)f.mv[r * 4 + 3];
        const double wc = -v[2];
        if (!(wc > 1e-9)) {
            behind++;
            continue;
        }
        const double sx = ((double)f.p00 * v[0] / wc + 1.0) * 0.5 * (double)f.w;
        const double sy = ((double)f.p11 * v[1] / wc + 1.0) * 0.5 * (double)f.h;
        xmin = std::fmin(xmin, sx), xmax = std::fmax(xmax, sx);
        ymin = std::fmin(ymin, sy), ymax = std::fmax(ymax, sy);
    }
    if (behind == 8) { // the whole box is behind the eye: a hit needs w_clip > 0
        rect[2] = rect[3] = 0;
        return;
    }
    if (behind != 0 || !(xmin <= xmax) || !(ymin <= ymax)) return; // box straddles the eye plane: no bound
    auto clampu = [](double v, double hi) { return (uint32_t)std::fmin(std::fmax(v, 0.0), hi); };
    rect[0] = clampu(std::floor(xmin) - 2.0, (double)f.w);
    rect[1] = clampu(std::floor(ymin) - 2.0, (double)f.h);
    rect[2] = clampu(std::ceil(xmax) + 2.0, (double)f.w);
    rect[3] = clampu(std::ceil(ymax) + 2.0, (double)f.h);
}

} // namespace inv
