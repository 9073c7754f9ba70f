// Host-side extras around the hot path: a tolerant OBJ reader for meshes that are not in the reference's fixed
// format, and writers that make the attachments inspectable (the reference presents its colour image to a window,
// src/invert.cpp:674-683, and never stores anything).
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "internal.h"

namespace inv {

// General OBJ: `v x y z [w]` with free-form numbers (strtof: correctly rounded, so coordinates can differ from the
// fixed-format loader's by a few ulp on the same text), `f` with 3 or more corners of the forms i, i/t, i//n, i/t/n,
// negative (relative) indices, CRLF or LF, a missing final newline; every other line kind is ignored.  Polygons are
// fanned into triangles (corner 0, k, k+1).
int parse_obj_general(const char* text, uint64_t size, std::vector<float>& xyzw, std::vector<uint32_t>& idx) {
    xyzw.clear();
    idx.clear();
    if (!text && size) return fail(INV_ERR_ARG, "parse_obj_general: null text");
    std::string buf(text, text + size); // NUL-terminated copy for strtof/strtol
    const char* p = buf.c_str();
    const char* end = p + buf.size();
    uint64_t line = 0;
    char msg[160];
    std::vector<long> corners;
    while (p < end) {
        line++;
        const char* eol = (const char*)std::memchr(p, '\n', (size_t)(end - p));
        const char* next = eol ? eol + 1 : end;
        const char* stop = eol ? eol : end;
        while (p < stop && (*p == ' ' || *p == '\t')) p++;
        if (p + 1 < stop && p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
            float v[3];
            const char* q = p + 1;
            for (int i = 0; i < 3; i++) {
                char* after = nullptr;
                errno = 0;
                v[i] = std::strtof(q, &after);
                if (after == q || after > stop) {
                    std::snprintf(msg, sizeof msg, "parse_obj_general: line %llu: expected three coordinates", (unsigned long long)line);
                    return fail(INV_ERR_FORMAT, msg);
                }
                q = after;
            }
            xyzw.push_back(v[0]), xyzw.push_back(v[1]), xyzw.push_back(v[2]), xyzw.push_back(0.0f);
        } else if (p + 1 < stop && p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
            corners.clear();
            const char* q = p + 1;
            for (;;) {
                while (q < stop && (*q == ' ' || *q == '\t' || *q == '\r')) q++;
                if (q >= stop) break;
                char* after = nullptr;
                long i = std::strtol(q, &after, 10);
                if (after == q || after > stop || i == 0) {
                    std::snprintf(msg, sizeof msg, "parse_obj_general: line %llu: bad face corner", (unsigned long long)line);
                    return fail(INV_ERR_FORMAT, msg);
                }
                const long nv = (long)(xyzw.size() / 4);
                corners.push_back(i > 0 ? i - 1 : nv + i); // negative indices count back from the last vertex read
                q = after;
                while (q < stop && *q != ' ' && *q != '\t' && *q != '\r') q++; // skip /t/n
            }
            if (corners.size() < 3) {
                std::snprintf(msg, sizeof msg, "parse_obj_general: line %llu: a face needs three corners", (unsigned long long)line);
                return fail(INV_ERR_FORMAT, msg);
            }
            for (size_t k = 1; k + 1 < corners.size(); k++) {
                const long tri[3] = {corners[0], corners[k], corners[k + 1]};
                for (long c : tri) {
                    if (c < 0 || c > 0x7FFFFFFEL) {
                        std::snprintf(msg, sizeof msg, "parse_obj_general: line %llu: vertex index out of range", (unsigned long long)line);
                        return fail(INV_ERR_FORMAT, msg);
                    }
                    idx.push_back((uint32_t)c);
                }
            }
        }
        p = next;
    }
    const uint32_t nv = (uint32_t)(xyzw.size() / 4);
    for (size_t i = 0; i < idx.size(); i++)
        if (idx[i] >= nv) {
            std::snprintf(msg, sizeof msg, "parse_obj_general: face %zu refers to vertex %u but the file has %u", i / 3, idx[i] + 1u, nv);
            return fail(INV_ERR_FORMAT, msg);
        }
    return INV_OK;
}

namespace {
struct File {
    FILE* fp;
    explicit File(const char* path) : fp(std::fopen(path, "wb")) {}
    ~File() {
        if (fp) std::fclose(fp);
    }
};
} // namespace

// Binary PPM (P6) of a colour attachment: 4 bytes per pixel in, R,G,B out.
int write_ppm(const char* path, const uint8_t* color, uint32_t w, uint32_t h, bool rgba_order) {
    File f(path);
    if (!f.fp) return fail(INV_ERR_IO, std::string("inv_write_ppm: cannot open ") + path);
    std::fprintf(f.fp, "P6\n%u %u\n255\n", w, h);
    std::vector<uint8_t> row((size_t)w * 3);
    for (uint32_t y = 0; y < h; y++) {
        const uint8_t* src = color + (size_t)y * w * 4;
        for (uint32_t x = 0; x < w; x++) {
            row[3 * x + 0] = src[4 * x + (rgba_order ? 0 : 2)]; // B,G,R,A memory order unless INV_RGBA (src/invert.cpp:519)
            row[3 * x + 1] = src[4 * x + 1];
            row[3 * x + 2] = src[4 * x + (rgba_order ? 2 : 0)];
        }
        if (std::fwrite(row.data(), 1, row.size(), f.fp) != row.size()) return fail(INV_ERR_IO, "inv_write_ppm: short write");
    }
    return INV_OK;
}

// PFM (Pf, little-endian, bottom row first) of a depth attachment.
int write_pfm(const char* path, const float* depth, uint32_t w, uint32_t h) {
    File f(path);
    if (!f.fp) return fail(INV_ERR_IO, std::string("inv_write_pfm: cannot open ") + path);
    std::fprintf(f.fp, "Pf\n%u %u\n-1.0\n", w, h);
    for (uint32_t y = h; y-- > 0;)
        if (std::fwrite(depth + (size_t)y * w, 4, w, f.fp) != w) return fail(INV_ERR_IO, "inv_write_pfm: short write");
    return INV_OK;
}

// Raw little-endian uint32 primitive ids, preceded by an 16-byte header "PID1", width, height, 0.
int write_primid(const char* path, const uint32_t* primid, uint32_t w, uint32_t h) {
    File f(path);
    if (!f.fp) return fail(INV_ERR_IO, std::string("inv_write_primid: cannot open ") + path);
    const uint32_t header[4] = {0x31444950u /* "PID1" */, w, h, 0u};
    if (std::fwrite(header, 4, 4, f.fp) != 4 || std::fwrite(primid, 4, (size_t)w * h, f.fp) != (size_t)w * h)
        return fail(INV_ERR_IO, "inv_write_primid: short write");
    return INV_OK;
}

} // namespace inv
