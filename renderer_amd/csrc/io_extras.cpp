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

// PNG (8-bit RGBA, sRGB chunk: the attachment is B8G8R8A8_SRGB, src/invert.cpp:519) of a colour attachment.  The
// image data is a zlib stream of stored (uncompressed) deflate blocks: valid for every PNG reader, no compressor needed.
namespace {

struct Crc32 {
    uint32_t table[256];
    Crc32() {
        for (uint32_t n = 0; n < 256; n++) {
            uint32_t c = n;
            for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[n] = c;
        }
    }
    uint32_t update(uint32_t crc, const uint8_t* p, size_t n) const {
        for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 0xFFu] ^ (crc >> 8);
        return crc;
    }
};

void put_be32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back((uint8_t)(x >> 24)), v.push_back((uint8_t)(x >> 16)), v.push_back((uint8_t)(x >> 8)), v.push_back((uint8_t)x);
}

bool write_chunk(FILE* fp, const Crc32& crc, const char type[4], const std::vector<uint8_t>& data) {
    std::vector<uint8_t> head;
    put_be32(head, (uint32_t)data.size());
    head.insert(head.end(), type, type + 4);
    uint32_t c = crc.update(0xFFFFFFFFu, head.data() + 4, 4);
    c = crc.update(c, data.data(), data.size()) ^ 0xFFFFFFFFu;
    std::vector<uint8_t> tail;
    put_be32(tail, c);
    return std::fwrite(head.data(), 1, head.size(), fp) == head.size() &&
           (data.empty() || std::fwrite(data.data(), 1, data.size(), fp) == data.size()) &&
           std::fwrite(tail.data(), 1, 4, fp) == 4;
}

} // namespace

int write_png(const char* path, const uint8_t* color, uint32_t w, uint32_t h, bool rgba_order) {
    // one IDAT chunk holds the whole zlib stream: its length field is 32 bits
    const uint64_t raw = (uint64_t)h * ((uint64_t)w * 4 + 1);
    if (raw + raw / 65535 * 5 + 64 > 0x7FFFFFFFull) return fail(INV_ERR_ARG, "inv_write_png: image too large for one IDAT chunk");
    File f(path);
    if (!f.fp) return fail(INV_ERR_IO, std::string("inv_write_png: cannot open ") + path);
    static const Crc32 crc;
    const uint8_t signature[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n'};
    bool ok = std::fwrite(signature, 1, 8, f.fp) == 8;
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, w), put_be32(ihdr, h);
    const uint8_t fmt[5] = {8 /* bits */, 6 /* RGBA */, 0, 0, 0};
    ihdr.insert(ihdr.end(), fmt, fmt + 5);
    ok = ok && write_chunk(f.fp, crc, "IHDR", ihdr);
    ok = ok && write_chunk(f.fp, crc, "sRGB", std::vector<uint8_t>{0}); // rendering intent: perceptual
    // scanlines: filter byte 0, then R,G,B,A
    std::vector<uint8_t> lines((size_t)raw);
    uint32_t a = 1, b = 0; // Adler-32 of the uncompressed data
    for (uint32_t y = 0; y < h; y++) {
        uint8_t* dst = lines.data() + (size_t)y * ((size_t)w * 4 + 1);
        const uint8_t* src = color + (size_t)y * w * 4;
        *dst++ = 0;
        for (uint32_t x = 0; x < w; x++) {
            dst[4 * x + 0] = src[4 * x + (rgba_order ? 0 : 2)]; // B,G,R,A memory order unless INV_RGBA
            dst[4 * x + 1] = src[4 * x + 1];
            dst[4 * x + 2] = src[4 * x + (rgba_order ? 2 : 0)];
            dst[4 * x + 3] = src[4 * x + 3];
        }
    }
    for (size_t i = 0; i < lines.size();) { // sums reduced every 4096 bytes: no 32-bit overflow
        const size_t n = std::min<size_t>(4096, lines.size() - i);
        for (size_t k = 0; k < n; k++) a += lines[i + k], b += a;
        a %= 65521u, b %= 65521u, i += n;
    }
    std::vector<uint8_t> z;
    z.reserve(lines.size() + lines.size() / 65535 * 5 + 16);
    z.push_back(0x78), z.push_back(0x01); // zlib header: deflate, 32 KiB window, no preset dictionary
    for (size_t i = 0; i < lines.size() || i == 0;) {
        const size_t n = std::min<size_t>(65535, lines.size() - i);
        const bool last = i + n >= lines.size();
        z.push_back(last ? 1 : 0); // stored block
        z.push_back((uint8_t)n), z.push_back((uint8_t)(n >> 8)), z.push_back((uint8_t)~n), z.push_back((uint8_t)(~n >> 8));
        z.insert(z.end(), lines.begin() + (std::ptrdiff_t)i, lines.begin() + (std::ptrdiff_t)(i + n));
        i += n;
        if (last) break;
    }
    put_be32(z, (b << 16) | a);
    ok = ok && write_chunk(f.fp, crc, "IDAT", z);
    ok = ok && write_chunk(f.fp, crc, "IEND", std::vector<uint8_t>());
    if (!ok) return fail(INV_ERR_IO, "inv_write_png: short write");
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
