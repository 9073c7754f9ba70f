// Fixed-format OBJ loader: the host half of struct Obj (src/invert.cpp:25-209).
//
// The reference turns each coordinate `[-]D.DDDDDDDe[-]DDD` into a float32 with its
// own arithmetic (src/invert.cpp:139-152), which differs from a correctly rounded
// strtof by up to 3 ulp on 38 % of the bunny's coordinates.  That arithmetic
// defines the geometry, so it is reproduced exactly:
//
//     value = ((float)D + (float)DDDDDDD / 1e7f) * scale[exp] * (+-1.0f)
//
// where scale[] is 1.0f multiplied (or divided) by 10.0f `exp` times in float32.
// Unlike the reference this parser is bounds-checked and strict: anything that is
// not in the fixed format is INV_ERR_FORMAT instead of undefined behaviour.
#include <cstdio>
#include <cstring>

#include "internal.h"

namespace inv {
namespace {

struct Pow10Tables {
    float down[1000]; // 10^-e by repeated division   (src/invert.cpp:146-149)
    float up[1000];   // 10^+e by repeated multiplication (src/invert.cpp:142-145)
    Pow10Tables() {
        float d = 1.0f, u = 1.0f;
        for (int e = 0; e < 1000; e++) {
            down[e] = d;
            up[e] = u;
            d /= 10.0f;
            u *= 10.0f;
        }
    }
};

const Pow10Tables& pow10_tables() {
    static const Pow10Tables t;
    return t;
}

struct Cursor {
    const char* p;
    const char* end;
    bool digit() const { return p < end && *p >= '0' && *p <= '9'; }
    bool is(char c) const { return p < end && *p == c; }
};

bool read_coordinate(Cursor& c, float* out) {
    const Pow10Tables& tab = pow10_tables();
    bool negative = c.is('-');
    if (negative) c.p++;
    if (!c.digit()) return false;
    unsigned whole = (unsigned)(*c.p++ - '0'); // src/invert.cpp:105
    if (!c.is('.')) return false;
    c.p++;
    unsigned frac = 0; // seven digits read as one integer (src/invert.cpp:108-116)
    for (int j = 0; j < 7; j++) {
        if (!c.digit()) return false;
        frac = frac * 10u + (unsigned)(*c.p++ - '0');
    }
    if (!c.is('e')) return false;
    c.p++;
    bool negative_exp = c.is('-'); // a '+' is not accepted, as in the reference (src/invert.cpp:120-125)
    if (negative_exp) c.p++;
    unsigned expo = 0; // three digits (src/invert.cpp:127-135)
    for (int j = 0; j < 3; j++) {
        if (!c.digit()) return false;
        expo = expo * 10u + (unsigned)(*c.p++ - '0');
    }
    float scale = negative_exp ? tab.down[expo] : tab.up[expo];
    // float32, left to right (src/invert.cpp:152)
    *out = ((float)whole + ((float)frac / 10000000.0f)) * scale * (negative ? -1.0f : 1.0f);
    return true;
}

} // namespace

int parse_obj_text(const char* text, uint64_t size, std::vector<float>& xyzw, std::vector<uint32_t>& idx) {
    xyzw.clear();
    idx.clear();
    if (!text && size) return fail(INV_ERR_ARG, "parse_obj: null text");
    if (size == 0 || text[size - 1] != '\n')
        return fail(INV_ERR_FORMAT, "parse_obj: every line must end in LF (src/invert.cpp:69-74)");
    Cursor c{text, text + size};
    uint64_t line = 1;
    char msg[160];
    while (c.p < c.end) {
        char kind = *c.p; // the first byte of a line decides its kind (src/invert.cpp:59-66, 92, 161)
        if (kind == 'v') {
            c.p++;
            if (!c.is(' ')) goto bad;
            c.p++;
            float v[3];
            for (int i = 0; i < 3; i++) {
                if (!read_coordinate(c, &v[i])) goto bad;
                if (!c.is(i < 2 ? ' ' : '\n')) goto bad; // one separator byte (src/invert.cpp:137)
                c.p++;
            }
            xyzw.push_back(v[0]);
            xyzw.push_back(v[1]);
            xyzw.push_back(v[2]);
            xyzw.push_back(0.0f); // lane 3 pads the 16-byte stride; the reference leaves it uninitialised
            line++;
        } else if (kind == 'f') {
            c.p++;
            if (!c.is(' ')) goto bad;
            c.p++;
            for (int i = 0; i < 3; i++) {
                uint32_t value = 0;
                int digits = 0;
                while (c.digit()) { // at most 8 digits (src/invert.cpp:167-176)
                    if (++digits > 8) goto bad;
                    value = value * 10u + (uint32_t)(*c.p++ - '0');
                }
                if (digits == 0 || value == 0) goto bad;
                // any run of ' ', LF, CR follows an index (src/invert.cpp:178-180)
                while (c.p < c.end && (*c.p == ' ' || *c.p == '\n' || *c.p == '\r')) {
                    if (*c.p == '\n') line++;
                    c.p++;
                }
                idx.push_back(value - 1u); // 1-based -> 0-based (src/invert.cpp:192)
            }
        } else {
            while (*c.p != '\n') c.p++; // other lines are skipped (src/invert.cpp:199-207)
            c.p++;
            line++;
        }
    }
    {
        uint32_t nv = (uint32_t)(xyzw.size() / 4);
        for (size_t i = 0; i < idx.size(); i++)
            if (idx[i] >= nv) {
                std::snprintf(msg, sizeof msg, "parse_obj: face %zu refers to vertex %u but the file has %u", i / 3,
                              idx[i] + 1u, nv);
                return fail(INV_ERR_FORMAT, msg);
            }
    }
    return INV_OK;
bad:
    std::snprintf(msg, sizeof msg, "parse_obj: line %llu is not in the fixed format the reference loader reads",
                  (unsigned long long)line);
    xyzw.clear();
    idx.clear();
    return fail(INV_ERR_FORMAT, msg);
}

} // namespace inv
