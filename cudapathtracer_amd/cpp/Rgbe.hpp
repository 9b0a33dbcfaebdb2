// Rgbe.hpp -- Radiance RGBE (.hdr) reader/writer for environment maps and HDR output.
// The format on the input side of the path (reference: Common/rgbe.c, used at DefaultScene/tracer.cpp:130-138):
// text header ("#?..." line, FORMAT=32-bit_rle_rgbe, blank line, "-Y H +X W"), then flat RGBE pixels or new-style
// run-length scanlines (2,2,hi,lo; per channel: count>128 = run of count-128, else `count` literals).
// rgbe -> float follows rgbe.c:93-106 (mantissa * 2^(e-136), no +0.5), float -> rgbe follows rgbe.c:70-88.
#pragma once
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace CUDATracer {
namespace Rgbe {

inline void ToFloat(const unsigned char rgbe[4], float* rgb) {
    if (rgbe[3]) {
        float f = (float)std::ldexp(1.0, (int)rgbe[3] - (128 + 8));
        rgb[0] = rgbe[0] * f; rgb[1] = rgbe[1] * f; rgb[2] = rgbe[2] * f;
    } else rgb[0] = rgb[1] = rgb[2] = 0.f;
}
inline void FromFloat(const float* rgb, unsigned char rgbe[4]) {
    float v = rgb[0]; if (rgb[1] > v) v = rgb[1]; if (rgb[2] > v) v = rgb[2];
    if (v < 1e-32 || v != v) { rgbe[0] = rgbe[1] = rgbe[2] = rgbe[3] = 0; return; }      // rgbe.c:78 compares in double; NaN (undefined there) -> black
    int e; v = (float)(std::frexp(v, &e) * 256.0 / v);
    rgbe[0] = (unsigned char)(rgb[0] * v); rgbe[1] = (unsigned char)(rgb[1] * v); rgbe[2] = (unsigned char)(rgb[2] * v);
    rgbe[3] = (unsigned char)(e + 128);
}

// Decodes a whole .hdr image held in memory into float RGB rows (top row first).  Returns false on malformed input.
// The set of byte strings accepted -- and every pixel of an accepted one -- is that of the reference's reader as
// DefaultScene/tracer.cpp:130-134 calls it, RGBE_ReadHeader(f, &w, &h, NULL) + RGBE_ReadPixels_RLE (Common/rgbe.c:134-190,
// 327-412); tests/test_rgbe.py holds this function to the reference's own rgbe.c (compiled unmodified by the test
// infrastructure) on a corpus of well-formed, odd and damaged files.  What follows from that:
//  * header lines are what fgets(buf, 128, fp) yields: at most 127 bytes, cut after a newline, compared as C strings;
//  * the "#?" magic is optional (rgbe.c:149-152: the check is commented out), and the first line goes through the same
//    loop as every other line;
//  * an empty line before FORMAT= is an error; the line after "FORMAT=32-bit_rle_rgbe\n" must be exactly "\n"
//    (rgbe.c:180-184), so GAMMA= / EXPOSURE= / comment lines are only accepted BEFORE the FORMAT line;
//  * the size line is parsed by sscanf("-Y %d +X %d") (rgbe.c:187).
// Deviations, all where the reference reads or allocates out of bounds: a size that is not positive is refused (the
// reference "accepts" it, reads nothing and its caller builds a vector of 3*w*h floats), and a size whose pixel data
// cannot fit in the bytes that follow is refused before anything is allocated (the reference fails at the first short
// read, after its caller has allocated).
inline bool Decode(const unsigned char* p, size_t n, std::vector<float>& rgb, int& width, int& height) {
    size_t pos = 0;
    auto fgets128 = [&](std::string& out) {         // fgets(buf, 128, fp) followed by C-string use of buf
        out.clear();
        if (pos >= n) return false;
        while (pos < n && out.size() < 127) { const char c = (char)p[pos++]; out.push_back(c); if (c == '\n') break; }
        const size_t z = out.find('\0');
        if (z != std::string::npos) out.resize(z);
        return true;
    };
    std::string l;
    if (!fgets128(l)) return false;
    for (;;) {
        if (l.empty() || l[0] == '\n') return false;                // "no FORMAT specifier found"
        if (l == "FORMAT=32-bit_rle_rgbe\n") break;
        if (!fgets128(l)) return false;
    }
    if (!fgets128(l) || l != "\n") return false;                     // "missing blank line after FORMAT specifier"
    if (!fgets128(l) || std::sscanf(l.c_str(), "-Y %d +X %d", &height, &width) < 2) return false;
    if (width <= 0 || height <= 0) return false;
    {   // the header is untrusted: no allocation before the pixel count is known to fit the bytes that follow.  Flat data
        // takes 4 bytes per pixel; a run-length scanline at least its 4-byte marker plus, per channel, 2 bytes per run of
        // at most 127 pixels.
        const unsigned long long W = (unsigned long long)width, H = (unsigned long long)height, left = n - pos;
        const unsigned long long flatBytes = W * H * 4ull, rleBytes = H * (4ull + 4ull * 2ull * ((W + 126ull) / 127ull));
        if ((flatBytes < rleBytes ? flatBytes : rleBytes) > left) return false;
    }
    rgb.assign((size_t)width * height * 3, 0.f);
    std::vector<unsigned char> scan((size_t)width * 4);
    float* out = rgb.data();
    auto flat = [&](size_t pixels) {
        if (pos + pixels * 4 > n) return false;
        for (size_t i = 0; i < pixels; i++, pos += 4, out += 3) ToFloat(p + pos, out);
        return true;
    };
    if (width < 8 || width > 0x7fff) return flat((size_t)width * height);
    for (int y = 0; y < height; y++) {
        if (pos + 4 > n) return false;
        const unsigned char* h = p + pos;
        if (h[0] != 2 || h[1] != 2 || (h[2] & 0x80)) return flat((size_t)width * (height - y));    // not run-length encoded
        if (((int)h[2] << 8 | h[3]) != width) return false;
        pos += 4;
        for (int c = 0; c < 4; c++) {
            unsigned char* q = &scan[(size_t)c * width]; unsigned char* end = q + width;
            while (q < end) {
                if (pos + 2 > n) return false;
                int count = p[pos++];
                if (count > 128) { count -= 128; if (count > end - q) return false; std::memset(q, p[pos++], count); q += count; }
                else { if (count == 0 || count > end - q || pos + count > n) return false; std::memcpy(q, p + pos, count); pos += count; q += count; }
            }
        }
        for (int x = 0; x < width; x++, out += 3) {
            unsigned char px[4] = { scan[x], scan[x + width], scan[x + 2 * (size_t)width], scan[x + 3 * (size_t)width] };
            ToFloat(px, out);
        }
    }
    return true;
}

// Encodes float RGB rows as a run-length .hdr image.
inline std::vector<unsigned char> Encode(const float* rgb, int width, int height) {
    std::string hdr = "#?RGBE\nFORMAT=32-bit_rle_rgbe\n\n-Y " + std::to_string(height) + " +X " + std::to_string(width) + "\n";
    std::vector<unsigned char> o(hdr.begin(), hdr.end());
    std::vector<unsigned char> scan((size_t)width * 4);
    const bool rle = width >= 8 && width <= 0x7fff;
    for (int y = 0; y < height; y++) {
        for (int x = 0; x < width; x++) {
            unsigned char px[4]; FromFloat(rgb + ((size_t)y * width + x) * 3, px);
            if (!rle) { o.insert(o.end(), px, px + 4); continue; }
            for (int c = 0; c < 4; c++) scan[(size_t)c * width + x] = px[c];
        }
        if (!rle) continue;
        const unsigned char h[4] = { 2, 2, (unsigned char)(width >> 8), (unsigned char)(width & 0xff) };
        o.insert(o.end(), h, h + 4);
        for (int c = 0; c < 4; c++) {
            const unsigned char* d = &scan[(size_t)c * width];
            int cur = 0;
            while (cur < width) {
                int run = 1;                                   // length of the run starting at cur
                while (cur + run < width && run < 127 && d[cur + run] == d[cur]) run++;
                if (run >= 4) { o.push_back((unsigned char)(128 + run)); o.push_back(d[cur]); cur += run; continue; }
                int lit = cur, len = 0;                         // literal block up to the next run of >= 4
                while (lit + len < width && len < 128) {
                    int r = 1; while (lit + len + r < width && r < 4 && d[lit + len + r] == d[lit + len]) r++;
                    if (r >= 4) break;
                    len++;
                }
                o.push_back((unsigned char)len); o.insert(o.end(), d + lit, d + lit + len); cur += len;
            }
        }
    }
    return o;
}

inline bool ReadFile(const std::string& path, std::vector<float>& rgb, int& width, int& height) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::vector<unsigned char> buf; unsigned char tmp[65536]; size_t r;
    while ((r = std::fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + r);
    std::fclose(f);
    return Decode(buf.data(), buf.size(), rgb, width, height);
}
inline bool WriteFile(const std::string& path, const float* rgb, int width, int height) {
    std::vector<unsigned char> o = Encode(rgb, width, height);
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    bool ok = std::fwrite(o.data(), 1, o.size(), f) == o.size();
    std::fclose(f);
    return ok;
}

}  // namespace Rgbe
}  // namespace CUDATracer
