// texture.cc — texture files -> texels (reference: VerStarting/texture.cc:60-109).
//
// The reference decodes through SDL2_image; this build has no third-party
// decoder, so three self-contained formats are read directly: binary PPM (P6,
// maxval 255), uncompressed 24/32-bit BMP, uncompressed true-colour TGA.  The
// texel conversion is the reference's: colour = byte / 255.0
// (texture.cc:100-104), rows stored top to bottom, alpha dropped.
#include "texture.h"

#include <cstdint>
#include <cstdio>
#include <cstring>

namespace raytracer {
namespace {

bool ReadAll(const char* path, std::vector<uint8_t>* out) {
  FILE* f = fopen(path, "rb");
  if (!f) return false;
  uint8_t buf[65536];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, f)) > 0) out->insert(out->end(), buf, buf + n);
  fclose(f);
  return true;
}

bool SaneSize(long w, long h) { return w > 0 && h > 0 && w <= 30000 && h <= 30000; }

Texture* FromRgbRows(size_t w, size_t h, const std::vector<uint8_t>& rgb) {
  Texture* t = new Texture;
  t->width = w;
  t->height = h;
  t->colors.resize(w * h);
  for (size_t i = 0; i < w * h; i++) {
    t->colors[i] = {(double)rgb[i * 3] / 255.0, (double)rgb[i * 3 + 1] / 255.0,
                    (double)rgb[i * 3 + 2] / 255.0};
  }
  return t;
}

Texture* DecodePPM(const std::vector<uint8_t>& d) {
  size_t pos = 2;
  long vals[3];
  for (int k = 0; k < 3; k++) {
    for (;;) {  // whitespace and comments
      while (pos < d.size() && (d[pos] == ' ' || d[pos] == '\t' || d[pos] == '\n' || d[pos] == '\r')) pos++;
      if (pos < d.size() && d[pos] == '#') {
        while (pos < d.size() && d[pos] != '\n') pos++;
      } else {
        break;
      }
    }
    long v = 0;
    bool any = false;
    while (pos < d.size() && d[pos] >= '0' && d[pos] <= '9') {
      v = v * 10 + (d[pos++] - '0');
      any = true;
      if (v > 1000000) return nullptr;
    }
    if (!any) return nullptr;
    vals[k] = v;
  }
  pos++;  // the single whitespace byte after maxval
  if (!SaneSize(vals[0], vals[1]) || vals[2] != 255) return nullptr;
  const size_t w = (size_t)vals[0], h = (size_t)vals[1];
  if (d.size() < pos + w * h * 3) return nullptr;
  std::vector<uint8_t> rgb(d.begin() + pos, d.begin() + pos + w * h * 3);
  return FromRgbRows(w, h, rgb);
}

uint32_t Le32(const uint8_t* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
uint16_t Le16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

Texture* DecodeBMP(const std::vector<uint8_t>& d) {
  if (d.size() < 54) return nullptr;
  const uint32_t data_off = Le32(&d[10]);
  const uint32_t hdr = Le32(&d[14]);
  if (hdr < 40 || (uint64_t)data_off < 14ull + hdr) return nullptr;  // pixels may not overlap the headers
  const int32_t w = (int32_t)Le32(&d[18]);
  int32_t h = (int32_t)Le32(&d[22]);
  const uint16_t bpp = Le16(&d[28]);
  const uint32_t compression = Le32(&d[30]);
  if (h == INT32_MIN) return nullptr;  // -h would overflow
  const bool top_down = h < 0;
  if (top_down) h = -h;
  if (!SaneSize(w, h) || (bpp != 24 && bpp != 32) || (compression != 0 && compression != 3)) return nullptr;
  if (compression == 3) {
    // BI_BITFIELDS: only the standard layout (stored B, G, R, A) is read below.
    // The masks follow the 40-byte header (or sit inside a V4/V5 header).
    if (bpp != 32 || d.size() < 54 + 12) return nullptr;
    if (Le32(&d[54]) != 0x00ff0000u || Le32(&d[58]) != 0x0000ff00u || Le32(&d[62]) != 0x000000ffu) return nullptr;
    if (hdr == 40 && data_off < 54 + 12) return nullptr;
  }
  const size_t bytes_pp = bpp / 8;
  const size_t stride = ((size_t)w * bytes_pp + 3) & ~(size_t)3;
  if (d.size() < (size_t)data_off + stride * (size_t)h) return nullptr;
  std::vector<uint8_t> rgb((size_t)w * h * 3);
  for (int32_t y = 0; y < h; y++) {
    const uint8_t* row = &d[data_off + stride * (size_t)(top_down ? y : h - 1 - y)];
    for (int32_t x = 0; x < w; x++) {
      const uint8_t* px = row + (size_t)x * bytes_pp;  // stored B, G, R[, A]
      uint8_t* o = &rgb[((size_t)y * w + x) * 3];
      o[0] = px[2]; o[1] = px[1]; o[2] = px[0];
    }
  }
  return FromRgbRows((size_t)w, (size_t)h, rgb);
}

Texture* DecodeTGA(const std::vector<uint8_t>& d) {
  if (d.size() < 18) return nullptr;
  const uint8_t id_len = d[0], cmap_type = d[1], img_type = d[2];
  if (cmap_type != 0 || img_type != 2) return nullptr;  // uncompressed true colour only
  const long w = Le16(&d[12]), h = Le16(&d[14]);
  const uint8_t bpp = d[16], desc = d[17];
  if (!SaneSize(w, h) || (bpp != 24 && bpp != 32)) return nullptr;
  const size_t bytes_pp = bpp / 8, off = 18 + (size_t)id_len;
  if (d.size() < off + (size_t)w * h * bytes_pp) return nullptr;
  const bool top_down = (desc & 0x20) != 0;
  std::vector<uint8_t> rgb((size_t)w * h * 3);
  for (long y = 0; y < h; y++) {
    const uint8_t* row = &d[off + (size_t)(top_down ? y : h - 1 - y) * w * bytes_pp];
    for (long x = 0; x < w; x++) {
      const uint8_t* px = row + (size_t)x * bytes_pp;
      uint8_t* o = &rgb[((size_t)y * w + x) * 3];
      o[0] = px[2]; o[1] = px[1]; o[2] = px[0];
    }
  }
  return FromRgbRows((size_t)w, (size_t)h, rgb);
}

}  // namespace

Texture* Texture::LoadFromFile(const char* fname) {
  fprintf(stderr, "info: loading texture \"%s\"\n", fname);
  std::vector<uint8_t> d;
  if (!ReadAll(fname, &d) || d.size() < 4) return nullptr;
  Texture* t = nullptr;
  if (d[0] == 'P' && d[1] == '6') t = DecodePPM(d);
  else if (d[0] == 'B' && d[1] == 'M') t = DecodeBMP(d);
  else t = DecodeTGA(d);
  if (!t) fprintf(stderr, "error: unsupported or damaged texture file \"%s\"\n", fname);
  return t;
}

}  // namespace raytracer
