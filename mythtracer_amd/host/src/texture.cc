// texture.cc — texture files -> texels (reference: VerStarting/texture.cc:60-109).
//
// The reference decodes through SDL2_image; this build has no third-party
// decoder, so four self-contained formats are read directly: PNG (8-bit
// grey / grey+alpha / RGB / RGBA / palette, non-interlaced; own inflate),
// binary PPM (P6, maxval 255), uncompressed 24/32-bit BMP, uncompressed
// true-colour TGA.  The texel conversion is the reference's: the image is taken
// as RGBA32 and colour = byte / 255.0 (texture.cc:88-104), rows stored top to
// bottom, alpha dropped.  (JPEG stays out: its decoders do not agree bit for bit.)
#include "texture.h"

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

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

// ---- PNG -------------------------------------------------------------------
// RFC 1951 inflate (stored, fixed and dynamic Huffman blocks), straight from
// the specification: canonical codes decoded bit by bit by length.
struct BitReader {
  const uint8_t* p;
  size_t n, pos = 0;
  uint32_t bitbuf = 0;
  int bitcnt = 0;
  bool ok = true;
  int Bits(int need) {
    while (bitcnt < need) {
      if (pos >= n) { ok = false; return 0; }
      bitbuf |= (uint32_t)p[pos++] << bitcnt;
      bitcnt += 8;
    }
    const int v = (int)(bitbuf & ((1u << need) - 1u));
    bitbuf >>= need;
    bitcnt -= need;
    return v;
  }
};
struct Huffman {
  uint16_t count[16] = {0};
  uint16_t symbol[320] = {0};
  bool Build(const uint8_t* lengths, int n) {
    for (uint16_t& c : count) c = 0;
    for (int i = 0; i < n; i++) count[lengths[i]]++;
    int left = 1;
    for (int len = 1; len < 16; len++) {
      left <<= 1;
      left -= count[len];
      if (left < 0) return false;  // over-subscribed
    }
    uint16_t offs[16];
    offs[1] = 0;
    for (int len = 1; len < 15; len++) offs[len + 1] = (uint16_t)(offs[len] + count[len]);
    for (int i = 0; i < n; i++) {
      if (lengths[i] != 0) symbol[offs[lengths[i]]++] = (uint16_t)i;
    }
    return true;
  }
  int Decode(BitReader& br) const {
    int code = 0, first = 0, index = 0;
    for (int len = 1; len < 16; len++) {
      code |= br.Bits(1);
      if (!br.ok) return -1;
      const int c = count[len];
      if (code - c < first) return symbol[index + (code - first)];
      index += c;
      first += c;
      first <<= 1;
      code <<= 1;
    }
    return -1;
  }
};

bool Inflate(const uint8_t* src, size_t n, size_t expected, std::vector<uint8_t>* out) {
  static const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31,
                                        35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
  static const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
  static const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513,
                                         769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
  static const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8,
                                         9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
  BitReader br{src, n};
  out->clear();
  out->reserve(expected);
  for (bool last = false; !last;) {
    last = br.Bits(1) != 0;
    const int type = br.Bits(2);
    if (!br.ok) return false;
    if (type == 0) {  // stored
      br.bitbuf = 0;
      br.bitcnt = 0;
      if (br.pos + 4 > n) return false;
      const unsigned len = src[br.pos] | (src[br.pos + 1] << 8), nlen = src[br.pos + 2] | (src[br.pos + 3] << 8);
      br.pos += 4;
      if ((len ^ 0xffffu) != nlen || br.pos + len > n) return false;
      out->insert(out->end(), src + br.pos, src + br.pos + len);
      br.pos += len;
    } else if (type == 1 || type == 2) {
      Huffman lit, dist;
      uint8_t lengths[320];
      if (type == 1) {
        int i = 0;
        for (; i < 144; i++) lengths[i] = 8;
        for (; i < 256; i++) lengths[i] = 9;
        for (; i < 280; i++) lengths[i] = 7;
        for (; i < 288; i++) lengths[i] = 8;
        lit.Build(lengths, 288);
        for (i = 0; i < 30; i++) lengths[i] = 5;
        dist.Build(lengths, 30);
      } else {
        static const uint8_t kOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        const int nlen = br.Bits(5) + 257, ndist = br.Bits(5) + 1, ncode = br.Bits(4) + 4;
        if (!br.ok || nlen > 286 || ndist > 30) return false;
        uint8_t cl[19] = {0};
        for (int i = 0; i < ncode; i++) cl[kOrder[i]] = (uint8_t)br.Bits(3);
        Huffman lencode;
        if (!br.ok || !lencode.Build(cl, 19)) return false;
        int idx = 0;
        while (idx < nlen + ndist) {
          const int sym = lencode.Decode(br);
          if (sym < 0) return false;
          if (sym < 16) {
            lengths[idx++] = (uint8_t)sym;
          } else {
            int rep, val = 0;
            if (sym == 16) {
              if (idx == 0) return false;
              val = lengths[idx - 1];
              rep = 3 + br.Bits(2);
            } else if (sym == 17) {
              rep = 3 + br.Bits(3);
            } else {
              rep = 11 + br.Bits(7);
            }
            if (!br.ok || idx + rep > nlen + ndist) return false;
            while (rep--) lengths[idx++] = (uint8_t)val;
          }
        }
        if (lengths[256] == 0 || !lit.Build(lengths, nlen) || !dist.Build(lengths + nlen, ndist)) return false;
      }
      for (;;) {
        const int sym = lit.Decode(br);
        if (sym < 0) return false;
        if (sym < 256) {
          out->push_back((uint8_t)sym);
        } else if (sym == 256) {
          break;
        } else {
          if (sym - 257 >= 29) return false;
          const int len = kLenBase[sym - 257] + br.Bits(kLenExtra[sym - 257]);
          const int ds = dist.Decode(br);
          if (ds < 0 || ds >= 30) return false;
          const size_t d = (size_t)kDistBase[ds] + (size_t)br.Bits(kDistExtra[ds]);
          if (!br.ok || d > out->size()) return false;
          for (int k = 0; k < len; k++) out->push_back((*out)[out->size() - d]);
        }
        if (out->size() > expected) return false;  // more than the image can hold
      }
    } else {
      return false;
    }
  }
  return true;
}

uint32_t Be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | (p[1] << 16) | (p[2] << 8) | p[3]; }

Texture* DecodePNG(const std::vector<uint8_t>& d) {
  static const uint8_t kSig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (d.size() < 8 + 25 || memcmp(d.data(), kSig, 8) != 0) return nullptr;
  long w = 0, h = 0;
  int depth = 0, ctype = -1, interlace = 0;
  std::vector<uint8_t> idat, palette;
  bool seen_ihdr = false;
  for (size_t pos = 8; pos + 12 <= d.size();) {
    const uint32_t len = Be32(&d[pos]);
    if (len > 0x7fffffffu || pos + 12 + (size_t)len > d.size()) return nullptr;
    const uint8_t* type = &d[pos + 4];
    const uint8_t* body = &d[pos + 8];
    if (!memcmp(type, "IHDR", 4)) {
      if (len != 13) return nullptr;
      w = (long)Be32(body);
      h = (long)Be32(body + 4);
      depth = body[8];
      ctype = body[9];
      interlace = body[12];
      if (body[10] != 0 || body[11] != 0) return nullptr;
      seen_ihdr = true;
    } else if (!memcmp(type, "PLTE", 4)) {
      palette.assign(body, body + len);
    } else if (!memcmp(type, "IDAT", 4)) {
      idat.insert(idat.end(), body, body + len);
    } else if (!memcmp(type, "IEND", 4)) {
      break;
    }
    pos += 12 + (size_t)len;  // (chunk CRCs are not verified)
  }
  if (!seen_ihdr || !SaneSize(w, h) || depth != 8 || interlace != 0) return nullptr;
  int channels;
  switch (ctype) {
    case 0: channels = 1; break;  // grey
    case 2: channels = 3; break;  // RGB
    case 3: channels = 1; break;  // palette index
    case 4: channels = 2; break;  // grey + alpha
    case 6: channels = 4; break;  // RGBA
    default: return nullptr;
  }
  if (ctype == 3 && (palette.empty() || palette.size() % 3 != 0)) return nullptr;
  if (idat.size() < 6) return nullptr;
  // zlib wrapper: CMF/FLG, deflate data, Adler-32 (checked)
  if ((idat[0] & 0x0f) != 8 || ((idat[0] << 8) | idat[1]) % 31 != 0 || (idat[1] & 0x20)) return nullptr;
  const size_t stride = (size_t)w * channels, raw_size = (stride + 1) * (size_t)h;
  std::vector<uint8_t> raw;
  if (!Inflate(idat.data() + 2, idat.size() - 6, raw_size, &raw) || raw.size() != raw_size) return nullptr;
  uint32_t a = 1, b = 0;
  for (uint8_t v : raw) {
    a = (a + v) % 65521u;
    b = (b + a) % 65521u;
  }
  if (((b << 16) | a) != Be32(&idat[idat.size() - 4])) return nullptr;
  // undo the scanline filters (PNG specification, section 9)
  std::vector<uint8_t> img(stride * (size_t)h);
  for (long y = 0; y < h; y++) {
    const uint8_t ft = raw[(size_t)y * (stride + 1)];
    const uint8_t* in = &raw[(size_t)y * (stride + 1) + 1];
    uint8_t* cur = &img[(size_t)y * stride];
    const uint8_t* up = y > 0 ? &img[(size_t)(y - 1) * stride] : nullptr;
    if (ft > 4) return nullptr;
    for (size_t x = 0; x < stride; x++) {
      const int left = x >= (size_t)channels ? cur[x - channels] : 0;
      const int above = up ? up[x] : 0;
      const int upleft = (up && x >= (size_t)channels) ? up[x - channels] : 0;
      int pred = 0;
      if (ft == 1) pred = left;
      else if (ft == 2) pred = above;
      else if (ft == 3) pred = (left + above) >> 1;
      else if (ft == 4) {
        const int p = left + above - upleft, pa = abs(p - left), pb = abs(p - above), pc = abs(p - upleft);
        pred = (pa <= pb && pa <= pc) ? left : (pb <= pc ? above : upleft);
      }
      cur[x] = (uint8_t)(in[x] + pred);
    }
  }
  std::vector<uint8_t> rgb((size_t)w * h * 3);
  for (size_t i = 0; i < (size_t)w * h; i++) {
    const uint8_t* px = &img[i * channels];
    uint8_t* o = &rgb[i * 3];
    if (ctype == 2 || ctype == 6) {
      o[0] = px[0]; o[1] = px[1]; o[2] = px[2];
    } else if (ctype == 3) {
      if ((size_t)px[0] * 3 + 2 >= palette.size()) return nullptr;
      o[0] = palette[px[0] * 3]; o[1] = palette[px[0] * 3 + 1]; o[2] = palette[px[0] * 3 + 2];
    } else {
      o[0] = o[1] = o[2] = px[0];
    }
  }
  return FromRgbRows((size_t)w, (size_t)h, rgb);
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
  if (d[0] == 0x89 && d[1] == 'P') t = DecodePNG(d);
  else if (d[0] == 'P' && d[1] == '6') t = DecodePPM(d);
  else if (d[0] == 'B' && d[1] == 'M') t = DecodeBMP(d);
  else t = DecodeTGA(d);
  if (!t) fprintf(stderr, "error: unsupported or damaged texture file \"%s\"\n", fname);
  return t;
}

}  // namespace raytracer
