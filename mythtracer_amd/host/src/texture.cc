// texture.cc — texture files -> texels (reference: VerStarting/texture.cc:60-109).
//
// The reference decodes through SDL2_image; this build has no third-party
// decoder, so five self-contained formats are read directly: PNG (every colour
// type and bit depth of the specification, Adam7 interlacing; own inflate),
// JPEG (baseline and progressive Huffman, 8 bit; libjpeg's integer arithmetic
// restated, see DecodeJPEG), binary
// PPM (P6, maxval 255), uncompressed 24/32-bit BMP, uncompressed true-colour
// TGA.  The texel conversion is the reference's: the image is taken
// as RGBA32 and colour = byte / 255.0 (texture.cc:88-104), rows stored top to
// bottom, alpha dropped.
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
  if (!seen_ihdr || !SaneSize(w, h) || interlace > 1) return nullptr;
  int channels;
  switch (ctype) {
    case 0: channels = 1; break;  // grey
    case 2: channels = 3; break;  // RGB
    case 3: channels = 1; break;  // palette index
    case 4: channels = 2; break;  // grey + alpha
    case 6: channels = 4; break;  // RGBA
    default: return nullptr;
  }
  // bit depths the specification allows per colour type (section 11.2.2)
  const bool depth_ok = ctype == 0 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)
                      : ctype == 3 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8)
                                   : (depth == 8 || depth == 16);
  if (!depth_ok) return nullptr;
  if (ctype == 3 && (palette.empty() || palette.size() % 3 != 0)) return nullptr;
  if (idat.size() < 6) return nullptr;
  // zlib wrapper: CMF/FLG, deflate data, Adler-32 (checked)
  if ((idat[0] & 0x0f) != 8 || ((idat[0] << 8) | idat[1]) % 31 != 0 || (idat[1] & 0x20)) return nullptr;
  // The image arrives as one sub-image, or as the seven of Adam7 (section 8.2): pass p holds the pixels
  // (x0 + i dx, y0 + j dy); every sub-image has its own rows of filtered bytes, empty ones are left out.
  static const int kX0[7] = {0, 4, 0, 2, 0, 1, 0}, kY0[7] = {0, 0, 4, 0, 2, 0, 1};
  static const int kDx[7] = {8, 8, 4, 4, 2, 2, 1}, kDy[7] = {8, 8, 8, 4, 4, 2, 2};
  const int n_pass = interlace ? 7 : 1;
  const int bits_pp = depth * channels;          // bits per pixel
  const size_t bpp = (size_t)(bits_pp + 7) / 8;  // the filters' "corresponding byte" distance
  size_t raw_size = 0;
  long pw[7], ph[7];
  for (int p = 0; p < n_pass; p++) {
    pw[p] = interlace ? (w - kX0[p] + kDx[p] - 1) / kDx[p] : w;
    ph[p] = interlace ? (h - kY0[p] + kDy[p] - 1) / kDy[p] : h;
    if (pw[p] > 0 && ph[p] > 0) raw_size += (((size_t)pw[p] * bits_pp + 7) / 8 + 1) * (size_t)ph[p];
  }
  std::vector<uint8_t> raw;
  if (!Inflate(idat.data() + 2, idat.size() - 6, raw_size, &raw) || raw.size() != raw_size) return nullptr;
  uint32_t a = 1, b = 0;
  for (uint8_t v : raw) {
    a = (a + v) % 65521u;
    b = (b + a) % 65521u;
  }
  if (((b << 16) | a) != Be32(&idat[idat.size() - 4])) return nullptr;
  // What SDL2_image's libpng reader asks for before it hands over RGBA32 (IMG_png.c): 16-bit samples stripped to
  // their high byte (png_set_strip_16), samples below 8 bits unpacked, grey ones scaled to 8 bits
  // (png_set_expand: 1 bit x 255, 2 bits x 85, 4 bits x 17), palette indices looked up, grey replicated.
  std::vector<uint8_t> rgb((size_t)w * h * 3);
  std::vector<uint8_t> prev, cur;
  size_t at = 0;
  for (int p = 0; p < n_pass; p++) {
    if (pw[p] <= 0 || ph[p] <= 0) continue;
    const size_t stride = ((size_t)pw[p] * bits_pp + 7) / 8;
    prev.assign(stride, 0);
    cur.assign(stride, 0);
    for (long y = 0; y < ph[p]; y++) {
      const uint8_t ft = raw[at];
      const uint8_t* in = &raw[at + 1];
      at += stride + 1;
      if (ft > 4) return nullptr;
      // undo the scanline filter (PNG specification, section 9)
      for (size_t x = 0; x < stride; x++) {
        const int left = x >= bpp ? cur[x - bpp] : 0;
        const int above = prev[x];
        const int upleft = x >= bpp ? prev[x - bpp] : 0;
        int pred = 0;
        if (ft == 1) pred = left;
        else if (ft == 2) pred = above;
        else if (ft == 3) pred = (left + above) >> 1;
        else if (ft == 4) {
          const int q = left + above - upleft, pa = abs(q - left), pb = abs(q - above), pc = abs(q - upleft);
          pred = (pa <= pb && pa <= pc) ? left : (pb <= pc ? above : upleft);
        }
        cur[x] = (uint8_t)(in[x] + pred);
      }
      const long oy = interlace ? kY0[p] + y * kDy[p] : y;
      for (long x = 0; x < pw[p]; x++) {
        const long ox = interlace ? kX0[p] + x * kDx[p] : x;
        int v[4] = {0, 0, 0, 0};
        for (int c = 0; c < channels; c++) {
          if (depth == 8) v[c] = cur[(size_t)x * channels + c];
          else if (depth == 16) v[c] = cur[((size_t)x * channels + c) * 2];  // the high byte
          else {
            const size_t bit = (size_t)x * depth;  // (one channel below 8 bits)
            v[c] = (cur[bit >> 3] >> (8 - depth - (int)(bit & 7))) & ((1 << depth) - 1);
            if (ctype == 0) v[c] *= depth == 1 ? 255 : (depth == 2 ? 85 : 17);
          }
        }
        uint8_t* o = &rgb[((size_t)oy * w + ox) * 3];
        if (ctype == 2 || ctype == 6) {
          o[0] = (uint8_t)v[0]; o[1] = (uint8_t)v[1]; o[2] = (uint8_t)v[2];
        } else if (ctype == 3) {
          if ((size_t)v[0] * 3 + 2 >= palette.size()) return nullptr;
          o[0] = palette[v[0] * 3]; o[1] = palette[v[0] * 3 + 1]; o[2] = palette[v[0] * 3 + 2];
        } else {
          o[0] = o[1] = o[2] = (uint8_t)v[0];
        }
      }
      prev.swap(cur);
    }
  }
  return FromRgbRows((size_t)w, (size_t)h, rgb);
}

// ---- JPEG ---------------------------------------------------------------------
// Sequential and progressive DCT (spectral selection and successive approximation, ITU T.81 annex G -- what a
// 3ds-Max-era exporter's "progressive" check box writes), Huffman, 8 bit, grey or YCbCr with 1x1 / 2x1 / 2x2 chroma
// sampling, interleaved and per-component scans, restart intervals.  The reference decodes JPEG through SDL2_image,
// i.e. libjpeg with its defaults; decoders differ in the last bit unless they
// use the same integer arithmetic, so this one restates libjpeg's: the
// "islow" inverse DCT (jidctint.c: 13-bit constants, two passes), "fancy"
// (triangle-filter) chroma upsampling with replicated edge rows (jdsample.c)
// and the fixed-point YCbCr -> RGB tables (jdcolor.c).  tests/test_host_cpu.py
// pins it to libjpeg-turbo's output (through PIL).
struct JpegBits {
  const uint8_t* p;
  size_t n, pos;
  uint32_t buf = 0;
  int cnt = 0;
  bool marker_hit = false;
  int Bit() {
    if (cnt == 0) {
      int c = 0;
      if (!marker_hit && pos < n) {
        c = p[pos++];
        if (c == 0xff) {
          const int c2 = pos < n ? p[pos] : 0xd9;
          if (c2 == 0) pos++;                     // stuffed zero
          else { marker_hit = true; pos--; c = 0; }  // a marker: feed zeros (libjpeg does the same)
        }
      }
      buf = (uint32_t)c;
      cnt = 8;
    }
    cnt--;
    return (int)((buf >> cnt) & 1u);
  }
  int Bits(int k) {
    int v = 0;
    while (k--) v = (v << 1) | Bit();
    return v;
  }
  void Reset() { cnt = 0; marker_hit = false; }
};
struct JpegHuff {
  int mincode[17], maxcode[18], valptr[17];
  uint8_t vals[256];
  bool present = false;
  void Build(const uint8_t counts[16], const uint8_t* symbols, int nsym) {
    memcpy(vals, symbols, (size_t)nsym);
    int code = 0, k = 0;
    for (int len = 1; len <= 16; len++) {
      valptr[len] = k;
      mincode[len] = code;
      code += counts[len - 1];
      k += counts[len - 1];
      maxcode[len] = counts[len - 1] ? code - 1 : -1;
      code <<= 1;
    }
    maxcode[17] = 0x7fffffff;
    present = true;
  }
  int Decode(JpegBits& br) const {
    int code = 0;
    for (int len = 1; len <= 16; len++) {
      code = (code << 1) | br.Bit();
      if (maxcode[len] >= 0 && code <= maxcode[len] && code >= mincode[len]) return vals[valptr[len] + code - mincode[len]];
    }
    return -1;
  }
};
inline int JpegExtend(int v, int t) { return v < (1 << (t - 1)) ? v - (1 << t) + 1 : v; }

// jidctint.c jpeg_idct_islow: dequantized coefficients (natural order) -> 64 samples
void JpegIdctIslow(const int* in, uint8_t* out, int stride) {
  constexpr int CB = 13, P1 = 2;
  constexpr long F0298 = 2446, F0390 = 3196, F0541 = 4433, F0765 = 6270, F0899 = 7373, F1175 = 9633,
                 F1501 = 12299, F1847 = 15137, F1961 = 16069, F2053 = 16819, F2562 = 20995, F3072 = 25172;
  auto descale = [](long x, int n) { return (x + (1L << (n - 1))) >> n; };
  long ws[64];
  for (int c = 0; c < 8; c++) {  // pass 1: columns
    long z2 = in[16 + c], z3 = in[48 + c];
    long z1 = (z2 + z3) * F0541;
    long tmp2 = z1 + z3 * (-F1847), tmp3 = z1 + z2 * F0765;
    z2 = in[c];
    z3 = in[32 + c];
    long tmp0 = (z2 + z3) * (1L << CB), tmp1 = (z2 - z3) * (1L << CB);
    const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = in[56 + c]; tmp1 = in[40 + c]; tmp2 = in[24 + c]; tmp3 = in[8 + c];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    long z4 = tmp1 + tmp3;
    const long z5 = (z3 + z4) * F1175;
    tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
    z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    ws[c] = descale(tmp10 + tmp3, CB - P1);      ws[56 + c] = descale(tmp10 - tmp3, CB - P1);
    ws[8 + c] = descale(tmp11 + tmp2, CB - P1);  ws[48 + c] = descale(tmp11 - tmp2, CB - P1);
    ws[16 + c] = descale(tmp12 + tmp1, CB - P1); ws[40 + c] = descale(tmp12 - tmp1, CB - P1);
    ws[24 + c] = descale(tmp13 + tmp0, CB - P1); ws[32 + c] = descale(tmp13 - tmp0, CB - P1);
  }
  auto clamp = [](long v) { v += 128; return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); };
  for (int r = 0; r < 8; r++) {  // pass 2: rows
    const long* w = ws + r * 8;
    long z2 = w[2], z3 = w[6];
    long z1 = (z2 + z3) * F0541;
    long tmp2 = z1 + z3 * (-F1847), tmp3 = z1 + z2 * F0765;
    long tmp0 = (w[0] + w[4]) * (1L << CB), tmp1 = (w[0] - w[4]) * (1L << CB);
    const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    long z4 = tmp1 + tmp3;
    const long z5 = (z3 + z4) * F1175;
    tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
    z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    uint8_t* o = out + r * stride;
    constexpr int S = CB + P1 + 3;
    o[0] = clamp(descale(tmp10 + tmp3, S)); o[7] = clamp(descale(tmp10 - tmp3, S));
    o[1] = clamp(descale(tmp11 + tmp2, S)); o[6] = clamp(descale(tmp11 - tmp2, S));
    o[2] = clamp(descale(tmp12 + tmp1, S)); o[5] = clamp(descale(tmp12 - tmp1, S));
    o[3] = clamp(descale(tmp13 + tmp0, S)); o[4] = clamp(descale(tmp13 - tmp0, S));
  }
}

Texture* DecodeJPEG(const std::vector<uint8_t>& d) {
  static const uint8_t kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48,
                                      41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22,
                                      15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55,
                                      62, 63};
  // bw x bh: blocks of the component in the padded (whole-MCU) image, the layout of `coef` and `plane`;
  // cbw x cbh: blocks that hold pixels of the component -- what a scan of this component ALONE covers (ITU T.81 A.2.3)
  struct Comp {
    int id, h, v, tq, td, ta, bw, bh, cbw, cbh, dc_pred, pw, ph;
    bool q_latched;
    int q[64];
    std::vector<int16_t> coef;
    std::vector<uint8_t> plane;
  };
  int qt[4][64];
  bool qt_ok[4] = {false, false, false, false};
  JpegHuff hdc[4], hac[4];
  std::vector<Comp> comps;
  long W = 0, H = 0, mcux = 0, mcuy = 0;
  int restart = 0, adobe_transform = -1, hmax = 1, vmax = 1;
  bool sof = false, progressive = false, any_scan = false;
  size_t pos = 2;
  if (d.size() < 4 || d[0] != 0xff || d[1] != 0xd8) return nullptr;
  while (pos + 2 <= d.size()) {
    if (d[pos] != 0xff) return nullptr;
    const int m = d[pos + 1];
    pos += 2;
    if (m == 0xff) { pos--; continue; }  // fill byte
    if (m == 0xd8 || (m >= 0xd0 && m <= 0xd7) || m == 0x01) continue;
    if (m == 0xd9) break;
    if (pos + 2 > d.size()) return nullptr;
    const size_t len = ((size_t)d[pos] << 8) | d[pos + 1];
    if (len < 2 || pos + len > d.size()) return nullptr;
    const uint8_t* b = &d[pos + 2];
    const size_t bl = len - 2;
    if (m == 0xdb) {  // DQT
      for (size_t i = 0; i < bl;) {
        const int pq = b[i] >> 4, tq = b[i] & 15;
        i++;
        if (tq > 3 || i + (pq ? 128 : 64) > bl) return nullptr;
        for (int k = 0; k < 64; k++) {
          qt[tq][kZigzag[k]] = pq ? ((b[i] << 8) | b[i + 1]) : b[i];
          i += pq ? 2 : 1;
        }
        qt_ok[tq] = true;
      }
    } else if (m == 0xc0 || m == 0xc1 || m == 0xc2) {  // SOF0 / SOF1 (sequential) / SOF2 (progressive), Huffman, 8 bit
      if (sof || bl < 6 || b[0] != 8) return nullptr;
      progressive = m == 0xc2;
      H = (b[1] << 8) | b[2];
      W = (b[3] << 8) | b[4];
      const int nc = b[5];
      if ((nc != 1 && nc != 3) || bl < 6 + 3 * (size_t)nc || !SaneSize(W, H)) return nullptr;
      for (int i = 0; i < nc; i++) {
        Comp c{};
        c.id = b[6 + 3 * i];
        c.h = b[7 + 3 * i] >> 4;
        c.v = b[7 + 3 * i] & 15;
        c.tq = b[8 + 3 * i];
        if (c.tq > 3 || c.h < 1 || c.v < 1 || c.h > 2 || c.v > 2) return nullptr;
        comps.push_back(c);
      }
      if (nc == 3) {
        // luminance at full resolution, both chroma planes sampled alike: 1x1, 2x1 (h2v1) or 2x2 (h2v2)
        hmax = comps[0].h;
        vmax = comps[0].v;
        if (comps[1].h != 1 || comps[1].v != 1 || comps[2].h != 1 || comps[2].v != 1 || (hmax == 1 && vmax == 2)) return nullptr;
      } else {
        hmax = vmax = 1;  // a single component is never interleaved: one block per MCU
        comps[0].h = comps[0].v = 1;
      }
      mcux = (W + 8 * hmax - 1) / (8 * hmax);
      mcuy = (H + 8 * vmax - 1) / (8 * vmax);
      {
        // every block costs at least one bit in some scan: a header that promises more blocks than the file can
        // hold is refused before anything is allocated
        long blocks_per_mcu = 0;
        for (const Comp& c : comps) blocks_per_mcu += c.h * c.v;
        if (d.size() * 8 < (size_t)(mcux * mcuy * blocks_per_mcu)) return nullptr;
      }
      for (Comp& c : comps) {
        c.bw = (int)(mcux * c.h);
        c.bh = (int)(mcuy * c.v);
        c.cbw = (int)(((W * c.h + hmax - 1) / hmax + 7) / 8);
        c.cbh = (int)(((H * c.v + vmax - 1) / vmax + 7) / 8);
        c.pw = c.bw * 8;
        c.ph = c.bh * 8;
        c.coef.assign((size_t)c.bw * c.bh * 64, 0);
        c.q_latched = false;
      }
      sof = true;
    } else if (m >= 0xc3 && m <= 0xcf && m != 0xc4 && m != 0xc8 && m != 0xcc) {
      return nullptr;  // lossless, differential, arithmetic coding: not supported
    } else if (m == 0xc4) {  // DHT
      for (size_t i = 0; i < bl;) {
        if (i + 17 > bl) return nullptr;
        const int tc = b[i] >> 4, th = b[i] & 15;
        int total = 0;
        for (int k = 0; k < 16; k++) total += b[i + 1 + k];
        if (th > 3 || tc > 1 || total > 256 || i + 17 + (size_t)total > bl) return nullptr;
        (tc ? hac[th] : hdc[th]).Build(&b[i + 1], &b[i + 17], total);
        i += 17 + (size_t)total;
      }
    } else if (m == 0xdd) {  // DRI
      if (bl < 2) return nullptr;
      restart = (b[0] << 8) | b[1];
    } else if (m == 0xee) {  // Adobe APP14
      if (bl >= 12 && !memcmp(b, "Adobe", 5)) adobe_transform = b[11];
    } else if (m == 0xda) {  // SOS: one scan (jdhuff.c decode_mcu; jdphuff.c decode_mcu_DC/AC_first/refine)
      if (!sof || bl < 1) return nullptr;
      const int ns = b[0];
      if (ns < 1 || ns > (int)comps.size() || bl < 1 + 2 * (size_t)ns + 3) return nullptr;
      Comp* sc[3];
      for (int i = 0; i < ns; i++) {
        sc[i] = nullptr;
        for (Comp& c : comps) {
          if (c.id == b[1 + 2 * i]) {
            c.td = b[2 + 2 * i] >> 4;
            c.ta = b[2 + 2 * i] & 15;
            if (c.td < 4 && c.ta < 4) sc[i] = &c;
          }
        }
        if (!sc[i]) return nullptr;
        for (int j = 0; j < i; j++) {
          if (sc[j] == sc[i]) return nullptr;
        }
        if (!sc[i]->q_latched) {  // the table a component is dequantized with is the one in force at its first scan (jdinput.c latch_quant_tables)
          if (!qt_ok[sc[i]->tq]) return nullptr;
          memcpy(sc[i]->q, qt[sc[i]->tq], sizeof sc[i]->q);
          sc[i]->q_latched = true;
        }
      }
      const int Ss = b[1 + 2 * ns], Se = b[2 + 2 * ns], Ah = b[3 + 2 * ns] >> 4, Al = b[3 + 2 * ns] & 15;
      if (!progressive) {
        if (Ss != 0 || Se != 63 || Ah != 0 || Al != 0) return nullptr;
      } else {
        // jdphuff.c start_pass_phuff_decoder's checks: a DC scan is Ss = Se = 0, an AC scan has one component and
        // 1 <= Ss <= Se <= 63; successive approximation goes one bit at a time
        if (Ss > Se || Se > 63 || Al > 13 || (Ss == 0 && Se != 0) || (Ss != 0 && ns != 1) || (Ah != 0 && Ah - 1 != Al)) return nullptr;
      }
      const bool dc_scan = Ss == 0;
      for (int i = 0; i < ns; i++) {
        if ((!progressive || (dc_scan && Ah == 0)) && !hdc[sc[i]->td].present) return nullptr;
        if ((!progressive || !dc_scan) && !hac[sc[i]->ta].present) return nullptr;
        sc[i]->dc_pred = 0;
      }
      JpegBits br{d.data(), d.size(), pos + len};
      const long nx = ns > 1 ? mcux : sc[0]->cbw, ny = ns > 1 ? mcuy : sc[0]->cbh;
      int until_restart = restart;
      unsigned eobrun = 0;
      for (long my = 0; my < ny; my++) {
        for (long mx = 0; mx < nx; mx++) {
          if (restart && until_restart == 0) {
            br.Reset();
            // skip to the RSTn marker
            while (br.pos + 1 < d.size() && !(d[br.pos] == 0xff && d[br.pos + 1] >= 0xd0 && d[br.pos + 1] <= 0xd7)) br.pos++;
            br.pos += 2;
            for (int i = 0; i < ns; i++) sc[i]->dc_pred = 0;
            eobrun = 0;
            until_restart = restart;
          }
          for (int i = 0; i < ns; i++) {
            Comp& c = *sc[i];
            const int nbx = ns > 1 ? c.h : 1, nby = ns > 1 ? c.v : 1;
            for (int by = 0; by < nby; by++) {
              for (int bx = 0; bx < nbx; bx++) {
                const long gx = ns > 1 ? mx * c.h + bx : mx, gy = ns > 1 ? my * c.v + by : my;
                int16_t* blk = &c.coef[((size_t)gy * c.bw + gx) * 64];
                if (!progressive) {  // sequential: the whole block
                  const int t = hdc[c.td].Decode(br);
                  if (t < 0 || t > 11) return nullptr;
                  c.dc_pred += t ? JpegExtend(br.Bits(t), t) : 0;
                  if (c.dc_pred < -32768 || c.dc_pred > 32767) return nullptr;  // beyond any 8-bit image (libjpeg keeps it in a short)
                  blk[0] = (int16_t)c.dc_pred;
                  for (int k = 1; k < 64;) {
                    const int rs = hac[c.ta].Decode(br);
                    if (rs < 0) return nullptr;
                    const int r = rs >> 4, sz = rs & 15;
                    if (sz == 0) {
                      if (r != 15) break;  // end of block
                      k += 16;
                      continue;
                    }
                    k += r;
                    if (k > 63) return nullptr;
                    blk[kZigzag[k]] = (int16_t)JpegExtend(br.Bits(sz), sz);
                    k++;
                  }
                } else if (dc_scan && Ah == 0) {  // DC, first scan: the difference, scaled up by Al
                  const int t = hdc[c.td].Decode(br);
                  if (t < 0 || t > 11) return nullptr;
                  c.dc_pred += t ? JpegExtend(br.Bits(t), t) : 0;
                  if (c.dc_pred < -32768 || c.dc_pred > 32767) return nullptr;
                  blk[0] = (int16_t)(c.dc_pred * (1 << Al));
                } else if (dc_scan) {  // DC, refinement: one more bit
                  if (br.Bit()) blk[0] = (int16_t)(blk[0] | (1 << Al));
                } else if (Ah == 0) {  // AC band, first scan
                  if (eobrun > 0) {
                    eobrun--;  // a band of zeroes
                  } else {
                    for (int k = Ss; k <= Se; k++) {
                      const int rs = hac[c.ta].Decode(br);
                      if (rs < 0) return nullptr;
                      const int r = rs >> 4, sz = rs & 15;
                      if (sz) {
                        k += r;
                        if (k > 63) return nullptr;
                        blk[kZigzag[k]] = (int16_t)(JpegExtend(br.Bits(sz), sz) * (1 << Al));
                      } else if (r == 15) {
                        k += 15;  // sixteen zeroes
                      } else {    // EOBr: 2^r + appended bits blocks end here, this one included
                        eobrun = 1u << r;
                        if (r) eobrun += (unsigned)br.Bits(r);
                        eobrun--;
                        break;
                      }
                    }
                  }
                } else {  // AC band, refinement
                  const int p1 = 1 << Al, m1 = -(1 << Al);
                  int k = Ss;
                  auto correct = [&](int16_t* cf) {  // a correction bit: 1 = the magnitude grows by one step
                    if (br.Bit() && (*cf & p1) == 0) *cf = (int16_t)(*cf >= 0 ? *cf + p1 : *cf + m1);
                  };
                  if (eobrun == 0) {
                    for (; k <= Se; k++) {
                      const int rs = hac[c.ta].Decode(br);
                      if (rs < 0) return nullptr;
                      int r = rs >> 4, sv = rs & 15;
                      if (sv) {
                        sv = br.Bit() ? p1 : m1;  // (the size of a newly non-zero coefficient is always 1)
                      } else if (r != 15) {
                        eobrun = 1u << r;
                        if (r) eobrun += (unsigned)br.Bits(r);
                        break;  // the rest of the block is handled by the end-of-band logic
                      }
                      // advance over already-nonzero coefficients and r still-zero ones, correcting the nonzero ones
                      do {
                        int16_t* cf = &blk[kZigzag[k]];
                        if (*cf != 0) {
                          correct(cf);
                        } else if (--r < 0) {
                          break;  // reached the target zero coefficient
                        }
                        k++;
                      } while (k <= Se);
                      if (sv) {
                        if (k > 63) return nullptr;
                        blk[kZigzag[k]] = (int16_t)sv;
                      }
                    }
                  }
                  if (eobrun > 0) {
                    for (; k <= Se; k++) {
                      int16_t* cf = &blk[kZigzag[k]];
                      if (*cf != 0) correct(cf);
                    }
                    eobrun--;
                  }
                }
              }
            }
          }
          if (restart) until_restart--;
        }
      }
      any_scan = true;
      // the next marker: behind the entropy-coded bytes (stuffed zeroes and restart markers belong to them)
      size_t q = br.pos;
      while (q + 1 < d.size() && !(d[q] == 0xff && d[q + 1] != 0 && !(d[q + 1] >= 0xd0 && d[q + 1] <= 0xd7) && d[q + 1] != 0xff)) q++;
      if (q + 1 >= d.size()) break;  // no EOI: what has been decoded is the image (libjpeg warns and goes on)
      pos = q;
      continue;
    }
    pos += len;
  }
  if (!sof || !any_scan) return nullptr;
  // ---- dequantization and inverse DCT of every block (jddctmgr.c + jidctint.c)
  for (Comp& c : comps) {
    if (!c.q_latched) return nullptr;  // a component no scan brought
    c.plane.assign((size_t)c.pw * c.ph, 0);
    int in[64];
    for (int by = 0; by < c.bh; by++) {
      for (int bx = 0; bx < c.bw; bx++) {
        const int16_t* blk = &c.coef[((size_t)by * c.bw + bx) * 64];
        for (int k = 0; k < 64; k++) in[k] = blk[k] * c.q[k];
        JpegIdctIslow(in, &c.plane[(size_t)by * 8 * c.pw + (size_t)bx * 8], c.pw);
      }
    }
    std::vector<int16_t>().swap(c.coef);
  }
  // ---- upsampling (jdsample.c, do_fancy_upsampling) and colour conversion (jdcolor.c)
  std::vector<uint8_t> rgb((size_t)W * H * 3);
  if (comps.size() == 1) {
    for (long y = 0; y < H; y++) {
      for (long x = 0; x < W; x++) {
        const uint8_t g = comps[0].plane[(size_t)y * comps[0].pw + x];
        uint8_t* o = &rgb[((size_t)y * W + x) * 3];
        o[0] = o[1] = o[2] = g;
      }
    }
    return FromRgbRows((size_t)W, (size_t)H, rgb);
  }
  const long cw = (W + hmax - 1) / hmax, chh = (H + vmax - 1) / vmax;  // downsampled_width / height of the chroma planes
  std::vector<uint8_t> up[2];
  for (int ci = 0; ci < 2; ci++) {
    const Comp& c = comps[1 + ci];
    std::vector<uint8_t>& u = up[ci];
    u.assign((size_t)W * H + (size_t)2 * W + 16, 0);
    auto in = [&](long row, long col) -> int {  // edge rows replicated (jdmainct.c context rows)
      row = row < 0 ? 0 : (row >= chh ? chh - 1 : row);
      return c.plane[(size_t)row * c.pw + col];
    };
    if (hmax == 1 && vmax == 1) {
      for (long y = 0; y < H; y++) memcpy(&u[(size_t)y * W], &c.plane[(size_t)y * c.pw], (size_t)W);
    } else if (cw <= 2) {  // jinit_upsampler: planes this narrow get plain replication (h2v1_upsample / h2v2_upsample)
      for (long y = 0; y < H; y++) {
        for (long x = 0; x < W; x++) u[(size_t)y * W + x] = (uint8_t)in(y / vmax, x / 2);
      }
    } else if (hmax == 2 && vmax == 1) {  // h2v1_fancy_upsample
      for (long y = 0; y < H; y++) {
        std::vector<int> o((size_t)cw * 2);
        {
          o[0] = in(y, 0);
          o[1] = (in(y, 0) * 3 + in(y, 1) + 2) >> 2;
          for (long i = 1; i < cw - 1; i++) {
            const int v = in(y, i) * 3;
            o[2 * i] = (v + in(y, i - 1) + 1) >> 2;
            o[2 * i + 1] = (v + in(y, i + 1) + 2) >> 2;
          }
          o[2 * cw - 2] = (in(y, cw - 1) * 3 + in(y, cw - 2) + 1) >> 2;
          o[2 * cw - 1] = in(y, cw - 1);
        }
        for (long x = 0; x < W; x++) u[(size_t)y * W + x] = (uint8_t)o[x];
      }
    } else {  // h2v2_fancy_upsample
      for (long y = 0; y < H; y++) {
        const long r0 = y / 2, r1 = (y & 1) ? r0 + 1 : r0 - 1;  // nearer / farther input row
        std::vector<int> o((size_t)cw * 2);
        auto colsum = [&](long i) { return in(r0, i) * 3 + in(r1, i); };
        {
          int thiscs = colsum(0), nextcs = colsum(1), lastcs;
          o[0] = (thiscs * 4 + 8) >> 4;
          o[1] = (thiscs * 3 + nextcs + 7) >> 4;
          lastcs = thiscs;
          thiscs = nextcs;
          for (long i = 1; i < cw - 1; i++) {
            nextcs = colsum(i + 1);
            o[2 * i] = (thiscs * 3 + lastcs + 8) >> 4;
            o[2 * i + 1] = (thiscs * 3 + nextcs + 7) >> 4;
            lastcs = thiscs;
            thiscs = nextcs;
          }
          o[2 * cw - 2] = (thiscs * 3 + lastcs + 8) >> 4;
          o[2 * cw - 1] = (thiscs * 4 + 7) >> 4;
        }
        for (long x = 0; x < W; x++) u[(size_t)y * W + x] = (uint8_t)o[x];
      }
    }
  }
  const bool is_rgb = adobe_transform == 0 || (comps[0].id == 'R' && comps[1].id == 'G' && comps[2].id == 'B');
  auto clamp8 = [](int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); };
  for (long y = 0; y < H; y++) {
    for (long x = 0; x < W; x++) {
      const int Y = comps[0].plane[(size_t)y * comps[0].pw + x];
      const int cb = up[0][(size_t)y * W + x], cr = up[1][(size_t)y * W + x];
      uint8_t* o = &rgb[((size_t)y * W + x) * 3];
      if (is_rgb) {
        o[0] = (uint8_t)Y; o[1] = (uint8_t)cb; o[2] = (uint8_t)cr;
      } else {  // ycc_rgb_convert: SCALEBITS 16, ONE_HALF 32768
        const int xb = cb - 128, xr = cr - 128;
        const int r = (int)((91881L * xr + 32768) >> 16);
        const int bb = (int)((116130L * xb + 32768) >> 16);
        const int g = (int)(((-22554L * xb + 32768) + (-46802L * xr)) >> 16);
        o[0] = clamp8(Y + r);
        o[1] = clamp8(Y + g);
        o[2] = clamp8(Y + bb);
      }
    }
  }
  return FromRgbRows((size_t)W, (size_t)H, rgb);
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
  else if (d[0] == 0xff && d[1] == 0xd8) t = DecodeJPEG(d);
  else if (d[0] == 'P' && d[1] == '6') t = DecodePPM(d);
  else if (d[0] == 'B' && d[1] == 'M') t = DecodeBMP(d);
  else t = DecodeTGA(d);
  if (!t) fprintf(stderr, "error: unsupported or damaged texture file \"%s\"\n", fname);
  return t;
}

}  // namespace raytracer
