// image_formats.cpp -- the less common 8-bit formats a pbrlab texture file may come in: BMP, TGA, PNM (P5/P6), GIF (first
// frame), PSD (composited RGB image) and Radiance HDR opened under a non-.hdr name (tone-mapped to 8 bits).
//
// Reference: io::LoadImageFromFile (src/io/image-io.cc:98-152) hands every file that is not .exr / .hdr to stb_image's
// stbi_load (vendored v2.25), which tries JPEG, PNG, BMP, GIF, PSD, PIC, PNM, HDR, TGA in that order.  These decoders are
// written from scratch against the file formats; where stb_image's reading of a format is peculiar, its result is what a
// pbrlab scene gets, so it is reproduced and marked "stb:" (tests compare every decoder with the reference's own
// stb_image on generated files, tests/test_io_cpu.py).  Not decoded: Softimage PIC.
#include <cmath>
#include <cstring>

#include "image_codec.h"

namespace pbio {
namespace {

constexpr size_t kMaxPixels = size_t(1) << 28;  // same cap as the other decoders (image_codec.cpp)

// Byte cursor over the file image.  Reading past the end yields zeros, like a stream that has run dry.
struct Cursor {
  const uint8_t* p;
  size_t n, i = 0;
  Cursor(const uint8_t* data, size_t size) : p(data), n(size) {}
  bool eof() const { return i >= n; }
  uint8_t u8() { return i < n ? p[i++] : uint8_t(0); }
  uint32_t le16() {
    uint32_t a = u8();
    return a | (uint32_t(u8()) << 8);
  }
  uint32_t le32() {
    uint32_t a = le16();
    return a | (le16() << 16);
  }
  uint32_t be16() {
    uint32_t a = u8();
    return (a << 8) | u8();
  }
  uint32_t be32() {
    uint32_t a = be16();
    return (a << 16) | be16();
  }
  void skip(long long k) {
    if (k <= 0) return;  // stb: a negative skip puts the stream at its end; only corrupt headers produce one
    i = (size_t(k) > n - (i < n ? i : n)) ? n : i + size_t(k);
  }
  void read(uint8_t* dst, size_t k) {
    for (size_t j = 0; j < k; j++) dst[j] = u8();
  }
};

bool fail(std::string* err, const char* msg) {
  if (err) *err = msg;
  return false;
}
bool size_ok(long long w, long long h, long long c) { return w > 0 && h > 0 && c > 0 && size_t(w) * size_t(h) <= kMaxPixels; }

// ------------------------------------------------------------------ BMP
int top_bit(uint32_t z) {  // index of the highest set bit, -1 for 0
  int n = -1;
  while (z) n++, z >>= 1;
  return n;
}
int bit_count(uint32_t z) {
  int n = 0;
  for (; z; z &= z - 1) n++;
  return n;
}
// A masked field of `bits` bits whose top bit sits at position 7 + shift -> 8 bits, the field repeated to fill the byte
// (what stb's multiply/shift tables compute: 1 bit -> 0/255, 5 bits -> v<<3 | v>>2, ...)
int expand_field(uint32_t v, int shift, int bits) {
  static const uint32_t mul[9] = {0, 0xff, 0x55, 0x49, 0x11, 0x21, 0x41, 0x81, 0x01};
  static const int down[9] = {0, 0, 0, 1, 0, 2, 4, 6, 0};
  if (shift < 0) v <<= -shift;
  else v >>= shift;
  v >>= (8 - bits);
  return int(v * mul[bits]) >> down[bits];
}

}  // namespace

bool IsBmp(const uint8_t* f, size_t n) {
  if (n < 18 || f[0] != 'B' || f[1] != 'M') return false;
  const uint32_t hsz = uint32_t(f[14]) | (uint32_t(f[15]) << 8) | (uint32_t(f[16]) << 16) | (uint32_t(f[17]) << 24);
  return hsz == 12 || hsz == 40 || hsz == 56 || hsz == 108 || hsz == 124;
}

bool DecodeBmp(const uint8_t* file, size_t n, std::vector<uint8_t>* pixels, size_t* width, size_t* height, size_t* channels,
               std::string* err) {
  Cursor s(file, n);
  if (s.u8() != 'B' || s.u8() != 'M') return fail(err, "not a BMP file");
  s.skip(8);
  const long long data_offset = int32_t(s.le32());
  const int hsz = int(s.le32());
  if (hsz != 12 && hsz != 40 && hsz != 56 && hsz != 108 && hsz != 124) return fail(err, "BMP: unknown header size");
  long long w, h;
  if (hsz == 12) w = s.le16(), h = s.le16();
  else w = int32_t(s.le32()), h = int32_t(s.le32());
  if (s.le16() != 1) return fail(err, "BMP: plane count is not 1");
  const int bpp = int(s.le16());
  uint32_t mr = 0, mg = 0, mb = 0, ma = 0, all_a = 255;
  int header_read = 14;  // bytes before the info header, plus mask words that follow a 40/56-byte header
  if (hsz != 12) {
    const uint32_t compress = s.le32();
    if (compress == 1 || compress == 2) return fail(err, "BMP: RLE compression is not supported");
    s.skip(20);
    if (hsz == 40 || hsz == 56) {
      if (hsz == 56) s.skip(16);
      if (bpp == 16 || bpp == 32) {
        if (compress == 0) {
          if (bpp == 32) mr = 0xffu << 16, mg = 0xffu << 8, mb = 0xffu, ma = 0xffu << 24, all_a = 0;
          else mr = 31u << 10, mg = 31u << 5, mb = 31u;
        } else if (compress == 3) {
          mr = s.le32(), mg = s.le32(), mb = s.le32();
          header_read += 12;
          if (mr == mg && mg == mb) return fail(err, "BMP: bad channel masks");
        } else {
          return fail(err, "BMP: unknown compression");
        }
      }
    } else {
      mr = s.le32(), mg = s.le32(), mb = s.le32(), ma = s.le32();
      s.skip(4 + 48);
      if (hsz == 124) s.skip(16);
    }
  }
  const bool bottom_up = h > 0;
  if (h < 0) h = -h;
  const int comp = (bpp == 24 && ma == 0xff000000u) ? 3 : (ma ? 4 : 3);
  if (!size_ok(w, h, comp)) return fail(err, "BMP: bad dimensions");
  const size_t W = size_t(w), H = size_t(h);
  pixels->assign(W * H * size_t(comp), 0);
  uint8_t* out = pixels->data();
  size_t z = 0;
  if (bpp < 16) {
    // stb: the palette size is derived from the data offset; for the 12-byte header its formula is 12 bytes short
    long long psize = 0;
    if (hsz == 12) psize = (data_offset - header_read - 24) / 3;
    else psize = (data_offset - header_read - hsz) >> 2;
    if (psize <= 0 || psize > 256) return fail(err, "BMP: bad palette size");
    uint8_t pal[256][3];
    memset(pal, 0, sizeof(pal));
    for (long long i = 0; i < psize; i++) {
      pal[i][2] = s.u8(), pal[i][1] = s.u8(), pal[i][0] = s.u8();
      if (hsz != 12) s.u8();
    }
    s.skip(data_offset - header_read - hsz - psize * (hsz == 12 ? 3 : 4));
    size_t row_bytes;
    if (bpp == 1) row_bytes = (W + 7) >> 3;
    else if (bpp == 4) row_bytes = (W + 1) >> 1;
    else if (bpp == 8) row_bytes = W;
    else return fail(err, "BMP: unsupported bit depth");
    const size_t pad = (0 - row_bytes) & 3;
    for (size_t j = 0; j < H; j++) {
      size_t used = 0;
      uint32_t cur = 0;
      int left = 0;  // unread bits of cur
      for (size_t i = 0; i < W; i++) {
        if (left == 0) cur = s.u8(), left = 8, used++;
        left -= bpp;
        const uint32_t idx = (cur >> left) & ((1u << bpp) - 1u);
        out[z++] = pal[idx][0], out[z++] = pal[idx][1], out[z++] = pal[idx][2];
        if (comp == 4) out[z++] = 255;
      }
      s.skip((long long)(row_bytes - used + pad));
    }
  } else {
    s.skip(data_offset - header_read - hsz);
    const size_t row_bytes = bpp == 24 ? 3 * W : (bpp == 16 ? 2 * W : 0);
    const size_t pad = (0 - row_bytes) & 3;
    int direct = 0;  // 1: B,G,R bytes; 2: B,G,R,A bytes
    if (bpp == 24) direct = 1;
    else if (bpp == 32 && mb == 0xffu && mg == 0xff00u && mr == 0x00ff0000u && ma == 0xff000000u) direct = 2;
    int rs = 0, gs = 0, bs = 0, as = 0, rc = 0, gc = 0, bc = 0, ac = 0;
    if (!direct) {
      if (!mr || !mg || !mb) return fail(err, "BMP: bad channel masks");
      rs = top_bit(mr) - 7, rc = bit_count(mr);
      gs = top_bit(mg) - 7, gc = bit_count(mg);
      bs = top_bit(mb) - 7, bc = bit_count(mb);
      as = top_bit(ma) - 7, ac = bit_count(ma);
      if (rc > 8 || gc > 8 || bc > 8 || ac > 8) return fail(err, "BMP: channel masks wider than 8 bits");
    }
    for (size_t j = 0; j < H; j++) {
      for (size_t i = 0; i < W; i++) {
        uint32_t a;
        if (direct) {
          const uint8_t b = s.u8(), g = s.u8(), r = s.u8();
          out[z++] = r, out[z++] = g, out[z++] = b;
          a = direct == 2 ? s.u8() : 255u;
        } else {
          const uint32_t v = bpp == 16 ? s.le16() : s.le32();
          out[z++] = uint8_t(expand_field(v & mr, rs, rc));
          out[z++] = uint8_t(expand_field(v & mg, gs, gc));
          out[z++] = uint8_t(expand_field(v & mb, bs, bc));
          a = ma ? uint32_t(expand_field(v & ma, as, ac)) : 255u;
        }
        all_a |= a;
        if (comp == 4) out[z++] = uint8_t(a);
      }
      s.skip((long long)pad);
    }
  }
  if (comp == 4 && all_a == 0)  // stb: a 32-bit file whose alpha bytes are all 0 has no alpha: opaque
    for (size_t i = 3; i < pixels->size(); i += 4) out[i] = 255;
  if (bottom_up) {
    const size_t rb = W * size_t(comp);
    std::vector<uint8_t> tmp(rb);
    for (size_t j = 0; j < H / 2; j++) {
      uint8_t *a = out + j * rb, *b = out + (H - 1 - j) * rb;
      memcpy(tmp.data(), a, rb), memcpy(a, b, rb), memcpy(b, tmp.data(), rb);
    }
  }
  *width = W, *height = H, *channels = size_t(comp);
  return true;
}

// ------------------------------------------------------------------ TGA
namespace {
int tga_components(int bits, bool grey, bool* rgb16) {
  *rgb16 = false;
  switch (bits) {
    case 8: return 1;
    case 16:
      if (grey) return 2;
      *rgb16 = true;
      return 3;
    case 15: *rgb16 = true; return 3;
    case 24: return 3;
    case 32: return 4;
  }
  return 0;
}
void tga_rgb555(Cursor& s, uint8_t* out) {  // 5-5-5, top bit ignored (stb: 15/16-bit TGAs have no alpha)
  const uint32_t px = s.le16();
  out[0] = uint8_t((((px >> 10) & 31u) * 255u) / 31u);
  out[1] = uint8_t((((px >> 5) & 31u) * 255u) / 31u);
  out[2] = uint8_t(((px & 31u) * 255u) / 31u);
}
}  // namespace

// TGA has no signature: this is the plausibility test stb applies, after every other format has been ruled out
bool IsTga(const uint8_t* f, size_t n) {
  Cursor s(f, n);
  s.u8();
  const int cmap = s.u8();
  if (cmap > 1) return false;
  int t = s.u8();
  if (cmap == 1) {
    if (t != 1 && t != 9) return false;
    s.skip(4);
    const int pb = s.u8();
    if (pb != 8 && pb != 15 && pb != 16 && pb != 24 && pb != 32) return false;
    s.skip(4);
  } else {
    if (t != 2 && t != 3 && t != 10 && t != 11) return false;
    s.skip(9);
  }
  if (s.le16() < 1 || s.le16() < 1) return false;
  const int bpp = s.u8();
  if (cmap == 1 && bpp != 8 && bpp != 16) return false;
  return bpp == 8 || bpp == 15 || bpp == 16 || bpp == 24 || bpp == 32;
}

bool DecodeTga(const uint8_t* file, size_t n, std::vector<uint8_t>* pixels, size_t* width, size_t* height, size_t* channels,
               std::string* err) {
  Cursor s(file, n);
  const int id_len = s.u8();
  const int indexed = s.u8();
  int type = s.u8();
  const int pal_start = int(s.le16()), pal_len = int(s.le16()), pal_bits = s.u8();
  s.skip(4);  // x / y origin
  const size_t W = s.le16(), H = s.le16();
  const int bpp = s.u8();
  const int descriptor = s.u8();
  const bool rle = type >= 8;
  if (rle) type -= 8;
  const bool bottom_up = ((descriptor >> 5) & 1) == 0;
  bool rgb16 = false;
  const int comp = indexed ? tga_components(pal_bits, false, &rgb16) : tga_components(bpp, type == 3, &rgb16);
  if (!comp) return fail(err, "TGA: unsupported pixel format");
  if (!size_ok((long long)W, (long long)H, comp)) return fail(err, "TGA: bad dimensions");
  const size_t C = size_t(comp);
  pixels->assign(W * H * C, 0);
  uint8_t* out = pixels->data();
  s.skip(id_len);
  std::vector<uint8_t> pal;
  if (indexed) {
    s.skip(pal_start);  // stb: the index of the first colour-map entry is skipped as a byte count
    pal.assign(size_t(pal_len) * C + 4, 0);
    if (rgb16) {
      for (int i = 0; i < pal_len; i++) tga_rgb555(s, &pal[size_t(i) * C]);
    } else {
      if (s.n - (s.i < s.n ? s.i : s.n) < size_t(pal_len) * C) return fail(err, "TGA: truncated colour map");
      s.read(pal.data(), size_t(pal_len) * C);
    }
  }
  uint8_t px[4] = {0, 0, 0, 0};
  int run = 0;
  bool repeat = false;
  for (size_t i = 0; i < W * H; i++) {
    bool fetch = true;
    if (rle) {
      if (run == 0) {
        const int cmd = s.u8();
        run = 1 + (cmd & 127), repeat = (cmd >> 7) != 0;
      } else if (repeat) {
        fetch = false;
      }
    }
    if (fetch) {
      if (indexed) {
        size_t idx = bpp == 8 ? s.u8() : s.le16();
        if (idx >= size_t(pal_len)) idx = 0;
        for (size_t j = 0; j < C; j++) px[j] = pal[idx * C + j];
      } else if (rgb16) {
        tga_rgb555(s, px);
      } else {
        for (size_t j = 0; j < C; j++) px[j] = s.u8();
      }
    }
    for (size_t j = 0; j < C; j++) out[i * C + j] = px[j];
    run--;
  }
  if (bottom_up) {
    const size_t rb = W * C;
    std::vector<uint8_t> tmp(rb);
    for (size_t j = 0; j < H / 2; j++) {
      uint8_t *a = out + j * rb, *b = out + (H - 1 - j) * rb;
      memcpy(tmp.data(), a, rb), memcpy(a, b, rb), memcpy(b, tmp.data(), rb);
    }
  }
  if (comp >= 3 && !rgb16)  // stored B,G,R(,A)
    for (size_t i = 0; i < W * H; i++) std::swap(out[i * C], out[i * C + 2]);
  *width = W, *height = H, *channels = C;
  return true;
}

// ------------------------------------------------------------------ PNM (binary PGM / PPM, 8 bits)
bool IsPnm(const uint8_t* f, size_t n) { return n >= 2 && f[0] == 'P' && (f[1] == '5' || f[1] == '6'); }

bool DecodePnm(const uint8_t* file, size_t n, std::vector<uint8_t>* pixels, size_t* width, size_t* height, size_t* channels,
               std::string* err) {
  Cursor s(file, n);
  if (s.u8() != 'P') return fail(err, "not a PNM file");
  const int kind = s.u8();
  if (kind != '5' && kind != '6') return fail(err, "PNM: only P5 / P6 are supported");
  const size_t C = kind == '6' ? 3 : 1;
  char c = char(s.u8());
  auto is_space = [](char ch) { return ch == ' ' || ch == '\t' || ch == '\n' || ch == '\v' || ch == '\f' || ch == '\r'; };
  auto skip_blank = [&]() {  // white space and '#' comments
    for (;;) {
      while (!s.eof() && is_space(c)) c = char(s.u8());
      if (s.eof() || c != '#') break;
      while (!s.eof() && c != '\n' && c != '\r') c = char(s.u8());
    }
  };
  auto number = [&]() {
    long long v = 0;
    while (!s.eof() && c >= '0' && c <= '9') {
      v = v * 10 + (c - '0');
      if (v > (1ll << 40)) v = 1ll << 40;
      c = char(s.u8());
    }
    return v;
  };
  skip_blank();
  const long long w = number();
  skip_blank();
  const long long h = number();
  skip_blank();
  const long long maxv = number();  // the single white-space byte after it has been consumed
  if (maxv > 255) return fail(err, "PNM: more than 8 bits per sample");
  if (!size_ok(w, h, (long long)C)) return fail(err, "PNM: bad dimensions");
  pixels->assign(size_t(w) * size_t(h) * C, 0);
  s.read(pixels->data(), pixels->size());
  *width = size_t(w), *height = size_t(h), *channels = C;
  return true;
}

// ------------------------------------------------------------------ GIF (first frame, always RGBA)
bool IsGif(const uint8_t* f, size_t n) {
  return n >= 6 && memcmp(f, "GIF8", 4) == 0 && (f[4] == '7' || f[4] == '9') && f[5] == 'a';
}

bool DecodeGif(const uint8_t* file, size_t n, std::vector<uint8_t>* pixels, size_t* width, size_t* height, size_t* channels,
               std::string* err) {
  if (!IsGif(file, n)) return fail(err, "not a GIF file");
  Cursor s(file, n);
  s.skip(6);
  const size_t W = s.le16(), H = s.le16();
  const int flags = s.u8(), bg_index = s.u8();
  s.u8();  // aspect ratio
  // palette entries as {r, g, b, a}
  uint8_t gpal[256][4], lpal[256][4];
  memset(gpal, 0, sizeof(gpal)), memset(lpal, 0, sizeof(lpal));
  auto read_palette = [&](uint8_t pal[256][4], int count, int transparent) {
    for (int i = 0; i < count; i++) {
      pal[i][0] = s.u8(), pal[i][1] = s.u8(), pal[i][2] = s.u8();
      pal[i][3] = (i == transparent) ? 0 : 255;
    }
  };
  if (flags & 0x80) read_palette(gpal, 2 << (flags & 7), -1);
  if (W * H > kMaxPixels) return fail(err, "GIF: image too large");
  pixels->assign(W * H * 4, 0);
  std::vector<uint8_t> drawn(W * H + 1, 0);
  uint8_t* out = pixels->data();
  int transparent = -1, gce_flags = 0;
  for (;;) {
    const int tag = s.u8();
    if (tag == 0x21) {  // extension
      const int label = s.u8();
      int len;
      if (label == 0xF9) {  // graphic control
        len = s.u8();
        if (len != 4) {
          // stb: skips the block and then reads the next byte as a block tag, which a sub-block chain never is
          return fail(err, "GIF: malformed graphic control extension");
        }
        gce_flags = s.u8();
        s.le16();  // delay
        if (transparent >= 0) gpal[transparent][3] = 255;
        if (gce_flags & 1) {
          transparent = s.u8();
          gpal[transparent][3] = 0;
        } else {
          s.skip(1);
          transparent = -1;
        }
      }
      while ((len = s.u8()) != 0) s.skip(len);
      if (s.eof()) return fail(err, "GIF: truncated");
      continue;
    }
    if (tag != 0x2C) return fail(err, tag == 0x3B ? "GIF: no image" : "GIF: unknown block");
    // image descriptor
    const size_t x = s.le16(), y = s.le16(), w = s.le16(), h = s.le16();
    if (x + w > W || y + h > H) return fail(err, "GIF: frame outside the canvas");
    const int lflags = s.u8();
    const uint8_t(*pal)[4];
    if (lflags & 0x80) {
      read_palette(lpal, 2 << (lflags & 7), (gce_flags & 1) ? transparent : -1);
      pal = lpal;
    } else if (flags & 0x80) {
      pal = gpal;
    } else {
      return fail(err, "GIF: missing colour table");
    }
    // raster: rows of the frame, interlaced in four passes when bit 6 is set
    const bool interlaced = (lflags & 0x40) != 0;
    size_t col = 0, row = 0;     // position inside the frame
    int pass = interlaced ? 3 : 0;
    size_t row_step = interlaced ? 8 : 1;
    bool full = (w == 0) || (h == 0);
    auto put = [&](uint8_t index) {
      if (full) return;
      const size_t at = (y + row) * W + (x + col);
      drawn[at] = 1;
      if (pal[index][3] > 128) memcpy(out + at * 4, pal[index], 4);  // transparent pixels leave the canvas as it is
      if (++col < w) return;
      col = 0, row += row_step;
      while (row >= h && pass > 0) {
        row_step = size_t(1) << pass;
        row = row_step >> 1;
        pass--;
      }
      if (row >= h) full = true;
    };
    // LZW
    const int min_bits = s.u8();
    if (min_bits > 12) return fail(err, "GIF: bad LZW code size");
    const int clear = 1 << min_bits;
    struct Entry {
      int16_t prefix;
      uint8_t first, suffix;
    };
    std::vector<Entry> table(8192);
    for (int i = 0; i < clear; i++) table[size_t(i)] = {int16_t(-1), uint8_t(i), uint8_t(i)};
    int code_bits = min_bits + 1, mask = (1 << code_bits) - 1, avail = clear + 2, prev = -1;
    bool seen_clear = false;
    uint32_t acc = 0;
    int have = 0, block_left = 0;
    std::vector<uint8_t> run(8192);
    bool done = false;
    while (!done) {
      if (have < code_bits) {
        if (block_left == 0) {
          block_left = s.u8();
          if (block_left == 0) break;  // block terminator: the frame is what has been drawn
        }
        block_left--;
        acc |= uint32_t(s.u8()) << have;
        have += 8;
        continue;
      }
      const int code = int(acc & uint32_t(mask));
      acc >>= code_bits, have -= code_bits;
      if (code == clear) {
        code_bits = min_bits + 1, mask = (1 << code_bits) - 1, avail = clear + 2, prev = -1;
        seen_clear = true;
      } else if (code == clear + 1) {
        s.skip(block_left);
        int len;
        while ((len = s.u8()) > 0) s.skip(len);
        done = true;
      } else if (code <= avail) {
        if (!seen_clear) return fail(err, "GIF: no clear code");
        if (prev >= 0) {
          if (avail + 1 > 8192) return fail(err, "GIF: too many codes");
          Entry& e = table[size_t(avail++)];
          e.prefix = int16_t(prev);
          e.first = table[size_t(prev)].first;
          e.suffix = (code == avail) ? e.first : table[size_t(code)].first;
        } else if (code == avail) {
          return fail(err, "GIF: illegal code");
        }
        size_t len = 0;
        for (int c = code; c >= 0 && len < run.size(); c = table[size_t(c)].prefix) run[len++] = table[size_t(c)].suffix;
        while (len) put(run[--len]);
        if ((avail & mask) == 0 && avail <= 0x0FFF) code_bits++, mask = (1 << code_bits) - 1;
        prev = code;
      } else {
        return fail(err, "GIF: illegal code");
      }
    }
    // stb: canvas pixels the first frame did not touch take the background entry -- copied in the order stb stores its
    // palette (b, g, r), i.e. with red and blue exchanged -- when the background index is not 0
    if (bg_index > 0)
      for (size_t i = 0; i < W * H; i++)
        if (!drawn[i]) out[i * 4 + 0] = gpal[bg_index][2], out[i * 4 + 1] = gpal[bg_index][1], out[i * 4 + 2] = gpal[bg_index][0], out[i * 4 + 3] = 255;
    *width = W, *height = H, *channels = 4;
    return true;
  }
}

// ------------------------------------------------------------------ PSD (the merged image of an 8/16-bit RGB document)
bool IsPsd(const uint8_t* f, size_t n) { return n >= 4 && memcmp(f, "8BPS", 4) == 0; }

bool DecodePsd(const uint8_t* file, size_t n, std::vector<uint8_t>* pixels, size_t* width, size_t* height, size_t* channels,
               std::string* err) {
  Cursor s(file, n);
  if (s.be32() != 0x38425053u) return fail(err, "not a PSD file");
  if (s.be16() != 1) return fail(err, "PSD: unsupported version");
  s.skip(6);
  const int nch = int(s.be16());
  if (nch > 16) return fail(err, "PSD: unsupported channel count");
  const long long h = int32_t(s.be32()), w = int32_t(s.be32());
  const int depth = int(s.be16());
  if (depth != 8 && depth != 16) return fail(err, "PSD: bit depth is not 8 or 16");
  if (s.be16() != 3) return fail(err, "PSD: not an RGB document");
  s.skip(s.be32());  // colour mode data
  s.skip(s.be32());  // image resources
  s.skip(s.be32());  // layer and mask information
  const int compression = int(s.be16());
  if (compression > 1) return fail(err, "PSD: unknown compression");
  if (!size_ok(w, h, 4)) return fail(err, "PSD: bad dimensions");
  const size_t count = size_t(w) * size_t(h);
  pixels->assign(count * 4, 0);
  uint8_t* out = pixels->data();
  if (compression) s.skip((long long)h * nch * 2);  // per-row byte counts
  for (int ch = 0; ch < 4; ch++) {
    uint8_t* p = out + ch;
    if (ch >= nch) {
      for (size_t i = 0; i < count; i++) p[i * 4] = ch == 3 ? 255 : 0;
    } else if (compression) {  // PackBits over the whole plane
      size_t done = 0;
      while (done < count) {
        int len = s.u8();
        if (len == 128) {
          if (s.eof()) return fail(err, "PSD: truncated");
          continue;
        }
        if (len < 128) {
          len++;
          if (size_t(len) > count - done) return fail(err, "PSD: bad RLE data");
          for (int k = 0; k < len; k++) p[(done + size_t(k)) * 4] = s.u8();
        } else {
          len = 257 - len;
          if (size_t(len) > count - done) return fail(err, "PSD: bad RLE data");
          const uint8_t v = s.u8();
          for (int k = 0; k < len; k++) p[(done + size_t(k)) * 4] = v;
        }
        done += size_t(len);
      }
    } else if (depth == 16) {
      for (size_t i = 0; i < count; i++) p[i * 4] = uint8_t(s.be16() >> 8);
    } else {
      for (size_t i = 0; i < count; i++) p[i * 4] = s.u8();
    }
  }
  if (nch >= 4) {  // stb: colours of partly transparent pixels are un-blended from a white matte, in float
    for (size_t i = 0; i < count; i++) {
      uint8_t* px = out + 4 * i;
      if (px[3] != 0 && px[3] != 255) {
        const float a = px[3] / 255.0f;
        const float ra = 1.0f / a;
        const float inv_a = 255.0f * (1 - ra);
        for (int k = 0; k < 3; k++) {
          const float v = px[k] * ra + inv_a;
          px[k] = uint8_t(int(v));
        }
      }
    }
  }
  *width = size_t(w), *height = size_t(h), *channels = 4;
  return true;
}

// ------------------------------------------------------------------ Radiance HDR opened as an 8-bit image
// stb: value -> pow(value, 1/2.2) * 255 + 0.5, clamped, truncated (its defaults: scale 1, gamma 2.2)
void HdrToLdr(const std::vector<float>& rgb, std::vector<uint8_t>* out) {
  const float inv_gamma = 1.0f / 2.2f, scale = 1.0f;
  out->resize(rgb.size());
  for (size_t i = 0; i < rgb.size(); i++) {
    float z = float(std::pow(double(rgb[i] * scale), double(inv_gamma))) * 255 + 0.5f;
    if (z < 0) z = 0;
    if (z > 255) z = 255;
    (*out)[i] = uint8_t(int(z));
  }
}

}  // namespace pbio
