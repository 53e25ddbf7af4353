// image_codec.cpp -- see image_codec.h
#include "image_codec.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <sstream>

namespace pbio {
namespace {

// ---------------------------------------------------------------- checksums
uint32_t crc32_update(uint32_t crc, const uint8_t* p, size_t n) {
  static uint32_t table[256];
  static bool ready = false;
  if (!ready) {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
      table[i] = c;
    }
    ready = true;
  }
  crc = ~crc;
  for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xFF] ^ (crc >> 8);
  return ~crc;
}

uint32_t adler32(const uint8_t* p, size_t n) {
  uint32_t a = 1, b = 0;
  while (n) {
    const size_t k = n < 5552 ? n : 5552;
    for (size_t i = 0; i < k; ++i) a += p[i], b += a;
    a %= 65521u, b %= 65521u;
    p += k, n -= k;
  }
  return (b << 16) | a;
}

// ---------------------------------------------------------------- inflate (RFC 1951)
const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

struct BitReader {
  const uint8_t* p;
  size_t n, pos = 0;
  uint32_t acc = 0;
  int nbits = 0;
  bool overrun = false;
  BitReader(const uint8_t* s, size_t len) : p(s), n(len) {}
  uint32_t bits(int k) {
    while (nbits < k) {
      uint32_t byte = 0;
      if (pos < n) byte = p[pos++]; else overrun = true;
      acc |= byte << nbits;
      nbits += 8;
    }
    const uint32_t v = acc & ((k == 32) ? 0xFFFFFFFFu : ((1u << k) - 1u));
    acc >>= k, nbits -= k;
    return v;
  }
  void align() { acc = 0, nbits = 0; }
};

// canonical Huffman decoding table: count per length + symbols sorted by (length, value)
struct Huffman {
  uint16_t count[16];
  uint16_t symbol[288];
  bool build(const uint8_t* lengths, int n) {
    std::fill(count, count + 16, 0);
    for (int i = 0; i < n; ++i) count[lengths[i]]++;
    count[0] = 0;
    int left = 1;
    for (int len = 1; len < 16; ++len) {
      left <<= 1;
      left -= count[len];
      if (left < 0) return false;  // over-subscribed
    }
    uint16_t offs[16];
    offs[1] = 0;
    for (int len = 1; len < 15; ++len) offs[len + 1] = uint16_t(offs[len] + count[len]);
    for (int i = 0; i < n; ++i)
      if (lengths[i]) symbol[offs[lengths[i]]++] = uint16_t(i);
    return true;
  }
  int decode(BitReader* br) const {
    int code = 0, first = 0, index = 0;
    for (int len = 1; len < 16; ++len) {
      code |= int(br->bits(1));
      const int c = count[len];
      if (code - c < first) return symbol[index + (code - first)];
      index += c, first += c;
      first <<= 1, code <<= 1;
      if (br->overrun) return -1;
    }
    return -1;
  }
};

bool inflate_codes(BitReader* br, const Huffman& lit, const Huffman& dist, std::vector<uint8_t>* out, std::string* err) {
  for (;;) {
    const int sym = lit.decode(br);
    if (sym < 0) return *err = "bad huffman code", false;
    if (sym < 256) {
      out->push_back(uint8_t(sym));
    } else if (sym == 256) {
      return true;
    } else {
      if (sym > 285) return *err = "bad length code", false;
      const size_t len = kLenBase[sym - 257] + br->bits(kLenExtra[sym - 257]);
      const int ds = dist.decode(br);
      if (ds < 0 || ds > 29) return *err = "bad distance code", false;
      const size_t d = kDistBase[ds] + br->bits(kDistExtra[ds]);
      if (d > out->size()) return *err = "distance too far back", false;
      const size_t from = out->size() - d;
      for (size_t i = 0; i < len; ++i) out->push_back((*out)[from + i]);
    }
    if (br->overrun) return *err = "truncated stream", false;
  }
}

bool inflate_raw(BitReader* br, std::vector<uint8_t>* out, std::string* err) {
  for (;;) {
    const uint32_t final_block = br->bits(1), type = br->bits(2);
    if (type == 0) {
      br->align();
      if (br->pos + 4 > br->n) return *err = "truncated stored block", false;
      const uint32_t len = br->p[br->pos] | (br->p[br->pos + 1] << 8), nlen = br->p[br->pos + 2] | (br->p[br->pos + 3] << 8);
      br->pos += 4;
      if ((len ^ 0xFFFFu) != nlen) return *err = "stored block length mismatch", false;
      if (br->pos + len > br->n) return *err = "truncated stored block", false;
      out->insert(out->end(), br->p + br->pos, br->p + br->pos + len);
      br->pos += len;
    } else if (type == 1) {
      uint8_t l[288];
      for (int i = 0; i < 144; ++i) l[i] = 8;
      for (int i = 144; i < 256; ++i) l[i] = 9;
      for (int i = 256; i < 280; ++i) l[i] = 7;
      for (int i = 280; i < 288; ++i) l[i] = 8;
      uint8_t d[30];
      std::fill(d, d + 30, 5);
      Huffman lit, dist;
      lit.build(l, 288), dist.build(d, 30);
      if (!inflate_codes(br, lit, dist, out, err)) return false;
    } else if (type == 2) {
      const int nlen = int(br->bits(5)) + 257, ndist = int(br->bits(5)) + 1, ncode = int(br->bits(4)) + 4;
      if (nlen > 286 || ndist > 30) return *err = "bad code counts", false;
      static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
      uint8_t lengths[320];
      std::fill(lengths, lengths + 320, 0);
      for (int i = 0; i < ncode; ++i) lengths[order[i]] = uint8_t(br->bits(3));
      Huffman cl;
      if (!cl.build(lengths, 19)) return *err = "bad code-length code", false;
      uint8_t ll[320];
      int i = 0;
      while (i < nlen + ndist) {
        const int sym = cl.decode(br);
        if (sym < 0) return *err = "bad code-length symbol", false;
        if (sym < 16) {
          ll[i++] = uint8_t(sym);
        } else {
          uint8_t prev = 0;
          int rep;
          if (sym == 16) {
            if (i == 0) return *err = "repeat without a previous length", false;
            prev = ll[i - 1];
            rep = 3 + int(br->bits(2));
          } else if (sym == 17) {
            rep = 3 + int(br->bits(3));
          } else {
            rep = 11 + int(br->bits(7));
          }
          if (i + rep > nlen + ndist) return *err = "too many code lengths", false;
          while (rep--) ll[i++] = prev;
        }
      }
      if (ll[256] == 0) return *err = "no end-of-block code", false;
      Huffman lit, dist;
      if (!lit.build(ll, nlen)) return *err = "bad literal/length code", false;
      dist.build(ll + nlen, ndist);  // an incomplete distance code is legal
      if (!inflate_codes(br, lit, dist, out, err)) return false;
    } else {
      return *err = "reserved block type", false;
    }
    if (br->overrun) return *err = "truncated stream", false;
    if (final_block) return true;
  }
}

// ---------------------------------------------------------------- deflate: LZ77 (hash chains) + fixed Huffman codes
struct BitWriter {
  std::vector<uint8_t>* out;
  uint32_t acc = 0;
  int nbits = 0;
  void put(uint32_t v, int k) {  // LSB first
    acc |= v << nbits;
    nbits += k;
    while (nbits >= 8) out->push_back(uint8_t(acc)), acc >>= 8, nbits -= 8;
  }
  void put_code(uint32_t code, int k) {  // Huffman codes go MSB first
    uint32_t r = 0;
    for (int i = 0; i < k; ++i) r |= ((code >> i) & 1u) << (k - 1 - i);
    put(r, k);
  }
  void flush() {
    if (nbits) out->push_back(uint8_t(acc)), acc = 0, nbits = 0;
  }
};

void put_literal(BitWriter* bw, int sym) {  // fixed code of RFC 1951 §3.2.6
  if (sym < 144) bw->put_code(0x30 + sym, 8);
  else if (sym < 256) bw->put_code(0x190 + (sym - 144), 9);
  else if (sym < 280) bw->put_code(sym - 256, 7);
  else bw->put_code(0xC0 + (sym - 280), 8);
}

}  // namespace

bool ZlibInflate(const uint8_t* src, size_t n, std::vector<uint8_t>* out, std::string* err) {
  out->clear();
  if (n < 6) return *err = "zlib stream too short", false;
  if ((src[0] & 0x0F) != 8 || ((src[0] << 8) | src[1]) % 31 != 0) return *err = "bad zlib header", false;
  if (src[1] & 0x20) return *err = "preset dictionary not supported", false;
  BitReader br(src + 2, n - 2);
  return inflate_raw(&br, out, err);  // (like stb_image, the Adler-32 trailer is not verified)
}

void ZlibDeflate(const uint8_t* src, size_t n, std::vector<uint8_t>* out) {
  out->clear();
  out->push_back(0x78), out->push_back(0x5E);
  BitWriter bw{out};
  bw.put(1, 1), bw.put(1, 2);  // one final block, fixed codes
  const int kHashBits = 15, kChain = 32;
  std::vector<int32_t> head(size_t(1) << kHashBits, -1), prev(n ? n : 1, -1);
  auto hash = [&](size_t i) { return ((uint32_t(src[i]) << 10) ^ (uint32_t(src[i + 1]) << 5) ^ src[i + 2]) & ((1u << kHashBits) - 1u); };
  size_t i = 0;
  while (i < n) {
    size_t best_len = 0, best_dist = 0;
    if (i + 2 < n) {
      const uint32_t h = hash(i);
      int32_t cand = head[h];
      int chain = kChain;
      const size_t maxlen = std::min<size_t>(258, n - i);
      while (cand >= 0 && chain-- && i - size_t(cand) <= 32768) {
        size_t l = 0;
        while (l < maxlen && src[size_t(cand) + l] == src[i + l]) ++l;
        if (l > best_len) best_len = l, best_dist = i - size_t(cand);
        if (l == maxlen) break;
        cand = prev[size_t(cand)];
      }
    }
    const size_t step = best_len >= 3 ? best_len : 1;
    if (best_len >= 3) {
      int ls = 28;
      while (kLenBase[ls] > best_len) --ls;
      put_literal(&bw, 257 + ls);
      bw.put(uint32_t(best_len - kLenBase[ls]), kLenExtra[ls]);
      int ds = 29;
      while (kDistBase[ds] > best_dist) --ds;
      bw.put_code(uint32_t(ds), 5);
      bw.put(uint32_t(best_dist - kDistBase[ds]), kDistExtra[ds]);
    } else {
      put_literal(&bw, src[i]);
    }
    for (size_t k = 0; k < step; ++k, ++i) {
      if (i + 2 < n) {
        const uint32_t h = hash(i);
        prev[i] = head[h];
        head[h] = int32_t(i);
      }
    }
  }
  put_literal(&bw, 256);
  bw.flush();
  const uint32_t ad = adler32(src, n);
  out->push_back(uint8_t(ad >> 24)), out->push_back(uint8_t(ad >> 16)), out->push_back(uint8_t(ad >> 8)), out->push_back(uint8_t(ad));
}

// ---------------------------------------------------------------- PNG
namespace {
const uint8_t kPngSig[8] = {137, 80, 78, 71, 13, 10, 26, 10};

void put_chunk(std::vector<uint8_t>* f, const char* type, const uint8_t* data, size_t n) {
  const uint32_t len = uint32_t(n);
  f->push_back(uint8_t(len >> 24)), f->push_back(uint8_t(len >> 16)), f->push_back(uint8_t(len >> 8)), f->push_back(uint8_t(len));
  const size_t start = f->size();
  f->insert(f->end(), type, type + 4);
  if (n) f->insert(f->end(), data, data + n);
  const uint32_t crc = crc32_update(0, f->data() + start, n + 4);
  f->push_back(uint8_t(crc >> 24)), f->push_back(uint8_t(crc >> 16)), f->push_back(uint8_t(crc >> 8)), f->push_back(uint8_t(crc));
}

inline int paeth(int a, int b, int c) {
  const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
  if (pa <= pb && pa <= pc) return a;
  if (pb <= pc) return b;
  return c;
}

uint32_t be32(const uint8_t* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }

// undo the row filters of one (sub-)image in place; returns false on a bad filter byte
bool unfilter(uint8_t* data, size_t rows, size_t row_bytes, size_t bpp) {
  const size_t stride = row_bytes + 1;
  for (size_t y = 0; y < rows; ++y) {
    uint8_t* cur = data + y * stride + 1;
    const uint8_t* up = y ? data + (y - 1) * stride + 1 : nullptr;
    const int f = data[y * stride];
    if (f > 4) return false;
    for (size_t x = 0; x < row_bytes; ++x) {
      const int a = x >= bpp ? cur[x - bpp] : 0, b = up ? up[x] : 0, c = (up && x >= bpp) ? up[x - bpp] : 0;
      int add = 0;
      switch (f) {
        case 1: add = a; break;
        case 2: add = b; break;
        case 3: add = (a + b) >> 1; break;
        case 4: add = paeth(a, b, c); break;
      }
      cur[x] = uint8_t(cur[x] + add);
    }
  }
  return true;
}
}  // namespace

bool EncodePng(const uint8_t* px, size_t w, size_t h, size_t ch, std::vector<uint8_t>* file) {
  if (ch < 1 || ch > 4 || w == 0 || h == 0) return false;
  static const uint8_t ctype[5] = {0, 0, 4, 2, 6};
  const size_t rb = w * ch;
  std::vector<uint8_t> raw((rb + 1) * h), cand(rb);
  for (size_t y = 0; y < h; ++y) {  // per row: the filter with the smallest sum of |residual|
    const uint8_t* cur = px + y * rb;
    const uint8_t* up = y ? px + (y - 1) * rb : nullptr;
    long best = -1;
    for (int f = 0; f < 5; ++f) {
      long sum = 0;
      for (size_t x = 0; x < rb; ++x) {
        const int a = x >= ch ? cur[x - ch] : 0, b = up ? up[x] : 0, c = (up && x >= ch) ? up[x - ch] : 0;
        int pred = 0;
        switch (f) {
          case 1: pred = a; break;
          case 2: pred = b; break;
          case 3: pred = (a + b) >> 1; break;
          case 4: pred = paeth(a, b, c); break;
        }
        cand[x] = uint8_t(cur[x] - pred);
        sum += std::abs(int(int8_t(cand[x])));
      }
      if (best < 0 || sum < best) {
        best = sum;
        raw[y * (rb + 1)] = uint8_t(f);
        std::copy(cand.begin(), cand.end(), raw.begin() + long(y * (rb + 1) + 1));
      }
    }
  }
  std::vector<uint8_t> z;
  ZlibDeflate(raw.data(), raw.size(), &z);
  file->assign(kPngSig, kPngSig + 8);
  uint8_t ihdr[13] = {uint8_t(w >> 24), uint8_t(w >> 16), uint8_t(w >> 8), uint8_t(w), uint8_t(h >> 24), uint8_t(h >> 16),
                      uint8_t(h >> 8),  uint8_t(h),       8,               ctype[ch],  0,               0,
                      0};
  put_chunk(file, "IHDR", ihdr, 13);
  put_chunk(file, "IDAT", z.data(), z.size());
  put_chunk(file, "IEND", nullptr, 0);
  return true;
}

bool DecodePng(const uint8_t* file, size_t n, std::vector<uint8_t>* pixels, size_t* width, size_t* height,
               size_t* channels, std::string* err) {
  if (n < 8 || memcmp(file, kPngSig, 8) != 0) return *err = "not a PNG file", false;
  size_t pos = 8;
  uint32_t w = 0, h = 0;
  int depth = 0, color = 0, interlace = 0;
  bool have_ihdr = false, have_trns = false;
  uint8_t palette[256 * 4];
  size_t pal_len = 0;
  uint16_t key[3] = {0, 0, 0};
  std::vector<uint8_t> idat;
  for (;;) {
    if (pos + 8 > n) return *err = "truncated PNG", false;
    const uint32_t len = be32(file + pos);
    const uint8_t* type = file + pos + 4;
    const uint8_t* data = file + pos + 8;
    if (pos + 12 + size_t(len) > n) return *err = "truncated PNG chunk", false;
    if (!have_ihdr && memcmp(type, "IHDR", 4) != 0) return *err = "first chunk is not IHDR", false;
    if (memcmp(type, "IHDR", 4) == 0) {
      if (len != 13) return *err = "bad IHDR", false;
      w = be32(data), h = be32(data + 4);
      depth = data[8], color = data[9], interlace = data[12];
      if (w == 0 || h == 0 || w > (1u << 24) || h > (1u << 24) || uint64_t(w) * h > (1ull << 28)) return *err = "bad PNG size", false;
      if (depth != 1 && depth != 2 && depth != 4 && depth != 8 && depth != 16) return *err = "bad PNG bit depth", false;
      if (color > 6 || color == 1 || color == 5) return *err = "bad PNG colour type", false;
      if (color == 3 && depth == 16) return *err = "bad PNG colour type", false;
      if ((color == 2 || color == 4 || color == 6) && depth < 8) return *err = "bad PNG colour type", false;
      if (data[10] || data[11] || interlace > 1) return *err = "bad PNG compression/filter/interlace method", false;
      have_ihdr = true;
    } else if (memcmp(type, "PLTE", 4) == 0) {
      if (len > 256 * 3 || len % 3) return *err = "bad PLTE", false;
      pal_len = len / 3;
      for (size_t i = 0; i < pal_len; ++i) {
        palette[i * 4 + 0] = data[i * 3 + 0], palette[i * 4 + 1] = data[i * 3 + 1], palette[i * 4 + 2] = data[i * 3 + 2];
        palette[i * 4 + 3] = 255;
      }
    } else if (memcmp(type, "tRNS", 4) == 0) {
      if (!idat.empty()) return *err = "tRNS after IDAT", false;
      if (color == 3) {
        if (pal_len == 0 || len > pal_len) return *err = "bad tRNS", false;
        for (size_t i = 0; i < len; ++i) palette[i * 4 + 3] = data[i];
      } else {
        const size_t nc = (color & 2) ? 3 : 1;
        if ((color & 4) || len != nc * 2) return *err = "bad tRNS", false;
        for (size_t k = 0; k < nc; ++k) key[k] = uint16_t((data[2 * k] << 8) | data[2 * k + 1]);
      }
      have_trns = true;
    } else if (memcmp(type, "IDAT", 4) == 0) {
      if (color == 3 && pal_len == 0) return *err = "no PLTE", false;
      idat.insert(idat.end(), data, data + len);
    } else if (memcmp(type, "IEND", 4) == 0) {
      break;
    } else if (!(type[0] & 32)) {
      return *err = "unknown critical PNG chunk", false;
    }
    pos += 12 + size_t(len);
  }
  if (idat.empty()) return *err = "no IDAT", false;
  std::vector<uint8_t> raw;
  if (!ZlibInflate(idat.data(), idat.size(), &raw, err)) return false;

  const size_t file_n = (color == 3) ? 1 : ((color & 2) ? 3 : 1) + ((color & 4) ? 1 : 0);  // samples per pixel as stored
  const size_t bits_pp = file_n * size_t(depth);
  const size_t bpp = (bits_pp + 7) / 8;
  // samples of the whole image, 16 bits each (value as stored; sub-byte values unscaled)
  std::vector<uint16_t> samp(size_t(w) * h * file_n);
  size_t rpos = 0;
  static const int xo[7] = {0, 4, 0, 2, 0, 1, 0}, yo[7] = {0, 0, 4, 0, 2, 0, 1}, xs[7] = {8, 8, 4, 4, 2, 2, 1}, ys[7] = {8, 8, 8, 4, 4, 2, 2};
  const int passes = interlace ? 7 : 1;
  for (int p = 0; p < passes; ++p) {
    const size_t x0 = interlace ? xo[p] : 0, y0 = interlace ? yo[p] : 0, dx = interlace ? xs[p] : 1, dy = interlace ? ys[p] : 1;
    const size_t pw = (w + dx - 1 - x0) / dx, ph = (h + dy - 1 - y0) / dy;
    if (x0 >= w || y0 >= h || pw == 0 || ph == 0) continue;
    const size_t row_bytes = (pw * bits_pp + 7) / 8;
    if (rpos + (row_bytes + 1) * ph > raw.size()) return *err = "not enough pixel data", false;
    if (!unfilter(raw.data() + rpos, ph, row_bytes, bpp)) return *err = "bad PNG row filter", false;
    for (size_t y = 0; y < ph; ++y) {
      const uint8_t* row = raw.data() + rpos + y * (row_bytes + 1) + 1;
      for (size_t x = 0; x < pw; ++x) {
        uint16_t* dst = &samp[((y0 + y * dy) * w + (x0 + x * dx)) * file_n];
        for (size_t c = 0; c < file_n; ++c) {
          const size_t s = x * file_n + c;
          if (depth == 16) dst[c] = uint16_t((row[2 * s] << 8) | row[2 * s + 1]);
          else if (depth == 8) dst[c] = row[s];
          else {
            const size_t bit = s * size_t(depth);
            dst[c] = uint16_t((row[bit >> 3] >> (8 - depth - int(bit & 7))) & ((1 << depth) - 1));
          }
        }
      }
    }
    rpos += (row_bytes + 1) * ph;
  }

  // -> 8-bit output the way stb_image delivers it
  static const int scale_tab[9] = {0, 0xff, 0x55, 0, 0x11, 0, 0, 0, 0x01};
  const size_t npx = size_t(w) * h;
  size_t out_n;
  if (color == 3) {
    out_n = have_trns ? 4 : 3;
    pixels->resize(npx * out_n);
    for (size_t i = 0; i < npx; ++i) {
      const size_t idx = samp[i];  // an index beyond the palette reads zeros/255 like an untouched entry would not: reject
      if (idx >= 256) return *err = "bad palette index", false;
      const uint8_t* e = palette + idx * 4;
      uint8_t zero[4] = {0, 0, 0, 255};
      if (idx >= pal_len) e = zero;
      for (size_t c = 0; c < out_n; ++c) (*pixels)[i * out_n + c] = e[c];
    }
  } else {
    out_n = file_n + (have_trns ? 1 : 0);
    pixels->resize(npx * out_n);
    const int sc = depth < 8 ? scale_tab[depth] : 1;
    uint16_t k8[3] = {0, 0, 0};
    if (have_trns && depth < 16)
      for (int k = 0; k < 3; ++k) k8[k] = uint16_t((key[k] & 255) * (depth < 8 ? scale_tab[depth] : 1)) & 255;
    for (size_t i = 0; i < npx; ++i) {
      const uint16_t* s = &samp[i * file_n];
      uint8_t* d = &(*pixels)[i * out_n];
      bool match = have_trns;
      for (size_t c = 0; c < file_n; ++c) {
        if (depth == 16) {
          d[c] = uint8_t(s[c] >> 8);
          if (have_trns && s[c] != key[c]) match = false;
        } else {
          d[c] = uint8_t(s[c] * sc);
          if (have_trns && d[c] != uint8_t(k8[c])) match = false;
        }
      }
      if (have_trns) d[file_n] = match ? 0 : 255;
    }
  }
  *width = w, *height = h, *channels = out_n;
  return true;
}

bool DecodeHdr(const uint8_t* file, size_t n, std::vector<float>* pixels, size_t* width, size_t* height, std::string* err) {
  size_t pos = 0;
  auto getline = [&](std::string* s) {  // stbi__hdr_gettoken: up to '\n' (at most 1023 characters are kept)
    s->clear();
    while (pos < n && file[pos] != '\n') {
      if (s->size() < 1023) s->push_back(char(file[pos]));
      ++pos;
    }
    if (pos < n) ++pos;
  };
  auto get8 = [&]() -> int { return pos < n ? file[pos++] : 0; };
  std::string tok;
  getline(&tok);
  if (tok != "#?RADIANCE" && tok != "#?RGBE") return *err = "not a Radiance HDR file", false;
  bool valid = false;
  for (;;) {
    getline(&tok);
    if (tok.empty()) break;
    if (tok == "FORMAT=32-bit_rle_rgbe") valid = true;
    if (pos >= n) break;
  }
  if (!valid) return *err = "unsupported HDR format", false;
  getline(&tok);
  if (tok.compare(0, 3, "-Y ") != 0) return *err = "unsupported HDR data layout", false;
  char* rest = nullptr;
  const long hh = strtol(tok.c_str() + 3, &rest, 10);
  while (*rest == ' ') ++rest;
  if (strncmp(rest, "+X ", 3) != 0) return *err = "unsupported HDR data layout", false;
  const long ww = strtol(rest + 3, nullptr, 10);
  if (ww <= 0 || hh <= 0 || ww > (1 << 24) || hh > (1 << 24) || uint64_t(ww) * uint64_t(hh) > (1ull << 28)) return *err = "bad HDR size", false;
  const size_t w = size_t(ww), h = size_t(hh);
  pixels->assign(w * h * 3, 0.0f);
  auto convert = [&](float* out, const uint8_t* in) {  // stbi__hdr_convert, 3 channels
    if (in[3] != 0) {
      const float f1 = float(ldexp(1.0f, int(in[3]) - (128 + 8)));
      out[0] = in[0] * f1, out[1] = in[1] * f1, out[2] = in[2] * f1;
    } else {
      out[0] = out[1] = out[2] = 0.0f;
    }
  };
  auto flat_from = [&](size_t first) {
    for (size_t i = first; i < w * h; ++i) {
      uint8_t rgbe[4];
      for (int k = 0; k < 4; ++k) rgbe[k] = uint8_t(get8());
      convert(&(*pixels)[i * 3], rgbe);
    }
  };
  if (w < 8 || w >= 32768) {
    flat_from(0);
  } else {
    std::vector<uint8_t> scan(w * 4);
    for (size_t j = 0; j < h; ++j) {
      const int c1 = get8(), c2 = get8();
      int len = get8();
      if (c1 != 2 || c2 != 2 || (len & 0x80)) {  // not run-length encoded: these 4 bytes are pixel 0 of a flat file
        uint8_t rgbe[4] = {uint8_t(c1), uint8_t(c2), uint8_t(len), uint8_t(get8())};
        convert(&(*pixels)[0], rgbe);
        flat_from(1);
        break;
      }
      len = (len << 8) | get8();
      if (size_t(len) != w) return *err = "corrupt HDR: bad scanline length", false;
      for (int k = 0; k < 4; ++k) {
        size_t i = 0;
        while (i < w) {
          int count = get8();
          const size_t left = w - i;
          if (count > 128) {
            const uint8_t value = uint8_t(get8());
            count -= 128;
            if (size_t(count) > left) return *err = "corrupt HDR: bad RLE data", false;
            for (int z = 0; z < count; ++z) scan[i++ * 4 + size_t(k)] = value;
          } else {
            if (size_t(count) > left) return *err = "corrupt HDR: bad RLE data", false;
            if (count == 0 && pos >= n) return *err = "truncated HDR", false;
            for (int z = 0; z < count; ++z) scan[i++ * 4 + size_t(k)] = uint8_t(get8());
          }
        }
      }
      for (size_t i = 0; i < w; ++i) convert(&(*pixels)[(j * w + i) * 3], &scan[i * 4]);
    }
  }
  *width = w, *height = h;
  return true;
}

// ---------------------------------------------------------------- JPEG (baseline / extended sequential Huffman)
// Decodes to what stb_image returns for the same file (stbi_load, req_comp = 0): its integer IDCT, its "3:1" upsampling
// filters across block boundaries, its fixed-point YCbCr -> RGB, CMYK / YCCK through the Adobe transform flag.
namespace {
const uint8_t kZigzag[64 + 15] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20,
                                  13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45,
                                  38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

struct JpegHuff {
  uint8_t size[257];
  uint16_t code[256];
  uint8_t values[256];
  uint32_t maxcode[18];
  int delta[17];
  JpegHuff() {  // a table the file never defines decodes nothing (every lookup ends at the sentinel)
    memset(size, 0, sizeof(size)), memset(code, 0, sizeof(code)), memset(values, 0, sizeof(values));
    memset(maxcode, 0, sizeof(maxcode)), memset(delta, 0, sizeof(delta));
    maxcode[17] = 0xffffffffu;
  }
  bool build(const int* count) {  // JPEG spec C.2 (stbi__build_huffman)
    int k = 0;
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < count[i]; ++j) {
        if (k >= 256) return false;
        size[k++] = uint8_t(i + 1);
      }
    size[k] = 0;
    uint32_t c = 0;
    k = 0;
    int j;
    for (j = 1; j <= 16; ++j) {
      delta[j] = k - int(c);
      if (size[k] == j) {
        while (size[k] == j) code[k++] = uint16_t(c++);
        if (c - 1 >= (1u << j)) return false;
      }
      maxcode[j] = c << (16 - j);
      c <<= 1;
    }
    maxcode[j] = 0xffffffffu;
    return true;
  }
};

struct JpegComp {
  int id = 0, h = 0, v = 0, tq = 0, hd = 0, ha = 0, dc_pred = 0;
  int x = 0, y = 0, w2 = 0, h2 = 0;
  std::vector<uint8_t> data;
  std::vector<short> coeff;  // progressive: 64 coefficients per block, (w2 / 8) blocks per row
  int coeff_w = 0;
};

struct JpegDecoder {
  const uint8_t* p;
  size_t n, pos = 0;
  uint32_t code_buffer = 0;
  int code_bits = 0, nomore = 0;
  int marker = 0xff;  // 0xff = none
  int restart_interval = 0, todo = 0;
  JpegHuff huff_dc[4], huff_ac[4];
  uint16_t dequant[4][64];
  JpegComp comp[4];
  int img_x = 0, img_y = 0, img_n = 0, h_max = 1, v_max = 1, mcu_x = 0, mcu_y = 0;
  int scan_n = 0, order[4] = {0, 0, 0, 0};
  int rgb = 0, jfif = 0, app14 = -1;
  bool progressive = false;
  int spec_start = 0, spec_end = 63, succ_high = 0, succ_low = 0, eob_run = 0;  // progressive scan parameters
  std::string err;

  int get8() { return pos < n ? p[pos++] : 0; }
  int get16() {
    const int a = get8();
    return (a << 8) | get8();
  }
  bool eof() const { return pos >= n; }
  bool fail(const char* m) {
    err = m;
    return false;
  }
  void grow() {  // stbi__grow_buffer_unsafe: byte stuffing, a marker ends the entropy data (zero bits follow)
    do {
      uint32_t b = nomore ? 0u : uint32_t(get8());
      if (b == 0xff) {
        int c = get8();
        while (c == 0xff) c = get8();
        if (c != 0) {
          marker = c;
          nomore = 1;
          return;
        }
      }
      code_buffer |= b << (24 - code_bits);
      code_bits += 8;
    } while (code_bits <= 24);
  }
  int huff_decode(const JpegHuff& h) {
    if (code_bits < 16) grow();
    const uint32_t temp = code_buffer >> 16;
    int k;
    for (k = 1;; ++k)
      if (temp < h.maxcode[k]) break;
    if (k == 17) {
      code_bits -= 16;
      return -1;
    }
    if (k > code_bits) return -1;
    const int c = int((code_buffer >> (32 - k)) & ((1u << k) - 1u)) + h.delta[k];
    if (c < 0 || c > 255) return -1;
    code_bits -= k;
    code_buffer <<= k;
    return h.values[c];
  }
  int extend_receive(int nb) {
    if (code_bits < nb) grow();
    const int sgn = int32_t(code_buffer) >> 31;
    uint32_t k = (code_buffer << nb) | (code_buffer >> (32 - nb));
    const uint32_t mask = (1u << nb) - 1u;
    code_buffer = k & ~mask;
    k &= mask;
    code_bits -= nb;
    static const int bias[16] = {0, -1, -3, -7, -15, -31, -63, -127, -255, -511, -1023, -2047, -4095, -8191, -16383, -32767};
    return int(k) + (bias[nb] & ~sgn);
  }
  bool decode_block(short data[64], int b) {
    JpegComp& c = comp[b];
    const uint16_t* dq = dequant[c.tq];
    if (code_bits < 16) grow();
    const int t = huff_decode(huff_dc[c.hd]);
    if (t < 0 || t > 15) return fail("bad huffman code in JPEG");
    memset(data, 0, 64 * sizeof(short));
    const int diff = t ? extend_receive(t) : 0;
    const int dc = c.dc_pred + diff;
    c.dc_pred = dc;
    data[0] = short(dc * dq[0]);
    int k = 1;
    do {
      if (code_bits < 16) grow();
      const int rs = huff_decode(huff_ac[c.ha]);
      if (rs < 0) return fail("bad huffman code in JPEG");
      const int sbits = rs & 15, r = rs >> 4;
      if (sbits == 0) {
        if (rs != 0xf0) break;
        k += 16;
      } else {
        k += r;
        const unsigned zig = kZigzag[k++];
        data[zig] = short(extend_receive(sbits) * dq[zig]);
      }
    } while (k < 64);
    return true;
  }
  int get_bits(int nb) {
    if (code_bits < nb) grow();
    uint32_t k = (code_buffer << nb) | (code_buffer >> (32 - nb));
    const uint32_t mask = (1u << nb) - 1u;
    code_buffer = k & ~mask;
    code_bits -= nb;
    return int(k & mask);
  }
  bool get_bit() {
    if (code_bits < 1) grow();
    const uint32_t k = code_buffer;
    code_buffer <<= 1;
    --code_bits;
    return (k & 0x80000000u) != 0;
  }
  // Progressive scans (ITU T.81 annex G): one spectral band / one bit plane of the coefficients per scan.
  // DC: first pass codes the difference of (value >> Al); refinement passes add one bit each
  bool prog_dc(short* data, int b) {
    if (spec_end != 0) return fail("progressive JPEG: DC and AC in one scan");
    JpegComp& c = comp[b];
    if (code_bits < 16) grow();
    if (succ_high == 0) {
      memset(data, 0, 64 * sizeof(short));
      const int t = huff_decode(huff_dc[c.hd]);
      if (t < 0 || t > 15) return fail("bad huffman code in JPEG");
      const int diff = t ? extend_receive(t) : 0;
      const int dc = c.dc_pred + diff;
      c.dc_pred = dc;
      data[0] = short(uint32_t(dc) << succ_low);
    } else if (get_bit()) {
      data[0] = short(data[0] + short(1 << succ_low));
    }
    return true;
  }
  // AC: first pass codes run/size pairs with end-of-band runs spanning blocks; refinement passes append one bit to every
  // coefficient that is already non-zero while placing new +-1 coefficients
  bool prog_ac(short* data, int b) {
    if (spec_start == 0) return fail("progressive JPEG: DC and AC in one scan");
    const JpegHuff& hac = huff_ac[comp[b].ha];
    if (succ_high == 0) {
      if (eob_run) {
        --eob_run;
        return true;
      }
      int k = spec_start;
      do {
        if (code_bits < 16) grow();
        const int rs = huff_decode(hac);
        if (rs < 0) return fail("bad huffman code in JPEG");
        const int sbits = rs & 15, r = rs >> 4;
        if (sbits == 0) {
          if (r < 15) {
            eob_run = 1 << r;
            if (r) eob_run += get_bits(r);
            --eob_run;
            break;
          }
          k += 16;
        } else {
          k += r;
          const unsigned zig = kZigzag[k++];
          data[zig] = short(uint32_t(extend_receive(sbits)) << succ_low);
        }
      } while (k <= spec_end);
      return true;
    }
    const short bit = short(1 << succ_low);
    auto refine = [&](short* p) {  // one correction bit for a coefficient that is already non-zero
      if (get_bit() && (*p & bit) == 0) *p = short(*p > 0 ? *p + bit : *p - bit);
    };
    if (eob_run) {
      --eob_run;
      for (int k = spec_start; k <= spec_end; ++k) {
        short* p = &data[kZigzag[k]];
        if (*p != 0) refine(p);
      }
      return true;
    }
    int k = spec_start;
    do {
      const int rs = huff_decode(hac);
      if (rs < 0) return fail("bad huffman code in JPEG");
      int sbits = rs & 15, r = rs >> 4;
      if (sbits == 0) {
        if (r < 15) {
          eob_run = (1 << r) - 1;
          if (r) eob_run += get_bits(r);
          r = 64;  // to the end of the band
        }
      } else {
        if (sbits != 1) return fail("bad huffman code in JPEG");
        sbits = get_bit() ? bit : -bit;
      }
      while (k <= spec_end) {
        short* p = &data[kZigzag[k++]];
        if (*p != 0) {
          refine(p);
        } else {
          if (r == 0) {
            *p = short(sbits);
            break;
          }
          --r;
        }
      }
    } while (k <= spec_end);
    return true;
  }
  void reset() {
    code_bits = 0, code_buffer = 0, nomore = 0;
    for (JpegComp& c : comp) c.dc_pred = 0;
    marker = 0xff;
    todo = restart_interval ? restart_interval : 0x7fffffff;
    eob_run = 0;
  }
  int get_marker() {
    if (marker != 0xff) {
      const int x = marker;
      marker = 0xff;
      return x;
    }
    int x = get8();
    if (x != 0xff) return 0xff;
    while (x == 0xff) x = get8();
    return x;
  }
};

inline uint8_t clamp255(int x) { return uint8_t(unsigned(x) > 255u ? (x < 0 ? 0 : 255) : x); }

// jidctint-derived integer IDCT, 12-bit constants (stbi__idct_block)
void jpeg_idct(uint8_t* out, int stride, const short d[64]) {
  auto f2f = [](double x) { return int(x * 4096 + 0.5); };
  static const int c0 = f2f(0.5411961f), c1 = f2f(-1.847759065f), c2 = f2f(0.765366865f), c3 = f2f(1.175875602f),
                   c4 = f2f(0.298631336f), c5 = f2f(2.053119869f), c6 = f2f(3.072711026f), c7 = f2f(1.501321110f),
                   c8 = f2f(-0.899976223f), c9 = f2f(-2.562915447f), c10 = f2f(-1.961570560f), c11 = f2f(-0.390180644f);
  // 32-bit arithmetic that wraps instead of overflowing (identical on valid files; corrupt coefficients must not be UB)
  struct W {
    int v;
    W() : v(0) {}
    W(int x) : v(x) {}
    explicit operator int() const { return v; }
    W operator+(W o) const { return W(int(uint32_t(v) + uint32_t(o.v))); }
    W operator-(W o) const { return W(int(uint32_t(v) - uint32_t(o.v))); }
    W operator*(W o) const { return W(int(uint32_t(v) * uint32_t(o.v))); }
    W& operator+=(W o) { return *this = *this + o; }
  };
  int val[64];
  auto pass = [&](W s0, W s1, W s2, W s3, W s4, W s5, W s6, W s7, W& x0, W& x1, W& x2, W& x3, W& t0, W& t1, W& t2, W& t3) {
    W p1, p2, p3, p4, p5;
    p2 = s2, p3 = s6;
    p1 = (p2 + p3) * c0;
    t2 = p1 + p3 * c1;
    t3 = p1 + p2 * c2;
    p2 = s0, p3 = s4;
    t0 = (p2 + p3) * 4096;
    t1 = (p2 - p3) * 4096;
    x0 = t0 + t3, x3 = t0 - t3, x1 = t1 + t2, x2 = t1 - t2;
    t0 = s7, t1 = s5, t2 = s3, t3 = s1;
    p3 = t0 + t2, p4 = t1 + t3, p1 = t0 + t3, p2 = t1 + t2;
    p5 = (p3 + p4) * c3;
    t0 = t0 * c4, t1 = t1 * c5, t2 = t2 * c6, t3 = t3 * c7;
    p1 = p5 + p1 * c8, p2 = p5 + p2 * c9, p3 = p3 * c10, p4 = p4 * c11;
    t3 += p1 + p4, t2 += p2 + p3, t1 += p2 + p4, t0 += p1 + p3;
  };
  for (int i = 0; i < 8; ++i) {
    const short* c = d + i;
    int* v = val + i;
    if (c[8] == 0 && c[16] == 0 && c[24] == 0 && c[32] == 0 && c[40] == 0 && c[48] == 0 && c[56] == 0) {
      const int dc = c[0] * 4;
      v[0] = v[8] = v[16] = v[24] = v[32] = v[40] = v[48] = v[56] = dc;
    } else {
      W x0, x1, x2, x3, t0, t1, t2, t3;
      pass(c[0], c[8], c[16], c[24], c[32], c[40], c[48], c[56], x0, x1, x2, x3, t0, t1, t2, t3);
      x0 += 512, x1 += 512, x2 += 512, x3 += 512;
      v[0] = int(x0 + t3) >> 10, v[56] = int(x0 - t3) >> 10, v[8] = int(x1 + t2) >> 10, v[48] = int(x1 - t2) >> 10;
      v[16] = int(x2 + t1) >> 10, v[40] = int(x2 - t1) >> 10, v[24] = int(x3 + t0) >> 10, v[32] = int(x3 - t0) >> 10;
    }
  }
  for (int i = 0; i < 8; ++i) {
    const int* v = val + i * 8;
    uint8_t* o = out + i * stride;
    W x0, x1, x2, x3, t0, t1, t2, t3;
    pass(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], x0, x1, x2, x3, t0, t1, t2, t3);
    const int bias = 65536 + (128 << 17);
    x0 += bias, x1 += bias, x2 += bias, x3 += bias;
    o[0] = clamp255(int(x0 + t3) >> 17), o[7] = clamp255(int(x0 - t3) >> 17), o[1] = clamp255(int(x1 + t2) >> 17), o[6] = clamp255(int(x1 - t2) >> 17);
    o[2] = clamp255(int(x2 + t1) >> 17), o[5] = clamp255(int(x2 - t1) >> 17), o[3] = clamp255(int(x3 + t0) >> 17), o[4] = clamp255(int(x3 - t0) >> 17);
  }
}

// one output row of an upsampled component (stbi__resample_row_*): `near` / `far` are the two source rows
const uint8_t* jpeg_resample(uint8_t* out, const uint8_t* nr, const uint8_t* fr, int w, int hs, int vs) {
  if (hs == 1 && vs == 1) return nr;
  if (hs == 1 && vs == 2) {
    for (int i = 0; i < w; ++i) out[i] = uint8_t((3 * nr[i] + fr[i] + 2) >> 2);
    return out;
  }
  if (hs == 2 && vs == 1) {
    if (w == 1) {
      out[0] = out[1] = nr[0];
      return out;
    }
    out[0] = nr[0];
    out[1] = uint8_t((nr[0] * 3 + nr[1] + 2) >> 2);
    int i;
    for (i = 1; i < w - 1; ++i) {
      const int m = 3 * nr[i] + 2;
      out[i * 2 + 0] = uint8_t((m + nr[i - 1]) >> 2);
      out[i * 2 + 1] = uint8_t((m + nr[i + 1]) >> 2);
    }
    out[i * 2 + 0] = uint8_t((nr[w - 2] * 3 + nr[w - 1] + 2) >> 2);
    out[i * 2 + 1] = nr[w - 1];
    return out;
  }
  if (hs == 2 && vs == 2) {
    if (w == 1) {
      out[0] = out[1] = uint8_t((3 * nr[0] + fr[0] + 2) >> 2);
      return out;
    }
    int t1 = 3 * nr[0] + fr[0];
    out[0] = uint8_t((t1 + 2) >> 2);
    for (int i = 1; i < w; ++i) {
      const int t0 = t1;
      t1 = 3 * nr[i] + fr[i];
      out[i * 2 - 1] = uint8_t((3 * t0 + t1 + 8) >> 4);
      out[i * 2] = uint8_t((3 * t1 + t0 + 8) >> 4);
    }
    out[w * 2 - 1] = uint8_t((t1 + 2) >> 2);
    return out;
  }
  for (int i = 0; i < w; ++i)
    for (int j = 0; j < hs; ++j) out[i * hs + j] = nr[i];
  return out;
}

void jpeg_ycc_row(uint8_t* out, const uint8_t* y, const uint8_t* pcb, const uint8_t* pcr, int count, int step) {
  auto fx = [](float x) { return int(x * 4096.0f + 0.5f) << 8; };
  static const int k_r = fx(1.40200f), k_g1 = fx(0.71414f), k_g2 = fx(0.34414f), k_b = fx(1.77200f);
  for (int i = 0; i < count; ++i) {
    const int yf = (y[i] << 20) + (1 << 19);
    const int cr = pcr[i] - 128, cb = pcb[i] - 128;
    int r = yf + cr * k_r;
    int g = yf + (cr * -k_g1) + int(uint32_t(cb * -k_g2) & 0xffff0000u);
    int b = yf + cb * k_b;
    r >>= 20, g >>= 20, b >>= 20;
    out[0] = clamp255(r), out[1] = clamp255(g), out[2] = clamp255(b);
    out += step;
  }
}

inline uint8_t blinn8(uint8_t x, uint8_t y) {
  const unsigned t = unsigned(x) * y + 128;
  return uint8_t((t + (t >> 8)) >> 8);
}
}  // namespace

bool DecodeJpeg(const uint8_t* file, size_t n, std::vector<uint8_t>* pixels, size_t* width, size_t* height, size_t* channels,
                std::string* err) {
  std::unique_ptr<JpegDecoder> zp(new JpegDecoder());
  JpegDecoder& z = *zp;
  z.p = file, z.n = n;
  auto bail = [&](const char* m) {
    *err = m;
    return false;
  };
  if (z.get_marker() != 0xd8) return bail("not a JPEG file");
  auto process_marker = [&](int m) -> bool {
    switch (m) {
      case 0xff: return z.fail("expected a JPEG marker");
      case 0xdd:
        if (z.get16() != 4) return z.fail("bad DRI length");
        z.restart_interval = z.get16();
        return true;
      case 0xdb: {
        int L = z.get16() - 2;
        while (L > 0) {
          const int q = z.get8(), pq = q >> 4, t = q & 15;
          if (pq != 0 && pq != 1) return z.fail("bad DQT type");
          if (t > 3) return z.fail("bad DQT table");
          for (int i = 0; i < 64; ++i) z.dequant[t][kZigzag[i]] = uint16_t(pq ? z.get16() : z.get8());
          L -= pq ? 129 : 65;
        }
        return L == 0 ? true : z.fail("bad DQT length");
      }
      case 0xc4: {
        int L = z.get16() - 2;
        while (L > 0) {
          const int q = z.get8(), tc = q >> 4, th = q & 15;
          if (tc > 1 || th > 3) return z.fail("bad DHT header");
          int sizes[16], total = 0;
          for (int i = 0; i < 16; ++i) sizes[i] = z.get8(), total += sizes[i];
          if (total > 256) return z.fail("bad DHT table");
          L -= 17;
          JpegHuff& h = tc == 0 ? z.huff_dc[th] : z.huff_ac[th];
          if (!h.build(sizes)) return z.fail("bad code lengths in JPEG");
          for (int i = 0; i < total; ++i) h.values[i] = uint8_t(z.get8());
          L -= total;
        }
        return L == 0 ? true : z.fail("bad DHT length");
      }
    }
    if ((m >= 0xe0 && m <= 0xef) || m == 0xfe) {
      int L = z.get16();
      if (L < 2) return z.fail("bad APP / COM length");
      L -= 2;
      if (m == 0xe0 && L >= 5) {
        static const uint8_t tag[5] = {'J', 'F', 'I', 'F', 0};
        bool ok = true;
        for (int i = 0; i < 5; ++i)
          if (z.get8() != tag[i]) ok = false;
        L -= 5;
        if (ok) z.jfif = 1;
      } else if (m == 0xee && L >= 12) {
        static const uint8_t tag[6] = {'A', 'd', 'o', 'b', 'e', 0};
        bool ok = true;
        for (int i = 0; i < 6; ++i)
          if (z.get8() != tag[i]) ok = false;
        L -= 6;
        if (ok) {
          z.get8(), z.get16(), z.get16();
          z.app14 = z.get8();
          L -= 6;
        }
      }
      z.pos = std::min(z.n, z.pos + size_t(L));
      return true;
    }
    return z.fail("unknown JPEG marker");
  };
  // header: everything up to SOF
  int m = z.get_marker();
  while (!(m == 0xc0 || m == 0xc1 || m == 0xc2)) {
    if (!process_marker(m)) return bail(z.err.c_str());
    m = z.get_marker();
    while (m == 0xff) {
      if (z.eof()) return bail("no SOF in JPEG");
      m = z.get_marker();
    }
  }
  z.progressive = (m == 0xc2);
  {
    const int Lf = z.get16();
    if (Lf < 11) return bail("bad SOF length");
    if (z.get8() != 8) return bail("JPEG: only 8 bits per sample");
    z.img_y = z.get16(), z.img_x = z.get16();
    if (z.img_y == 0 || z.img_x == 0 || uint64_t(z.img_x) * uint64_t(z.img_y) > (1ull << 28)) return bail("bad JPEG size");
    const int c = z.get8();
    if (c != 3 && c != 1 && c != 4) return bail("bad JPEG component count");
    z.img_n = c;
    if (Lf != 8 + 3 * c) return bail("bad SOF length");
    for (int i = 0; i < c; ++i) {
      static const uint8_t rgb[3] = {'R', 'G', 'B'};
      JpegComp& k = z.comp[i];
      k.id = z.get8();
      if (c == 3 && k.id == rgb[i]) ++z.rgb;
      const int q = z.get8();
      k.h = q >> 4, k.v = q & 15;
      if (!k.h || k.h > 4 || !k.v || k.v > 4) return bail("bad JPEG sampling factors");
      k.tq = z.get8();
      if (k.tq > 3) return bail("bad JPEG quantisation table index");
    }
    for (int i = 0; i < c; ++i) z.h_max = std::max(z.h_max, z.comp[i].h), z.v_max = std::max(z.v_max, z.comp[i].v);
    const int mw = z.h_max * 8, mh = z.v_max * 8;
    z.mcu_x = (z.img_x + mw - 1) / mw, z.mcu_y = (z.img_y + mh - 1) / mh;
    for (int i = 0; i < c; ++i) {
      JpegComp& k = z.comp[i];
      k.x = (z.img_x * k.h + z.h_max - 1) / z.h_max;
      k.y = (z.img_y * k.v + z.v_max - 1) / z.v_max;
      k.w2 = z.mcu_x * k.h * 8, k.h2 = z.mcu_y * k.v * 8;
      k.data.assign(size_t(k.w2) * size_t(k.h2), 0);
      if (z.progressive) k.coeff_w = k.w2 / 8, k.coeff.assign(size_t(k.w2) * size_t(k.h2), 0);
    }
  }
  // scans
  m = z.get_marker();
  while (m != 0xd9) {
    if (m == 0xda) {
      const int Ls = z.get16();
      z.scan_n = z.get8();
      if (z.scan_n < 1 || z.scan_n > 4 || z.scan_n > z.img_n) return bail("bad SOS component count");
      if (Ls != 6 + 2 * z.scan_n) return bail("bad SOS length");
      for (int i = 0; i < z.scan_n; ++i) {
        const int id = z.get8(), q = z.get8();
        int which = 0;
        for (; which < z.img_n; ++which)
          if (z.comp[which].id == id) break;
        if (which == z.img_n) return bail("bad SOS component");
        z.comp[which].hd = q >> 4, z.comp[which].ha = q & 15;
        if (z.comp[which].hd > 3 || z.comp[which].ha > 3) return bail("bad SOS huffman table index");
        z.order[i] = which;
      }
      z.spec_start = z.get8(), z.spec_end = z.get8();
      const int aa = z.get8();
      z.succ_high = aa >> 4, z.succ_low = aa & 15;
      if (z.progressive) {
        if (z.spec_start > 63 || z.spec_end > 63 || z.spec_start > z.spec_end || z.succ_high > 13 || z.succ_low > 13) return bail("bad SOS");
      } else {
        if (z.spec_start != 0 || aa != 0) return bail("bad SOS");
        z.spec_end = 63;
      }
      // entropy-coded data
      z.reset();
      short data[64];
      bool stop = false;
      auto restart_or_stop = [&]() {  // after each MCU: at a restart interval's end a restart marker must follow
        if (--z.todo > 0) return false;
        if (z.code_bits < 24) z.grow();
        if (!(z.marker >= 0xd0 && z.marker <= 0xd7)) return true;  // not a restart: keep what was decoded
        z.reset();
        return false;
      };
      if (z.progressive && z.scan_n == 1) {
        const int c = z.order[0];
        JpegComp& k = z.comp[c];
        const int bw = (k.x + 7) >> 3, bh = (k.y + 7) >> 3;
        for (int j = 0; j < bh && !stop; ++j)
          for (int i = 0; i < bw; ++i) {
            short* blk = k.coeff.data() + 64 * (size_t(i) + size_t(j) * size_t(k.coeff_w));
            if (!(z.spec_start == 0 ? z.prog_dc(blk, c) : z.prog_ac(blk, c))) return bail(z.err.c_str());
            if (restart_or_stop()) {
              stop = true;
              break;
            }
          }
      } else if (z.progressive) {  // interleaved: DC only
        for (int j = 0; j < z.mcu_y && !stop; ++j)
          for (int i = 0; i < z.mcu_x; ++i) {
            for (int s = 0; s < z.scan_n; ++s) {
              const int c = z.order[s];
              JpegComp& k = z.comp[c];
              for (int y = 0; y < k.v; ++y)
                for (int x = 0; x < k.h; ++x) {
                  const size_t x2 = size_t(i * k.h + x), y2 = size_t(j * k.v + y);
                  if (!z.prog_dc(k.coeff.data() + 64 * (x2 + y2 * size_t(k.coeff_w)), c)) return bail(z.err.c_str());
                }
            }
            if (restart_or_stop()) {
              stop = true;
              break;
            }
          }
      } else if (z.scan_n == 1) {
        const int c = z.order[0];
        JpegComp& k = z.comp[c];
        const int bw = (k.x + 7) >> 3, bh = (k.y + 7) >> 3;
        for (int j = 0; j < bh && !stop; ++j)
          for (int i = 0; i < bw; ++i) {
            if (!z.decode_block(data, c)) return bail(z.err.c_str());
            jpeg_idct(k.data.data() + size_t(k.w2) * size_t(j) * 8 + size_t(i) * 8, k.w2, data);
            if (--z.todo <= 0) {
              if (z.code_bits < 24) z.grow();
              if (!(z.marker >= 0xd0 && z.marker <= 0xd7)) {
                stop = true;  // not a restart: keep what was decoded
                break;
              }
              z.reset();
            }
          }
      } else {
        for (int j = 0; j < z.mcu_y && !stop; ++j)
          for (int i = 0; i < z.mcu_x; ++i) {
            for (int s = 0; s < z.scan_n; ++s) {
              const int c = z.order[s];
              JpegComp& k = z.comp[c];
              for (int y = 0; y < k.v; ++y)
                for (int x = 0; x < k.h; ++x) {
                  const int x2 = (i * k.h + x) * 8, y2 = (j * k.v + y) * 8;
                  if (!z.decode_block(data, c)) return bail(z.err.c_str());
                  jpeg_idct(k.data.data() + size_t(k.w2) * size_t(y2) + size_t(x2), k.w2, data);
                }
            }
            if (--z.todo <= 0) {
              if (z.code_bits < 24) z.grow();
              if (!(z.marker >= 0xd0 && z.marker <= 0xd7)) {
                stop = true;
                break;
              }
              z.reset();
            }
          }
      }
      if (z.marker == 0xff) {  // skip stray bytes up to the next marker
        while (!z.eof()) {
          if (z.get8() == 255) {
            z.marker = z.get8();
            break;
          }
        }
      }
    } else if (m == 0xdc) {
      const int Ld = z.get16(), NL = z.get16();
      if (Ld != 4 || NL != z.img_y) return bail("bad DNL");
    } else {
      if (!process_marker(m)) return bail(z.err.c_str());
    }
    m = z.get_marker();
  }
  if (z.progressive) {  // all scans are in: dequantise and transform every block
    short data[64];
    for (int c = 0; c < z.img_n; ++c) {
      JpegComp& k = z.comp[c];
      const int bw = (k.x + 7) >> 3, bh = (k.y + 7) >> 3;
      const uint16_t* dq = z.dequant[k.tq];
      for (int j = 0; j < bh; ++j)
        for (int i = 0; i < bw; ++i) {
          const short* blk = k.coeff.data() + 64 * (size_t(i) + size_t(j) * size_t(k.coeff_w));
          for (int t = 0; t < 64; ++t) data[t] = short(int(blk[t]) * int(dq[t]));
          jpeg_idct(k.data.data() + size_t(k.w2) * size_t(j) * 8 + size_t(i) * 8, k.w2, data);
        }
    }
  }
  // upsample + colour conversion (load_jpeg_image)
  const int out_n = z.img_n >= 3 ? 3 : 1;
  const bool is_rgb = z.img_n == 3 && (z.rgb == 3 || (z.app14 == 0 && !z.jfif));
  struct Res {
    int hs, vs, w_lores, ystep, ypos;
    const uint8_t *line0, *line1;
    std::vector<uint8_t> buf;
  } res[4];
  for (int k = 0; k < z.img_n; ++k) {
    Res& r = res[k];
    r.hs = z.h_max / z.comp[k].h, r.vs = z.v_max / z.comp[k].v;
    r.ystep = r.vs >> 1;
    r.w_lores = (z.img_x + r.hs - 1) / r.hs;
    r.ypos = 0;
    r.line0 = r.line1 = z.comp[k].data.data();
    r.buf.assign(size_t(z.img_x) + 3 + 8, 0);
  }
  pixels->assign(size_t(out_n) * size_t(z.img_x) * size_t(z.img_y), 0);
  const uint8_t* co[4] = {nullptr, nullptr, nullptr, nullptr};
  std::vector<uint8_t> tmp(size_t(z.img_x) * 3);
  for (int j = 0; j < z.img_y; ++j) {
    uint8_t* out = pixels->data() + size_t(out_n) * size_t(z.img_x) * size_t(j);
    for (int k = 0; k < z.img_n; ++k) {
      Res& r = res[k];
      const bool y_bot = r.ystep >= (r.vs >> 1);
      co[k] = jpeg_resample(r.buf.data(), y_bot ? r.line1 : r.line0, y_bot ? r.line0 : r.line1, r.w_lores, r.hs, r.vs);
      if (++r.ystep >= r.vs) {
        r.ystep = 0;
        r.line0 = r.line1;
        if (++r.ypos < z.comp[k].y) r.line1 += z.comp[k].w2;
      }
    }
    if (z.img_n == 3) {
      if (is_rgb) {
        for (int i = 0; i < z.img_x; ++i) out[3 * i] = co[0][i], out[3 * i + 1] = co[1][i], out[3 * i + 2] = co[2][i];
      } else {
        jpeg_ycc_row(out, co[0], co[1], co[2], z.img_x, 3);
      }
    } else if (z.img_n == 4) {
      if (z.app14 == 0) {  // CMYK
        for (int i = 0; i < z.img_x; ++i) {
          const uint8_t mk = co[3][i];
          out[3 * i] = blinn8(co[0][i], mk), out[3 * i + 1] = blinn8(co[1][i], mk), out[3 * i + 2] = blinn8(co[2][i], mk);
        }
      } else if (z.app14 == 2) {  // YCCK
        jpeg_ycc_row(out, co[0], co[1], co[2], z.img_x, 3);
        for (int i = 0; i < z.img_x; ++i) {
          const uint8_t mk = co[3][i];
          out[3 * i] = blinn8(uint8_t(255 - out[3 * i]), mk), out[3 * i + 1] = blinn8(uint8_t(255 - out[3 * i + 1]), mk);
          out[3 * i + 2] = blinn8(uint8_t(255 - out[3 * i + 2]), mk);
        }
      } else {
        jpeg_ycc_row(out, co[0], co[1], co[2], z.img_x, 3);
      }
    } else {
      for (int i = 0; i < z.img_x; ++i) out[i] = co[0][i];
    }
  }
  *width = size_t(z.img_x), *height = size_t(z.img_y), *channels = size_t(out_n);
  return true;
}

// ---------------------------------------------------------------- OpenEXR (what tinyexr's LoadEXR returns)
namespace {
float half_to_float(uint16_t h) {  // tinyexr.h:928-950
  union { uint32_t u; float f; } o, magic;
  magic.u = 113u << 23;
  const uint32_t shifted_exp = 0x7c00u << 13;
  o.u = (h & 0x7fffu) << 13;
  const uint32_t e = shifted_exp & o.u;
  o.u += (127 - 15) << 23;
  if (e == shifted_exp) o.u += (128 - 16) << 23;  // Inf / NaN
  else if (e == 0) {                               // zero / denormal
    o.u += 1 << 23;
    o.f -= magic.f;
  }
  o.u |= (h & 0x8000u) << 16;
  return o.f;
}

struct ExrChannel {
  std::string name;
  int type = 0;  // 0 UINT, 1 HALF, 2 FLOAT
  size_t offset = 0;  // byte offset of this channel inside one pixel's worth of a scanline (x width)
};

// OpenEXR's RLE (ImfRle.cpp, as in tinyexr.h:1524-1556)
bool rle_uncompress(const uint8_t* in, size_t in_len, std::vector<uint8_t>* out, size_t max_len) {
  out->clear();
  size_t i = 0;
  while (i < in_len) {
    const int8_t c = int8_t(in[i++]);
    if (c < 0) {
      const size_t count = size_t(-int(c));
      if (i + count > in_len || out->size() + count > max_len) return false;
      out->insert(out->end(), in + i, in + i + count);
      i += count;
    } else {
      const size_t count = size_t(c) + 1;
      if (i >= in_len || out->size() + count > max_len) return false;
      out->insert(out->end(), count, in[i++]);
    }
  }
  return true;
}

// predictor + de-interleave applied to ZIP / RLE blocks (ImfZipCompressor.cpp, tinyexr.h:1404-1437)
void exr_unfilter(std::vector<uint8_t>* buf) {
  std::vector<uint8_t>& t = *buf;
  for (size_t i = 1; i < t.size(); ++i) t[i] = uint8_t(int(t[i - 1]) + int(t[i]) - 128);
  std::vector<uint8_t> out(t.size());
  const size_t half = (t.size() + 1) / 2;
  size_t a = 0, b = half;
  for (size_t s = 0; s < out.size();) {
    out[s++] = t[a++];
    if (s < out.size()) out[s++] = t[b++];
  }
  t.swap(out);
}
}  // namespace

// Single-part scanline OpenEXR -> RGBA float (LoadEXR, tinyexr.h:6004-6260): one channel is replicated into all
// four; otherwise R, G, B are required and A defaults to 1.  Compression NONE / RLE / ZIPS / ZIP; HALF and FLOAT channels.
namespace {
// ---- PIZ (OpenEXR): the 16-bit words of a block of 32 scan lines are (1) mapped to a dense value range through a table
// derived from a bitmap of the values that occur, (2) transformed channel by channel with a 2-D integer wavelet, (3) coded with
// a canonical Huffman code that has a run-length symbol.  Decoding undoes 3, 2, 1.  Format: OpenEXR's ImfPizCompressor /
// ImfHuf / ImfWav as tinyexr reads them (tinyexr.h:1694-3288).
struct MsbBits {  // bits most-significant first
  const uint8_t* p;
  size_t n, i = 0;
  uint64_t acc = 0;
  int have = 0;
  uint32_t get(int nb) {
    while (have < nb) acc = (acc << 8) | (i < n ? p[i] : 0u), ++i, have += 8;
    have -= nb;
    return uint32_t((acc >> have) & ((uint64_t(1) << nb) - 1u));
  }
};

bool piz_huffman(const uint8_t* in, size_t len, uint16_t* out, size_t count) {
  if (len < 20) return false;
  auto le32 = [&](size_t at) { return uint32_t(in[at]) | (uint32_t(in[at + 1]) << 8) | (uint32_t(in[at + 2]) << 16) | (uint32_t(in[at + 3]) << 24); };
  const uint32_t im = le32(0), iM = le32(4), nbits = le32(12);
  const uint32_t kSymbols = (1u << 16) + 1u;  // 65536 values + the run-length symbol (= iM)
  if (im >= kSymbols || iM >= kSymbols) return false;
  // code lengths, 6 bits each; 59..62 = 2..5 zero lengths, 63 = 6 + (next 8 bits) zero lengths
  std::vector<uint8_t> length(kSymbols, 0);
  MsbBits tb{in + 20, len - 20};
  for (uint32_t sym = im; sym <= iM; ++sym) {
    if (tb.i > tb.n + 8) return false;
    const uint32_t l = tb.get(6);
    length[sym] = uint8_t(l);
    if (l >= 59) {
      uint32_t zeros = l == 63 ? tb.get(8) + 6u : l - 59u + 2u;
      if (sym + zeros > iM + 1u) return false;
      while (zeros--) length[sym++] = 0;
      --sym;
    }
  }
  const size_t data_at = 20 + std::min(tb.i, tb.n);
  if (uint64_t(nbits) > 8ull * (len - data_at)) return false;
  // canonical code: the longest codes get the smallest values; within one length, values rise with the symbol
  uint64_t base[60], cnt[60] = {0};
  for (uint32_t sym = 0; sym < kSymbols; ++sym) cnt[length[sym]]++;
  uint64_t c = 0;
  for (int l = 58; l > 0; --l) {
    base[l] = c;
    c = (c + cnt[l]) >> 1;
  }
  std::vector<uint32_t> first(60, 0), order;  // symbols sorted by (length, symbol)
  order.reserve(size_t(iM - im) + 1);
  for (int l = 1; l <= 58; ++l) {
    first[size_t(l)] = uint32_t(order.size());
    if (cnt[l])
      for (uint32_t sym = im; sym <= iM; ++sym)
        if (length[sym] == l) order.push_back(sym);
  }
  MsbBits br{in + data_at, len - data_at};
  uint64_t left = nbits;
  size_t produced = 0;
  while (left > 0) {
    uint64_t v = 0;
    int l = 0;
    uint32_t sym = kSymbols;
    while (left > 0 && l < 58) {
      v = (v << 1) | br.get(1), ++l, --left;
      if (cnt[l] && v >= base[l] && v - base[l] < cnt[l]) {
        sym = order[first[size_t(l)] + size_t(v - base[l])];
        break;
      }
    }
    if (sym == kSymbols) return false;  // bits that are no code
    if (sym == iM) {  // run: repeat the previous word
      if (left < 8 || produced == 0) return false;
      const uint32_t run = br.get(8);
      left -= 8;
      if (produced + run > count) return false;
      for (uint32_t k = 0; k < run; ++k) out[produced + k] = out[produced - 1];
      produced += run;
    } else {
      if (produced >= count) return false;
      out[produced++] = uint16_t(sym);
    }
  }
  return produced == count;
}

// inverse of one wavelet butterfly; the 14-bit form is used when every value is below 2^14 (no wrap-around possible)
inline void wav_dec14(uint16_t l, uint16_t h, uint16_t& a, uint16_t& b) {
  const int ls = int16_t(l), hs = int16_t(h);
  const int ai = ls + (hs & 1) + (hs >> 1);
  a = uint16_t(int16_t(ai)), b = uint16_t(int16_t(ai - hs));
}
inline void wav_dec16(uint16_t l, uint16_t h, uint16_t& a, uint16_t& b) {
  const int m = l, d = h;
  const int bb = (m - (d >> 1)) & 0xffff;
  a = uint16_t((d + bb - 0x8000) & 0xffff), b = uint16_t(bb);
}
void piz_wavelet(uint16_t* in, int nx, int ox, int ny, int oy, uint16_t max_value) {
  const bool w14 = max_value < (1 << 14);
  const int n = nx > ny ? ny : nx;
  int p = 1;
  while (p <= n) p <<= 1;
  p >>= 1;
  int p2 = p;
  p >>= 1;
  auto dec = [&](uint16_t l, uint16_t h, uint16_t& a, uint16_t& b) { w14 ? wav_dec14(l, h, a, b) : wav_dec16(l, h, a, b); };
  while (p >= 1) {  // from the coarsest level down: rebuild 2x2 groups that are p apart
    const int oy1 = oy * p, oy2 = oy * p2, ox1 = ox * p, ox2 = ox * p2;
    uint16_t* py = in;
    uint16_t* const ey = in + oy * (ny - p2);
    for (; py <= ey; py += oy2) {
      uint16_t* px = py;
      uint16_t* const ex = py + ox * (nx - p2);
      for (; px <= ex; px += ox2) {
        uint16_t *p01 = px + ox1, *p10 = px + oy1, *p11 = p10 + ox1;
        uint16_t i00, i01, i10, i11;
        dec(*px, *p10, i00, i10);
        dec(*p01, *p11, i01, i11);
        dec(i00, i01, *px, *p01);
        dec(i10, i11, *p10, *p11);
      }
      if (nx & p) {  // odd column left over at this level
        uint16_t* p10 = px + oy1;
        uint16_t i00;
        dec(*px, *p10, i00, *p10);
        *px = i00;
      }
    }
    if (ny & p) {  // odd row
      uint16_t* px = py;
      uint16_t* const ex = py + ox * (nx - p2);
      for (; px <= ex; px += ox2) {
        uint16_t* p01 = px + ox1;
        uint16_t i00;
        dec(*px, *p01, i00, *p01);
        *px = i00;
      }
    }
    p2 = p;
    p >>= 1;
  }
}

// one PIZ chunk -> the uncompressed chunk layout (line after line, inside a line channel after channel)
bool piz_uncompress(const uint8_t* in, size_t in_len, std::vector<uint8_t>* out, size_t w, size_t lines, const std::vector<ExrChannel>& ch) {
  if (in_len < 4) return false;
  const uint32_t min_nz = in[0] | (uint32_t(in[1]) << 8), max_nz = in[2] | (uint32_t(in[3]) << 8);
  if (max_nz >= 8192 || min_nz > max_nz) return false;  // (tinyexr refuses the empty bitmap of an all-zero block too)
  size_t at = 4;
  if (at + (max_nz - min_nz + 1) + 4 > in_len) return false;
  std::vector<uint8_t> bitmap(8192, 0);
  memcpy(bitmap.data() + min_nz, in + at, max_nz - min_nz + 1);
  at += max_nz - min_nz + 1;
  std::vector<uint16_t> lut(1 << 16, 0);  // dense index -> value
  uint32_t k = 0;
  for (uint32_t v = 0; v < (1u << 16); ++v)
    if (v == 0 || (bitmap[v >> 3] & (1u << (v & 7)))) lut[k++] = uint16_t(v);
  const uint16_t max_value = uint16_t(k - 1);
  const int32_t hlen = int32_t(uint32_t(in[at]) | (uint32_t(in[at + 1]) << 8) | (uint32_t(in[at + 2]) << 16) | (uint32_t(in[at + 3]) << 24));
  at += 4;
  if (hlen < 0 || at + size_t(hlen) > in_len) return false;
  size_t words = 0;
  for (const ExrChannel& c : ch) words += w * lines * (c.type == 1 ? 1 : 2);
  std::vector<uint16_t> tmp(words, 0);
  (void)piz_huffman(in + at, size_t(hlen), tmp.data(), words);  // tinyexr goes on with whatever was decoded
  size_t start = 0;
  for (const ExrChannel& c : ch) {
    const int size = c.type == 1 ? 1 : 2;
    for (int j = 0; j < size; ++j) piz_wavelet(tmp.data() + start + size_t(j), int(w), size, int(lines), int(w) * size, max_value);
    start += w * lines * size_t(size);
  }
  for (uint16_t& v : tmp) v = lut[v];
  out->resize(words * 2);
  std::vector<size_t> next(ch.size());
  start = 0;
  for (size_t c = 0; c < ch.size(); ++c) next[c] = start, start += w * lines * (ch[c].type == 1 ? 1 : 2);
  size_t o = 0;
  for (size_t y = 0; y < lines; ++y)
    for (size_t c = 0; c < ch.size(); ++c) {
      const size_t nw = w * (ch[c].type == 1 ? 1 : 2);
      for (size_t q = 0; q < nw; ++q) {
        const uint16_t v = tmp[next[c] + q];
        (*out)[o++] = uint8_t(v & 255), (*out)[o++] = uint8_t(v >> 8);
      }
      next[c] += nw;
    }
  return true;
}
}  // namespace

bool DecodeExr(const uint8_t* file, size_t n, std::vector<float>* pixels, size_t* width, size_t* height, std::string* err) {
  auto rd32 = [&](size_t at) { return uint32_t(file[at]) | (uint32_t(file[at + 1]) << 8) | (uint32_t(file[at + 2]) << 16) | (uint32_t(file[at + 3]) << 24); };
  if (n < 8 || rd32(0) != 0x01312f76u) return *err = "not an OpenEXR file", false;
  if (file[4] != 2) return *err = "unsupported OpenEXR version", false;
  if (file[5] & 0x02) return *err = "tiled OpenEXR files are not decoded by this build", false;
  if (file[5] & 0x18) return *err = "multipart / deep OpenEXR files are not decoded (LoadEXR rejects them too)", false;
  size_t pos = 8;
  std::vector<ExrChannel> ch;
  int compression = -1, line_order = -1;
  int32_t dw[4] = {0, 0, -1, -1};
  bool have_dw = false, have_disp = false, have_par = false, have_swc = false, have_sww = false;
  for (;;) {
    if (pos >= n) return *err = "truncated OpenEXR header", false;
    if (file[pos] == 0) {
      ++pos;
      break;
    }
    std::string name, type;
    while (pos < n && file[pos]) name.push_back(char(file[pos++]));
    ++pos;
    while (pos < n && file[pos]) type.push_back(char(file[pos++]));
    ++pos;
    if (pos + 4 > n) return *err = "truncated OpenEXR header", false;
    const size_t size = rd32(pos);
    pos += 4;
    if (pos + size > n) return *err = "truncated OpenEXR header", false;
    const size_t at = pos;
    pos += size;
    if (name == "channels") {
      size_t q = at;
      while (q < at + size && file[q]) {
        ExrChannel c;
        while (q < at + size && file[q]) c.name.push_back(char(file[q++]));
        ++q;
        if (q + 16 > at + size) return *err = "bad OpenEXR channel list", false;
        c.type = int(rd32(q));
        const uint32_t xs = rd32(q + 8), ys = rd32(q + 12);
        if (xs != 1 || ys != 1) return *err = "subsampled OpenEXR channels are not decoded", false;
        q += 16;
        ch.push_back(c);
      }
    } else if (name == "compression" && size >= 1) {
      compression = file[at];
    } else if (name == "dataWindow" && size >= 16) {
      for (int k = 0; k < 4; ++k) dw[k] = int32_t(rd32(at + 4 * size_t(k)));
      have_dw = true;
    } else if (name == "displayWindow") {
      have_disp = true;
    } else if (name == "lineOrder" && size >= 1) {
      line_order = file[at];
    } else if (name == "pixelAspectRatio") {
      have_par = true;
    } else if (name == "screenWindowCenter") {
      have_swc = true;
    } else if (name == "screenWindowWidth") {
      have_sww = true;
    }
  }
  if (ch.empty() || compression < 0 || !have_dw || !have_disp || line_order < 0 || !have_par || !have_swc || !have_sww)
    return *err = "OpenEXR header lacks a required attribute", false;
  if (compression > 4) return *err = "OpenEXR compression PXR24 / B44 / DWA is not decoded by this build (use ZIP or PIZ)", false;
  if (dw[2] < dw[0] || dw[3] < dw[1]) return *err = "bad OpenEXR data window", false;
  const size_t w = size_t(int64_t(dw[2]) - int64_t(dw[0])) + 1, h = size_t(int64_t(dw[3]) - int64_t(dw[1])) + 1;  // (64-bit: the corners are arbitrary int32)
  if (w > (1u << 24) || h > (1u << 24) || uint64_t(w) * h > (1ull << 28)) return *err = "bad OpenEXR data window", false;
  size_t pixel_bytes = 0;
  for (ExrChannel& c : ch) {
    if (c.type != 1 && c.type != 2) return *err = "UINT OpenEXR channels are not decoded", false;
    c.offset = pixel_bytes;
    pixel_bytes += c.type == 1 ? 2 : 4;
  }
  const size_t block_lines = compression == 4 ? 32 : (compression == 3 ? 16 : 1);
  const size_t nblocks = (h + block_lines - 1) / block_lines;
  if (pos + nblocks * 8 > n) return *err = "truncated OpenEXR offset table", false;
  std::vector<std::vector<float>> img(ch.size(), std::vector<float>(w * h, 0.0f));
  std::vector<uint8_t> block;
  for (size_t b = 0; b < nblocks; ++b) {
    uint64_t off = 0;
    for (int k = 7; k >= 0; --k) off = (off << 8) | file[pos + b * 8 + size_t(k)];
    if (off + 8 > n) return *err = "bad OpenEXR chunk offset", false;
    const int32_t line = int32_t(rd32(size_t(off)));
    const uint32_t len = rd32(size_t(off) + 4);
    if (len == 0 || off + 8 + len > n) return *err = "bad OpenEXR chunk", false;
    const int64_t first = int64_t(line) - dw[1];
    if (first < 0 || first >= int64_t(h)) return *err = "bad OpenEXR chunk", false;
    const size_t lines = std::min<size_t>(block_lines, h - size_t(first));
    const size_t raw = w * lines * pixel_bytes;
    const uint8_t* src = file + off + 8;
    if (compression == 0 || len == raw) {  // stored (a block that did not shrink is stored raw too)
      if (len < raw) return *err = "short OpenEXR chunk", false;
      block.assign(src, src + raw);
    } else if (compression == 1) {
      if (!rle_uncompress(src, len, &block, raw) || block.size() != raw) return *err = "bad RLE data in OpenEXR chunk", false;
      exr_unfilter(&block);
    } else if (compression == 4) {
      if (!piz_uncompress(src, len, &block, w, lines, ch) || block.size() != raw) return *err = "bad PIZ data in OpenEXR chunk", false;
    } else {
      std::string zerr;
      if (!ZlibInflate(src, len, &block, &zerr) || block.size() != raw) return *err = "bad zlib data in OpenEXR chunk", false;
      exr_unfilter(&block);
    }
    for (size_t c = 0; c < ch.size(); ++c)
      for (size_t v = 0; v < lines; ++v) {
        const uint8_t* lp = block.data() + v * pixel_bytes * w + ch[c].offset * w;
        const size_t y = size_t(first) + v;
        const size_t row = line_order == 0 ? y : h - 1 - y;  // tinyexr stores decreasing-Y files upside down
        float* dst = &img[c][row * w];
        for (size_t u = 0; u < w; ++u) {
          if (ch[c].type == 1) {
            dst[u] = half_to_float(uint16_t(lp[2 * u] | (lp[2 * u + 1] << 8)));
          } else {
            uint32_t bits = uint32_t(lp[4 * u]) | (uint32_t(lp[4 * u + 1]) << 8) | (uint32_t(lp[4 * u + 2]) << 16) | (uint32_t(lp[4 * u + 3]) << 24);
            memcpy(&dst[u], &bits, 4);
          }
        }
      }
  }
  pixels->assign(w * h * 4, 0.0f);
  if (ch.size() == 1) {
    for (size_t i = 0; i < w * h; ++i)
      for (int k = 0; k < 4; ++k) (*pixels)[4 * i + size_t(k)] = img[0][i];
  } else {
    int idx[4] = {-1, -1, -1, -1};
    static const char* names[4] = {"R", "G", "B", "A"};
    for (size_t c = 0; c < ch.size() && c < 4; ++c)  // LoadEXR looks at the first four channels only
      for (int k = 0; k < 4; ++k)
        if (ch[c].name == names[k]) idx[k] = int(c);
    for (int k = 0; k < 3; ++k)
      if (idx[k] < 0) return *err = std::string(names[k]) + " channel not found in the OpenEXR file", false;
    for (size_t i = 0; i < w * h; ++i) {
      for (int k = 0; k < 3; ++k) (*pixels)[4 * i + size_t(k)] = img[size_t(idx[k])][i];
      (*pixels)[4 * i + 3] = idx[3] >= 0 ? img[size_t(idx[3])][i] : 1.0f;
    }
  }
  *width = w, *height = h;
  return true;
}

// ---------------------------------------------------------------- pbrlab's image functions
namespace {
std::string join_path(const std::string& dir, const std::string& name) {  // fs::path(dir) / name
  if (!name.empty() && name[0] == '/') return name;
  if (dir.empty()) return name;
  return dir.back() == '/' ? dir + name : dir + "/" + name;
}
std::string lower_ext(const std::string& name) {
  const size_t slash = name.find_last_of('/');
  const size_t dot = name.find_last_of('.');
  if (dot == std::string::npos || (slash != std::string::npos && dot < slash)) return "";
  std::string e = name.substr(dot);
  for (char& c : e) c = char(tolower(c));
  return e;
}
bool read_file(const std::string& path, std::vector<uint8_t>* out) {
  std::ifstream f(path.c_str(), std::ios::binary);
  if (!f) return false;
  out->assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
  return true;
}
bool is_pic(const std::vector<uint8_t>& d) {  // Softimage PIC
  return d.size() >= 92 && d[0] == 0x53 && d[1] == 0x80 && d[2] == 0xF6 && d[3] == 0x34 && memcmp(d.data() + 88, "PICT", 4) == 0;
}
bool is_radiance(const std::vector<uint8_t>& d) {
  return (d.size() >= 11 && memcmp(d.data(), "#?RADIANCE\n", 11) == 0) || (d.size() >= 7 && memcmp(d.data(), "#?RGBE\n", 7) == 0);
}
}  // namespace

bool LoadImageFromFile(const std::string& filename, const std::string& asset_path, std::vector<float>* pixels,
                       size_t* width, size_t* height, size_t* channels) {
  if (!pixels || !width || !height || !channels) return false;
  const std::string path = join_path(asset_path, filename);
  const std::string ext = lower_ext(filename);
  std::vector<uint8_t> bytes;
  if (!read_file(path, &bytes)) {
    std::cerr << "cannot open image file [" << path << "]" << std::endl;
    return false;
  }
  std::string err;
  if (ext == ".exr") {
    if (!DecodeExr(bytes.data(), bytes.size(), pixels, width, height, &err)) {
      std::cerr << "image file [" << path << "]: " << err << std::endl;
      return false;
    }
    *channels = 4;
    return true;
  }
  if (ext == ".hdr") {
    if (!DecodeHdr(bytes.data(), bytes.size(), pixels, width, height, &err)) {
      std::cerr << "image file [" << path << "]: " << err << std::endl;
      return false;
    }
    *channels = 3;
    return true;
  }
  if (bytes.size() >= 8 && memcmp(bytes.data(), kPngSig, 8) == 0) {
    std::vector<uint8_t> px8;
    if (!DecodePng(bytes.data(), bytes.size(), &px8, width, height, channels, &err)) {
      std::cerr << "image file [" << path << "]: " << err << std::endl;
      return false;
    }
    pixels->resize(px8.size());
    for (size_t i = 0; i < px8.size(); ++i) (*pixels)[i] = float(px8[i]) / float(255);
    return true;
  }
  if (bytes.size() >= 3 && bytes[0] == 0xFF && bytes[1] == 0xD8) {
    std::vector<uint8_t> px8;
    if (!DecodeJpeg(bytes.data(), bytes.size(), &px8, width, height, channels, &err)) {
      std::cerr << "image file [" << path << "]: " << err << std::endl;
      return false;
    }
    pixels->resize(px8.size());
    for (size_t i = 0; i < px8.size(); ++i) (*pixels)[i] = float(px8[i]) / float(255);
    return true;
  }
  // stb_image's order for what is left: BMP, GIF, PSD, PIC, PNM, Radiance (tone-mapped to 8 bits), and TGA last because it
  // has no signature
  typedef bool (*Decoder)(const uint8_t*, size_t, std::vector<uint8_t>*, size_t*, size_t*, size_t*, std::string*);
  Decoder dec = nullptr;
  if (IsBmp(bytes.data(), bytes.size())) dec = DecodeBmp;
  else if (IsGif(bytes.data(), bytes.size())) dec = DecodeGif;
  else if (IsPsd(bytes.data(), bytes.size())) dec = DecodePsd;
  else if (is_pic(bytes)) {
    std::cerr << "image file [" << path << "]: Softimage PIC is not decoded by this build (convert to .png)" << std::endl;
    return false;
  } else if (IsPnm(bytes.data(), bytes.size())) dec = DecodePnm;
  else if (is_radiance(bytes)) {
    std::vector<float> hdr;
    if (!DecodeHdr(bytes.data(), bytes.size(), &hdr, width, height, &err)) {
      std::cerr << "image file [" << path << "]: " << err << std::endl;
      return false;
    }
    std::vector<uint8_t> px8;
    HdrToLdr(hdr, &px8);
    *channels = 3;
    pixels->resize(px8.size());
    for (size_t i = 0; i < px8.size(); ++i) (*pixels)[i] = float(px8[i]) / float(255);
    return true;
  } else if (IsTga(bytes.data(), bytes.size())) dec = DecodeTga;
  if (!dec) {
    std::cerr << "image file [" << path << "]: not an image of any known type" << std::endl;
    return false;
  }
  std::vector<uint8_t> px8;
  if (!dec(bytes.data(), bytes.size(), &px8, width, height, channels, &err)) {
    std::cerr << "image file [" << path << "]: " << err << std::endl;
    return false;
  }
  pixels->resize(px8.size());
  for (size_t i = 0; i < px8.size(); ++i) (*pixels)[i] = float(px8[i]) / float(255);
  return true;
}

namespace {
template <typename T>
bool write_png_checked(const std::string& filename, const std::string& asset_path, const std::vector<T>& pixels,
                       size_t width, size_t height, size_t channels, std::vector<uint8_t>* px8, std::string* path) {
  *path = join_path(asset_path, filename);
  const size_t slash = path->find_last_of('/'), dot = path->find_last_of('.');
  const std::string ext = (dot == std::string::npos || (slash != std::string::npos && dot < slash)) ? "" : path->substr(dot);
  if (ext != ".png") {
    std::cerr << "warning! the file extension is not \"png\"" << std::endl;
    return false;
  }
  if (pixels.size() == 0 || pixels.size() != width * height * channels) {
    std::cerr << "the image data is broken" << std::endl;
    return false;
  }
  px8->resize(pixels.size());
  return true;
}
bool finish_png(const std::string& filename, const std::string& path, const std::vector<uint8_t>& px8, size_t width,
                size_t height, size_t channels) {
  std::vector<uint8_t> file;
  bool ok = EncodePng(px8.data(), width, height, channels, &file);
  if (ok) {
    FILE* fp = fopen(path.c_str(), "wb");
    ok = fp && fwrite(file.data(), 1, file.size(), fp) == file.size();
    if (fp) ok = (fclose(fp) == 0) && ok;
  }
  if (!ok) {
    std::cerr << "faild save image" << std::endl;
    return false;
  }
  std::cerr << "write png file [ " << filename << " ]" << std::endl;
  return true;
}
inline uint8_t to_byte(float x) {  // static_cast<unsigned char>(Clamp(x * 256.0f, 0.0f, 255.0f)), image-io.cc:37-40,203-205
  const float s = x * 256.0f;
  const float lo = (s < 255.0f) ? s : 255.0f;   // std::min(255, s): s when s < 255 (NaN -> 255)
  const float v = (0.0f < lo) ? lo : 0.0f;      // std::max(0, lo)
  return static_cast<uint8_t>(v);
}
}  // namespace

bool WritePNG(const std::string& filename, const std::string& asset_path, const std::vector<float>& pixels, size_t width,
              size_t height, size_t channels) {
  std::vector<uint8_t> px8;
  std::string path;
  if (!write_png_checked(filename, asset_path, pixels, width, height, channels, &px8, &path)) return false;
  for (size_t i = 0; i < pixels.size(); ++i) px8[i] = to_byte(pixels[i]);
  return finish_png(filename, path, px8, width, height, channels);
}

bool WritePNG(const std::string& filename, const std::string& asset_path, const std::vector<uint8_t>& pixels, size_t width,
              size_t height, size_t channels) {
  std::vector<uint8_t> px8;
  std::string path;
  if (!write_png_checked(filename, asset_path, pixels, width, height, channels, &px8, &path)) return false;
  px8 = pixels;
  return finish_png(filename, path, px8, width, height, channels);
}

float SrgbToLiner(float c) {
  if (c <= 0.04045f) return c / 12.92f;
  const float a = 0.055f;
  return powf((c + a) / (1.0f + a), 2.4f);
}

float LinerTosRGB(float c) {
  if (c <= 0.0031308f) return 12.92f * c;
  const float a = 0.055f;
  return powf((1.0f + a) * c, static_cast<float>(1.0 / 2.4)) - a;
}

void SrgbToLiner(const std::vector<float>& src, size_t width, size_t height, size_t channels, std::vector<float>* out) {
  std::vector<float> dst(width * height * channels);
  for (size_t i = 0; i < width * height; ++i)
    for (size_t k = 0; k < channels; ++k) dst[i * channels + k] = k < 3 ? SrgbToLiner(src[i * channels + k]) : src[i * channels + k];
  out->swap(dst);
}

void LinerToSrgb(const std::vector<float>& src, size_t width, size_t height, size_t channels, std::vector<float>* out) {
  std::vector<float> dst(width * height * channels);
  for (size_t i = 0; i < width * height; ++i)
    for (size_t k = 0; k < channels; ++k) dst[i * channels + k] = k < 3 ? LinerTosRGB(src[i * channels + k]) : src[i * channels + k];
  out->swap(dst);
}

void ResolveLayerToSrgb8(const float* rgba, const uint32_t* count, size_t width, size_t height, std::vector<uint8_t>* out) {
  out->resize(width * height * 4);
  for (size_t i = 0; i < width * height; ++i) {
    const float n = float(count[i]);
    for (size_t k = 0; k < 4; ++k) {
      const float c = rgba[i * 4 + k] / n;
      (*out)[i * 4 + k] = to_byte(k < 3 ? LinerTosRGB(c) : c);
    }
  }
}

}  // namespace pbio
