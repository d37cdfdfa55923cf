// oi_io.cpp -- host-side image readers of the Diffusion inputs-file interface:
// file -> thresholded 0/1 phase bytes in file order (x fastest, page/plane slowest).
//
// Restates the behaviour of the reference readers without libtiff:
//   TiffReader  (src/io/TiffReader.cpp:139-202 metadata, :289-444 decode): multi-directory
//               stacks and numbered file sequences (:85-88, :209-285), strips or tiles,
//               1/8/16/32/64 bits per sample, UINT/INT/IEEEFP, 1 sample per pixel, contiguous
//               planar config; 1-bit data honours FILLORDER and ignores PHOTOMETRIC;
//               voxel > threshold -> 1 else 0.
//               Encodings decoded here (what libtiff decodes for the reference): none, LZW (5),
//               Deflate (8 / 32946, zlib), PackBits (32773), JPEG (7, with JPEGTables; sequential and progressive), horizontal predictor (317 = 2), floating-point predictor (317 = 3);
//               classic TIFF and BigTIFF, both byte orders.  Anything else is reported, never guessed.
//   RawReader   (src/io/RawReader.cpp:289-371 getValue, :379-491 threshold).
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <atomic>
#include <climits>
#include <cmath>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/openimpala_amd.h"
#include "oi_hostpar.h"

namespace oi {
void set_error(const char* fmt, ...);
const char* last_error();
}
// Output buffers of the readers: plain malloc, or -- after oi_read_set_pinned(1) (oi_tort.cpp, the part of the library that
// knows HIP) -- page-locked host memory, so that the upload of a slab is one DMA transfer (the reference stages through
// pageable std::vectors).  The hooks keep this file free of any device API (it is also built alone, by g++, for ASan).
extern "C" void oi_free_host(void* p);
namespace oi {
void* (*g_host_alloc)(size_t) = nullptr;
void (*g_host_free)(void*) = nullptr;
void* host_alloc(size_t n, bool zero) {
  void* p = g_host_alloc ? g_host_alloc(n) : malloc(n);
  if (p && zero) memset(p, 0, n);
  return p;
}
}  // namespace oi


namespace {

// read-only view of a file: mapped, so a reader only pages in the bytes it decodes (a z-slab of a
// multi-gigabyte stack costs its own planes, not the file)
struct Bytes {
  const uint8_t* p = nullptr;
  size_t n = 0;
  bool mapped = false;
  Bytes() = default;
  Bytes(const Bytes&) = delete;
  Bytes& operator=(const Bytes&) = delete;
  ~Bytes() { if (mapped && p) munmap(const_cast<uint8_t*>(p), n); }
  bool open(const char* path) {
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) { ::close(fd); return false; }
    n = (size_t)st.st_size;
    if (n == 0) { ::close(fd); p = nullptr; return true; }
    void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (m == MAP_FAILED) { n = 0; return false; }
    p = (const uint8_t*)m;
    mapped = true;
    return true;
  }
  size_t size() const { return n; }
  const uint8_t* data() const { return p; }
  uint8_t operator[](size_t i) const { return p[i]; }
};

struct File {
  Bytes d;
  bool le = true;
  bool big = false;  // BigTIFF (magic 43): 8-byte offsets, 20-byte directory entries
  uint16_t u16(size_t o) const { return le ? (uint16_t)(d[o] | d[o + 1] << 8) : (uint16_t)(d[o] << 8 | d[o + 1]); }
  uint32_t u32(size_t o) const {
    return le ? (uint32_t)d[o] | (uint32_t)d[o + 1] << 8 | (uint32_t)d[o + 2] << 16 | (uint32_t)d[o + 3] << 24
              : (uint32_t)d[o] << 24 | (uint32_t)d[o + 1] << 16 | (uint32_t)d[o + 2] << 8 | (uint32_t)d[o + 3];
  }
  uint64_t u64(size_t o) const {
    return le ? (uint64_t)u32(o) | (uint64_t)u32(o + 4) << 32 : (uint64_t)u32(o) << 32 | (uint64_t)u32(o + 4);
  }
};

struct Ifd {
  uint32_t w = 0, h = 0, bps = 1, comp = 1, spp = 1, rps = 0, fmt = 1, fill = 1, planar = 1, predictor = 1;
  uint32_t tile_w = 0, tile_h = 0;
  std::vector<uint64_t> offs, counts;
  std::vector<uint8_t> jpeg_tables;  // tag 347: abbreviated table-specification stream shared by the strips / tiles (compression 7)
  bool tiled = false;
};

size_t type_size(uint16_t t) {
  switch (t) {
    case 1: case 2: case 6: case 7: return 1;
    case 3: case 8: return 2;
    case 4: case 9: case 11: case 13: return 4;
    case 5: case 10: case 12: case 16: case 17: case 18: return 8;
    default: return 0;
  }
}

bool read_values(const File& f, size_t entry, std::vector<uint64_t>& out) {
  const uint16_t type = f.u16(entry + 2);
  const uint64_t count = f.big ? f.u64(entry + 4) : f.u32(entry + 4);
  const size_t ts = type_size(type);
  if (ts == 0 || count > (1ull << 32)) return false;
  const size_t inline_bytes = f.big ? 8 : 4, val = entry + (f.big ? 12 : 8);
  size_t pos = val;
  const uint64_t need = (uint64_t)ts * count;  // (ts <= 8, count <= 2^32: no wrap)
  if (need > inline_bytes) {
    const uint64_t p64 = f.big ? f.u64(val) : f.u32(val);
    if (p64 > f.d.size()) return false;
    pos = (size_t)p64;
  }
  if (need > f.d.size() || pos > f.d.size() - need) return false;  // checked form of pos + need <= size
  out.resize(count);
  for (uint64_t i = 0; i < count; ++i) {
    if (ts == 1) out[i] = f.d[pos + i];
    else if (ts == 2) out[i] = f.u16(pos + 2 * i);
    else if (ts == 4) out[i] = f.u32(pos + 4 * i);
    else out[i] = (type == 16 || type == 17 || type == 18) ? f.u64(pos + 8 * i) : f.u32(pos + 8 * i);  // rationals: numerator
  }
  return true;
}

bool parse_ifd(const File& f, size_t off, Ifd& ifd, size_t& next) {
  const size_t cnt_bytes = f.big ? 8 : 2, ent_bytes = f.big ? 20 : 12, next_bytes = f.big ? 8 : 4;
  if (off + cnt_bytes > f.d.size()) return false;
  const uint64_t n = f.big ? f.u64(off) : f.u16(off);
  if (n > 65535 || off + cnt_bytes + ent_bytes * (size_t)n + next_bytes > f.d.size()) return false;
  std::vector<uint64_t> v;
  for (uint64_t e = 0; e < n; ++e) {
    const size_t ent = off + cnt_bytes + ent_bytes * (size_t)e;
    const uint16_t tag = f.u16(ent);
    if (!read_values(f, ent, v) || v.empty()) continue;
    switch (tag) {
      case 256: ifd.w = (uint32_t)v[0]; break;
      case 257: ifd.h = (uint32_t)v[0]; break;
      case 258: ifd.bps = (uint32_t)v[0]; break;
      case 259: ifd.comp = (uint32_t)v[0]; break;
      case 266: ifd.fill = (uint32_t)v[0]; break;
      case 273: ifd.offs = v; break;
      case 277: ifd.spp = (uint32_t)v[0]; break;
      case 278: ifd.rps = (uint32_t)v[0]; break;
      case 279: ifd.counts = v; break;
      case 284: ifd.planar = (uint32_t)v[0]; break;
      case 317: ifd.predictor = (uint32_t)v[0]; break;
      case 322: ifd.tile_w = (uint32_t)v[0]; ifd.tiled = true; break;
      case 323: ifd.tile_h = (uint32_t)v[0]; ifd.tiled = true; break;
      case 324: ifd.offs = v; ifd.tiled = true; break;
      case 325: ifd.counts = v; ifd.tiled = true; break;
      case 339: ifd.fmt = (uint32_t)v[0]; break;
      case 347: ifd.jpeg_tables.resize(v.size()); for (size_t i = 0; i < v.size(); ++i) ifd.jpeg_tables[i] = (uint8_t)v[i]; break;
      default: break;
    }
  }
  const size_t np = off + cnt_bytes + ent_bytes * (size_t)n;
  next = f.big ? (size_t)f.u64(np) : f.u32(np);
  return true;
}

// ---- JPEG (TIFF compression 7, "new-style" JPEG-in-TIFF; TIFF Technical Note 2) ----
// What libtiff + libjpeg decode for the reference when a tomogram was saved with JPEG compression: every strip / tile is a JPEG
// stream of its own (SOI ... EOI), usually abbreviated -- the quantisation and Huffman tables sit once in the JPEGTables tag.
// Decoded here: sequential DCT, Huffman coding, 8-bit precision, ONE component (the readers take one sample per pixel only),
// restart intervals; the inverse DCT is libjpeg's default integer one ("islow", Loeffler-Ligtenberg-Moschytz with 13-bit
// constants: what jpeg_idct_islow and libjpeg-turbo's SIMD versions of it compute); PROGRESSIVE streams (SOF2: spectral selection and
// successive approximation, round 5) the same way once their scans are in.  Arithmetic-coded, 12-bit and multi-component streams and
// old-style JPEG (compression 6) are reported.
struct JpegTables {
  uint16_t q[4][64] = {};   // quantisation tables in natural (de-zigzagged) order
  bool have_q[4] = {false, false, false, false};
  struct Huff { uint8_t bits[17] = {}; uint8_t vals[256] = {}; int mincode[17] = {}, maxcode[18] = {}, valptr[17] = {}; bool ok = false; } dc[4], ac[4];
  int restart = 0;
};
const uint8_t kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                             35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

bool jpeg_build_huff(JpegTables::Huff& h) {
  int code = 0, k = 0;
  for (int l = 1; l <= 16; ++l) {
    h.valptr[l] = k;
    h.mincode[l] = code;
    code += h.bits[l];
    k += h.bits[l];
    h.maxcode[l] = h.bits[l] ? code - 1 : -1;
    if (code > (1 << l)) return false;  // over-subscribed
    code <<= 1;
  }
  h.maxcode[17] = 0x7fffffff;
  h.ok = k <= 256;
  return h.ok;
}

// the marker segments of one stream up to (and including the header of) SOS; returns the offset of the entropy-coded data
// (progressive: the frame is SOF2; ss..se / ah, al = spectral band and successive-approximation bit positions of the scan just parsed;
// eoi: the stream ended where a further scan was looked for)
struct JpegFrame { int w = 0, h = 0, tq = 0, td = 0, ta = 0, ss = 0, se = 63, ah = 0, al = 0; bool have_sof = false, progressive = false, eoi = false; };
// start = 0: a whole stream (SOI first); start > 0: the segments after a scan of a progressive stream, up to the next SOS or EOI
bool jpeg_parse_segments(const uint8_t* p, size_t n, JpegTables& t, JpegFrame& fr, size_t& data_off, bool tables_only, std::string& why, size_t start = 0) {
  size_t i = start;
  if (start == 0) {
    if (n < 4 || p[0] != 0xFF || p[1] != 0xD8) { why = "no SOI marker"; return false; }
    i = 2;
  }
  while (i + 2 <= n) {
    if (p[i] != 0xFF) { why = "marker expected"; return false; }
    const unsigned m = p[i + 1];
    if (m == 0xFF) { ++i; continue; }  // fill byte
    if (m == 0xD9) {  // EOI: fine for a tables-only stream and after a scan
      data_off = i; fr.eoi = true;
      if (!tables_only && start == 0) why = "no scan in the JPEG stream";
      return tables_only || start > 0;
    }
    if (i + 4 > n) break;
    if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) { i += 2; continue; }
    const size_t len = (size_t)p[i + 2] << 8 | p[i + 3];
    if (len < 2 || i + 2 + len > n) { why = "truncated marker segment"; return false; }
    const uint8_t* s = p + i + 4;
    const size_t sl = len - 2;
    if (m == 0xDB) {  // DQT
      size_t o = 0;
      while (o < sl) {
        const int pq = s[o] >> 4, tq = s[o] & 15;
        ++o;
        // (8-bit baseline frames -- the only ones decoded here -- take 8-bit tables: Pq = 1 belongs to 12-bit precision, and a 16-bit
        // quantiser times a coefficient would leave the int range the dequantisation works in)
        if (tq > 3 || pq > 0 || o + 64 > sl) { why = pq ? "16-bit quantisation table in an 8-bit JPEG stream" : "bad quantisation table"; return false; }
        for (int z = 0; z < 64; ++z) {
          t.q[tq][kZigzag[z]] = pq ? (uint16_t)(s[o] << 8 | s[o + 1]) : s[o];
          o += pq ? 2 : 1;
        }
        t.have_q[tq] = true;
      }
    } else if (m == 0xC4) {  // DHT
      size_t o = 0;
      while (o < sl) {
        if (o + 17 > sl) { why = "bad Huffman table"; return false; }
        const int tc = s[o] >> 4, th = s[o] & 15;
        if (tc > 1 || th > 3) { why = "bad Huffman table id"; return false; }
        JpegTables::Huff& h = tc ? t.ac[th] : t.dc[th];
        int total = 0;
        h.bits[0] = 0;
        for (int l = 1; l <= 16; ++l) { h.bits[l] = s[o + l]; total += h.bits[l]; }
        o += 17;
        if (total > 256 || o + (size_t)total > sl) { why = "bad Huffman table"; return false; }
        memcpy(h.vals, s + o, (size_t)total);
        o += (size_t)total;
        if (!jpeg_build_huff(h)) { why = "inconsistent Huffman table"; return false; }
      }
    } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {  // SOF0 / SOF1: sequential, SOF2: progressive; Huffman
      if (sl < 6 || fr.have_sof) { why = "bad frame header"; return false; }
      fr.progressive = m == 0xC2;
      const int prec = s[0], nc = s[5];
      fr.h = s[1] << 8 | s[2];
      fr.w = s[3] << 8 | s[4];
      if (prec != 8) { why = "JPEG sample precision " + std::to_string(prec) + " (8 is read)"; return false; }
      if (nc != 1 || sl < 9) { why = std::to_string(nc) + "-component JPEG (one sample per pixel is read)"; return false; }
      fr.tq = s[8] & 3;
      fr.have_sof = true;
    } else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
      why = "JPEG process SOF" + std::to_string(m - 0xC0) + " (sequential and progressive Huffman are read)";
      return false;
    } else if (m == 0xDD) {  // DRI
      if (sl < 2) { why = "bad restart interval"; return false; }
      t.restart = s[0] << 8 | s[1];
    } else if (m == 0xDA) {  // SOS
      if (sl < 6 || s[0] != 1) { why = "bad scan header"; return false; }
      fr.td = s[2] >> 4; fr.ta = s[2] & 15;
      fr.ss = s[3]; fr.se = s[4]; fr.ah = s[5] >> 4; fr.al = s[5] & 15;
      if (fr.td > 3 || fr.ta > 3) { why = "bad scan header"; return false; }
      data_off = i + 2 + len;
      return !tables_only;
    }
    i += 2 + len;
  }
  why = tables_only ? "" : (start ? "JPEG stream ends without EOI" : "no scan in the JPEG stream");
  data_off = n;
  return tables_only;
}

// the entropy-coded segment: MSB-first bits, FF00 = a data byte FF; at a marker zeros are fed and the caller looks at what it is
struct JpegBits {
  const uint8_t* src; size_t n, pos;
  uint32_t acc = 0; int nbits = 0; bool hit_marker = false;
  JpegBits(const uint8_t* s, size_t n_, size_t p) : src(s), n(n_), pos(p) {}
  void fill() {
    while (nbits <= 24) {
      unsigned b = 0;
      if (!hit_marker && pos < n) {
        b = src[pos];
        if (b == 0xFF) {
          if (pos + 1 < n && src[pos + 1] == 0x00) pos += 2;
          else { hit_marker = true; b = 0; }
        } else {
          ++pos;
        }
      }
      acc |= (uint32_t)b << (24 - nbits);
      nbits += 8;
    }
  }
  int get(int c) {
    if (c == 0) return 0;
    if (nbits < c) fill();
    const int v = (int)(acc >> (32 - c));
    acc <<= c;
    nbits -= c;
    return v;
  }
  int decode(const JpegTables::Huff& h) {
    int code = 0;
    for (int l = 1; l <= 16; ++l) {
      code = code << 1 | get(1);
      if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
    }
    return -1;
  }
  // RSTn: byte-align, skip the marker
  bool restart() {
    acc = 0; nbits = 0; hit_marker = false;
    while (pos + 1 < n && !(src[pos] == 0xFF && src[pos + 1] >= 0xD0 && src[pos + 1] <= 0xD7)) ++pos;
    if (pos + 1 >= n) return false;
    pos += 2;
    return true;
  }
  // the marker that ends the scan (anything but FF00 / RSTn / fill bytes): its offset, or n
  size_t next_marker() const {
    size_t q = pos;
    while (q + 1 < n && !(src[q] == 0xFF && src[q + 1] != 0x00 && src[q + 1] != 0xFF && !(src[q + 1] >= 0xD0 && src[q + 1] <= 0xD7))) ++q;
    return q + 1 < n ? q : n;
  }
};
inline int jpeg_extend(int v, int s) { return s && v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

// libjpeg's jpeg_idct_islow on one dequantised block (natural order), output samples 0..255
void jpeg_idct_islow(const int* in, uint8_t* out, size_t stride) {
  constexpr int CB = 13, P1 = 2;
  constexpr long F0298 = 2446, F0390 = 3196, F0541 = 4433, F0765 = 6270, F0899 = 7373, F1175 = 9633, F1501 = 12299, F1847 = 15137,
                 F1961 = 16069, F2053 = 16819, F2562 = 20995, F3072 = 25172;
  auto descale = [](long x, int n) { return (x + (1L << (n - 1))) >> n; };
  long ws[64];
  for (int c = 0; c < 8; ++c) {
    const int* ip = in + c;
    long* wp = ws + c;
    if (!(ip[8] | ip[16] | ip[24] | ip[32] | ip[40] | ip[48] | ip[56])) {
      const long dc = (long)ip[0] * (1L << P1);
      for (int r = 0; r < 8; ++r) wp[8 * r] = dc;
      continue;
    }
    long z2 = ip[16], z3 = ip[48];
    long z1 = (z2 + z3) * F0541;
    long tmp2 = z1 + z3 * (-F1847), tmp3 = z1 + z2 * F0765;
    z2 = ip[0]; z3 = ip[32];
    long tmp0 = (z2 + z3) * (1L << CB), tmp1 = (z2 - z3) * (1L << CB);
    const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = ip[56]; tmp1 = ip[40]; tmp2 = ip[24]; tmp3 = ip[8];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    long z4 = tmp1 + tmp3;
    const long z5 = (z3 + z4) * F1175;
    tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
    z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    wp[0] = descale(tmp10 + tmp3, CB - P1); wp[56] = descale(tmp10 - tmp3, CB - P1);
    wp[8] = descale(tmp11 + tmp2, CB - P1); wp[48] = descale(tmp11 - tmp2, CB - P1);
    wp[16] = descale(tmp12 + tmp1, CB - P1); wp[40] = descale(tmp12 - tmp1, CB - P1);
    wp[24] = descale(tmp13 + tmp0, CB - P1); wp[32] = descale(tmp13 - tmp0, CB - P1);
  }
  for (int r = 0; r < 8; ++r) {
    const long* wp = ws + 8 * r;
    uint8_t* op = out + (size_t)r * stride;
    auto clamp = [](long v) { v += 128; return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); };
    long z2 = wp[2], z3 = wp[6];
    long z1 = (z2 + z3) * F0541;
    long tmp2 = z1 + z3 * (-F1847), tmp3 = z1 + z2 * F0765;
    long tmp0 = (wp[0] + wp[4]) * (1L << CB), tmp1 = (wp[0] - wp[4]) * (1L << CB);
    const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = wp[7]; tmp1 = wp[5]; tmp2 = wp[3]; tmp3 = wp[1];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    long z4 = tmp1 + tmp3;
    const long z5 = (z3 + z4) * F1175;
    tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
    z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    constexpr int SH = CB + P1 + 3;
    op[0] = clamp(descale(tmp10 + tmp3, SH)); op[7] = clamp(descale(tmp10 - tmp3, SH));
    op[1] = clamp(descale(tmp11 + tmp2, SH)); op[6] = clamp(descale(tmp11 - tmp2, SH));
    op[2] = clamp(descale(tmp12 + tmp1, SH)); op[5] = clamp(descale(tmp12 - tmp1, SH));
    op[3] = clamp(descale(tmp13 + tmp0, SH)); op[4] = clamp(descale(tmp13 - tmp0, SH));
  }
}

// The scans of a PROGRESSIVE stream (SOF2, one component): every scan carries one spectral band (DC alone, or AC coefficients ss..se in zigzag
// order) at one successive-approximation step (first pass: the bits down to `al`; refinement: bit `al`), tables may change between scans.
// The coefficients of the whole frame are collected, then dequantised and transformed like a sequential frame -- what libjpeg
// (jdphuff.c: decode_mcu_DC/AC_first/refine) leaves for a complete file; its block smoothing only acts on files whose scans are missing.
bool unjpeg_progressive(const uint8_t* src, size_t n, JpegTables& t, JpegFrame fr, size_t off, size_t bw, size_t bh, std::vector<int>& coef,
                        std::string& why) {
  coef.assign(bw * bh * 64, 0);  // zigzag order inside a block
  bool dc_seen = false;
  for (int scans = 0; scans < 4096; ++scans) {
    if (fr.ss > fr.se || fr.se > 63 || (fr.ss == 0 && fr.se != 0) || fr.al > 13 || (fr.ah != 0 && fr.ah != fr.al + 1)) { why = "bad progressive scan parameters"; return false; }
    const bool dc = fr.ss == 0, refine = fr.ah != 0;
    if (dc ? (!refine && !t.dc[fr.td].ok) : !t.ac[fr.ta].ok) { why = "missing Huffman table"; return false; }
    if (dc && !refine) dc_seen = true;
    JpegBits br(src, n, off);
    const int p1 = 1 << fr.al, m1 = -(1 << fr.al);
    int pred = 0, eobrun = 0;
    size_t mcu = 0;
    for (size_t b = 0; b < bw * bh; ++b, ++mcu) {
      if (t.restart && mcu && mcu % (size_t)t.restart == 0) {
        if (!br.restart()) { why = "missing restart marker"; return false; }
        pred = 0; eobrun = 0;
      }
      int* c = coef.data() + b * 64;
      if (dc) {
        if (!refine) {
          const int sz = br.decode(t.dc[fr.td]);
          if (sz < 0 || sz > 11) { why = "corrupt JPEG data (DC)"; return false; }
          pred += jpeg_extend(br.get(sz), sz);
          if (pred < -32768 || pred > 32767) { why = "corrupt JPEG data (DC predictor out of range)"; return false; }
          c[0] = pred * (1 << fr.al);
        } else if (br.get(1)) {
          c[0] |= p1;
        }
        continue;
      }
      const JpegTables::Huff& h = t.ac[fr.ta];
      if (!refine) {  // first pass over a band of AC coefficients
        if (eobrun > 0) { --eobrun; continue; }
        for (int k = fr.ss; k <= fr.se; ++k) {
          const int rs = br.decode(h);
          if (rs < 0) { why = "corrupt JPEG data (AC)"; return false; }
          const int r = rs >> 4, sz = rs & 15;
          if (sz) {
            k += r;
            if (k > fr.se) { why = "corrupt JPEG data (run)"; return false; }
            if (sz > 10) { why = "corrupt JPEG data (AC magnitude category above 10 at 8-bit precision)"; return false; }
            c[k] = jpeg_extend(br.get(sz), sz) * (1 << fr.al);
          } else if (r == 15) {
            k += 15;
          } else {  // end of band for this block and the next 2^r - 1 + (r bits)
            eobrun = (1 << r) + (r ? br.get(r) : 0) - 1;
            break;
          }
        }
        continue;
      }
      // refinement of a band: one more bit for every coefficient that is already nonzero, new +-1 coefficients in between
      auto correct = [&](int& v) {
        if (br.get(1) && (v & p1) == 0) v += v >= 0 ? p1 : m1;
      };
      int k = fr.ss;
      if (eobrun == 0) {
        for (; k <= fr.se; ++k) {
          const int rs = br.decode(h);
          if (rs < 0) { why = "corrupt JPEG data (AC)"; return false; }
          int r = rs >> 4, val = 0;
          const int sz = rs & 15;
          if (sz) {
            if (sz != 1) { why = "corrupt JPEG data (refinement)"; return false; }
            val = br.get(1) ? p1 : m1;
          } else if (r != 15) {
            eobrun = (1 << r) + (r ? br.get(r) : 0);
            break;  // the rest of the block: the end-of-band logic below
          }
          // over already-nonzero coefficients (a correction bit each) and r still-zero ones
          do {
            if (c[k] != 0) correct(c[k]);
            else if (--r < 0) break;
            ++k;
          } while (k <= fr.se);
          if (val) {
            if (k > fr.se) { why = "corrupt JPEG data (refinement run)"; return false; }
            c[k] = val;
          }
        }
      }
      if (eobrun > 0) {
        for (; k <= fr.se; ++k)
          if (c[k] != 0) correct(c[k]);
        --eobrun;
      }
    }
    // what follows the scan: tables, the next scan, or EOI
    const size_t mk = br.next_marker();
    if (mk >= n) { why = "JPEG stream ends without EOI"; return false; }
    fr.eoi = false;
    if (!jpeg_parse_segments(src, n, t, fr, off, false, why, mk)) return false;
    if (fr.eoi) {
      if (!dc_seen) { why = "progressive JPEG without a DC scan"; return false; }
      return true;
    }
  }
  why = "too many scans";
  return false;
}

// one strip / tile: `src` is its JPEG stream, `tables` the stream of the JPEGTables tag (may be empty); out = expect_w x expect_h
// 8-bit samples, row-major.  false + why on anything that is not a one-component Huffman-coded DCT stream of that size.
bool unjpeg(const uint8_t* src, size_t n, const std::vector<uint8_t>& tables, std::vector<uint8_t>& out, size_t expect_w, size_t expect_h,
            std::string& why) {
  JpegTables t;
  JpegFrame fr;
  size_t off = 0;
  if (!tables.empty() && !jpeg_parse_segments(tables.data(), tables.size(), t, fr, off, true, why)) {
    if (why.empty()) why = "bad JPEGTables";
    return false;
  }
  fr = JpegFrame();
  if (!jpeg_parse_segments(src, n, t, fr, off, false, why)) return false;
  if (!fr.have_sof || fr.w <= 0 || fr.h <= 0) { why = "no frame header"; return false; }
  if ((size_t)fr.w != expect_w || (size_t)fr.h > expect_h) { why = "JPEG frame size differs from the strip / tile"; return false; }
  const size_t bw = ((size_t)fr.w + 7) / 8, bh = ((size_t)fr.h + 7) / 8;
  std::vector<uint8_t> img(bw * 8 * bh * 8);
  int blk[64];
  if (fr.progressive) {
    std::vector<int> coef;
    if (!unjpeg_progressive(src, n, t, fr, off, bw, bh, coef, why)) return false;
    if (!t.have_q[fr.tq]) { why = "missing quantisation table"; return false; }
    for (size_t by = 0; by < bh; ++by)
      for (size_t bx = 0; bx < bw; ++bx) {
        const int* c = coef.data() + (by * bw + bx) * 64;
        for (int k = 0; k < 64; ++k) {
          // (an 8-bit frame's coefficients lie within +-2^13; a corrupt stream can shift a first-pass value up by 13 bits)
          if (c[k] < -32768 || c[k] > 32767) { why = "corrupt JPEG data (coefficient out of range)"; return false; }
          blk[kZigzag[k]] = c[k] * t.q[fr.tq][kZigzag[k]];
        }
        jpeg_idct_islow(blk, img.data() + (by * 8) * (bw * 8) + bx * 8, bw * 8);
      }
  } else {
    if (!t.have_q[fr.tq] || !t.dc[fr.td].ok || !t.ac[fr.ta].ok) { why = "missing quantisation / Huffman table"; return false; }
    JpegBits br(src, n, off);
    int pred = 0;
    size_t mcu = 0;
    for (size_t by = 0; by < bh; ++by)
      for (size_t bx = 0; bx < bw; ++bx, ++mcu) {
        if (t.restart && mcu && mcu % (size_t)t.restart == 0) {  // RSTn: byte-align, skip the marker, reset the predictor
          if (!br.restart()) { why = "missing restart marker"; return false; }
          pred = 0;
        }
        for (int k = 0; k < 64; ++k) blk[k] = 0;
        const int s = br.decode(t.dc[fr.td]);
        if (s < 0 || s > 11) { why = "corrupt JPEG data (DC)"; return false; }
        pred += jpeg_extend(br.get(s), s);
        // (an 8-bit sample's DC term lies within +-2^10 * 8; libjpeg range-limits, here a predictor that has run away is an error)
        if (pred < -32768 || pred > 32767) { why = "corrupt JPEG data (DC predictor out of range)"; return false; }
        blk[0] = pred * t.q[fr.tq][0];
        for (int k = 1; k < 64;) {
          const int rs = br.decode(t.ac[fr.ta]);
          if (rs < 0) { why = "corrupt JPEG data (AC)"; return false; }
          const int r = rs >> 4, sz = rs & 15;
          if (sz == 0) {
            if (r != 15) break;  // EOB
            k += 16;
            continue;
          }
          k += r;
          if (k > 63) { why = "corrupt JPEG data (run)"; return false; }
          if (sz > 10) { why = "corrupt JPEG data (AC magnitude category above 10 at 8-bit precision)"; return false; }
          blk[kZigzag[k]] = jpeg_extend(br.get(sz), sz) * t.q[fr.tq][kZigzag[k]];
          ++k;
        }
        jpeg_idct_islow(blk, img.data() + (by * 8) * (bw * 8) + bx * 8, bw * 8);
      }
  }
  out.assign(expect_w * expect_h, 0);
  for (size_t y = 0; y < (size_t)fr.h && y < expect_h; ++y) memcpy(out.data() + y * expect_w, img.data() + y * bw * 8, expect_w);
  return true;
}

// PackBits (TIFF compression 32773)
bool unpackbits(const uint8_t* src, size_t n, std::vector<uint8_t>& out, size_t expect) {
  out.clear();
  out.reserve(expect);
  size_t i = 0;
  while (i < n && out.size() < expect) {
    const int8_t c = (int8_t)src[i++];
    if (c >= 0) {
      const size_t len = (size_t)c + 1;
      if (i + len > n) return false;
      out.insert(out.end(), src + i, src + i + len);
      i += len;
    } else if (c != -128) {
      if (i >= n) return false;
      out.insert(out.end(), (size_t)(1 - c), src[i++]);
    }
  }
  return true;
}

// TIFF LZW (compression 5, TIFF 6.0 section 13): MSB-first codes of 9..12 bits, ClearCode 256,
// EndOfInformation 257, the code width grows one code EARLY (at 511 / 1023 / 2047 table entries)
bool unlzw(const uint8_t* src, size_t n, std::vector<uint8_t>& out, size_t expect) {
  out.clear();
  out.reserve(expect);
  struct Entry { uint32_t prev; uint32_t len; uint8_t first, last; };
  std::vector<Entry> tab(4096);
  for (uint32_t i = 0; i < 256; ++i) tab[i] = Entry{0xffffffffu, 1, (uint8_t)i, (uint8_t)i};
  uint32_t next = 258, width = 9, prev = 0xffffffffu;
  uint64_t acc = 0;
  int nbits = 0;
  size_t pos = 0;
  auto emit = [&](uint32_t code) {  // appends the string of `code`
    const uint32_t len = tab[code].len;
    const size_t base = out.size();
    out.resize(base + len);
    uint32_t c = code;
    for (uint32_t i = len; i-- > 0;) {
      out[base + i] = tab[c].last;
      c = tab[c].prev;
    }
  };
  while (out.size() < expect) {
    while (nbits < (int)width && pos < n) { acc = (acc << 8) | src[pos++]; nbits += 8; }
    if (nbits < (int)width) break;  // ran out of data (some writers omit EOI)
    const uint32_t code = (uint32_t)((acc >> (nbits - width)) & ((1u << width) - 1));
    nbits -= width;
    if (code == 257) break;
    if (code == 256) { next = 258; width = 9; prev = 0xffffffffu; continue; }
    if (prev == 0xffffffffu) {
      if (code >= 256) return false;
      emit(code);
    } else if (code < next) {
      emit(code);
      if (next < 4096) { tab[next] = Entry{prev, tab[prev].len + 1, tab[prev].first, tab[code].first}; ++next; }
    } else if (code == next && next < 4096) {  // KwKwK
      tab[next] = Entry{prev, tab[prev].len + 1, tab[prev].first, tab[prev].first};
      ++next;
      emit(code);
    } else {
      return false;
    }
    prev = code;
    if (next + 1 >= (1u << width) && width < 12) ++width;
  }
  if (out.size() > expect) out.resize(expect);
  return true;
}

// Deflate / "Adobe Deflate" (compression 8 and 32946): a zlib stream per strip or tile
bool uninflate(const uint8_t* src, size_t n, std::vector<uint8_t>& out, size_t expect) {
  out.assign(expect, 0);
  uLongf got = (uLongf)expect;
  const int rc = uncompress(out.data(), &got, src, (uLong)n);
  if (rc != Z_OK && rc != Z_BUF_ERROR) return false;  // Z_BUF_ERROR: more data than rows needed (padded strips)
  return rc == Z_OK || got == expect;
}

template <typename T>
T load_swapped(const uint8_t* p, bool swap) {
  uint8_t b[sizeof(T)];
  for (size_t i = 0; i < sizeof(T); ++i) b[i] = swap ? p[sizeof(T) - 1 - i] : p[i];
  T v;
  memcpy(&v, b, sizeof(T));
  return v;
}

template <typename T>
void store_swapped(uint8_t* p, T v, bool swap) {
  uint8_t b[sizeof(T)];
  memcpy(b, &v, sizeof(T));
  for (size_t i = 0; i < sizeof(T); ++i) p[i] = swap ? b[sizeof(T) - 1 - i] : b[i];
}

// undo the horizontal differencing predictor (tag 317 = 2) row by row, in place (integer samples, file byte order)
template <typename T>
void unpredict_rows(uint8_t* buf, size_t rows, size_t cols, bool swap) {
  for (size_t r = 0; r < rows; ++r) {
    uint8_t* row = buf + r * cols * sizeof(T);
    T acc = load_swapped<T>(row, swap);
    for (size_t c = 1; c < cols; ++c) {
      acc = (T)(acc + load_swapped<T>(row + c * sizeof(T), swap));
      store_swapped<T>(row + c * sizeof(T), acc, swap);
    }
  }
}

// undo the floating-point predictor (tag 317 = 3, TIFF Technical Note 3; one sample per pixel) row by row, in place: the
// writer split every row of IEEE samples into byte planes -- all most significant bytes first, whatever the byte order of the
// file -- and differenced the resulting row of BYTES horizontally.  On return the samples are back in the file's byte order.
void unpredict_float_rows(uint8_t* buf, size_t rows, size_t cols, size_t bytes_ps, bool file_le, std::vector<uint8_t>& tmp) {
  const size_t rb = cols * bytes_ps;
  tmp.resize(rb);
  for (size_t r = 0; r < rows; ++r) {
    uint8_t* row = buf + r * rb;
    for (size_t i = 1; i < rb; ++i) row[i] = (uint8_t)(row[i] + row[i - 1]);
    memcpy(tmp.data(), row, rb);
    for (size_t c = 0; c < cols; ++c)
      for (size_t b = 0; b < bytes_ps; ++b)  // plane b holds byte b counted from the most significant one
        row[c * bytes_ps + (file_le ? bytes_ps - 1 - b : b)] = tmp[b * cols + c];
  }
}

// interpretBytesAsDouble of the reference (TiffReader.cpp, used at :385,:430)
double sample_as_double(const uint8_t* p, uint32_t bps, uint32_t fmt, bool swap) {
  switch (bps) {
    case 8: return fmt == 2 ? (double)(int8_t)p[0] : (double)p[0];
    case 16: return fmt == 2 ? (double)load_swapped<int16_t>(p, swap) : (double)load_swapped<uint16_t>(p, swap);
    case 32:
      if (fmt == 3) return (double)load_swapped<float>(p, swap);
      return fmt == 2 ? (double)load_swapped<int32_t>(p, swap) : (double)load_swapped<uint32_t>(p, swap);
    case 64:
      if (fmt == 3) return load_swapped<double>(p, swap);
      return fmt == 2 ? (double)load_swapped<int64_t>(p, swap) : (double)load_swapped<uint64_t>(p, swap);
    default: return 0.0;
  }
}

bool host_is_le() {
  const uint16_t one = 1;
  return *reinterpret_cast<const uint8_t*>(&one) == 1;
}

// header + directory chain of one file
oi_status open_tiff(const char* path, File& f, std::vector<Ifd>& pages) {
  if (!f.d.open(path) || f.d.size() < 8) {
    oi::set_error("cannot read TIFF file '%s'", path);
    return OI_IO_ERROR;
  }
  if (f.d[0] == 'I' && f.d[1] == 'I') f.le = true;
  else if (f.d[0] == 'M' && f.d[1] == 'M') f.le = false;
  else { oi::set_error("'%s' is not a TIFF file", path); return OI_IO_ERROR; }
  const uint16_t magic = f.u16(2);
  size_t off = 0;
  if (magic == 42) {
    off = f.u32(4);
  } else if (magic == 43 && f.d.size() >= 16 && f.u16(4) == 8) {
    f.big = true;
    off = (size_t)f.u64(8);
  } else {
    oi::set_error("'%s': neither classic TIFF (42) nor BigTIFF (43)", path);
    return OI_IO_ERROR;
  }
  pages.clear();
  while (off) {  // TiffReader.cpp:175-177: depth = number of directories
    Ifd ifd;
    size_t next = 0;
    if (!parse_ifd(f, off, ifd, next)) { oi::set_error("'%s': corrupt IFD", path); return OI_IO_ERROR; }
    pages.push_back(ifd);
    if (pages.size() > (1u << 20)) { oi::set_error("'%s': IFD loop", path); return OI_IO_ERROR; }
    off = next;
  }
  if (pages.empty()) { oi::set_error("'%s': no directories", path); return OI_IO_ERROR; }
  return OI_OK;
}

oi_status check_first_page(const char* path, const Ifd& p0) {
  const bool valid_bps = p0.bps == 1 || p0.bps == 8 || p0.bps == 16 || p0.bps == 32 || p0.bps == 64;
  if (p0.w == 0 || p0.h == 0 || !valid_bps || p0.planar != 1 || p0.spp != 1) {  // TiffReader.cpp:167-173
    oi::set_error("'%s': unsupported TIFF (W=%u H=%u BPS=%u planar=%u SPP=%u)", path, p0.w, p0.h, p0.bps, p0.planar,
                  p0.spp);
    return OI_IO_ERROR;
  }
  return OI_OK;
}

// one directory -> one thresholded W x H slice
oi_status decode_page(const char* path, const File& f, const Ifd& pg, const Ifd& p0, size_t k, double threshold,
                      uint8_t* slice, std::vector<uint8_t>& buf) {
  const size_t W = p0.w, H = p0.h;
  if (pg.w != p0.w || pg.h != p0.h || pg.bps != p0.bps) {
    oi::set_error("'%s': page %zu differs in size/depth from page 0", path, k);
    return OI_IO_ERROR;
  }
  const bool known = pg.comp == 1 || pg.comp == 5 || pg.comp == 8 || pg.comp == 32946 || pg.comp == 32773 || pg.comp == 7;
  if (!known) {
    oi::set_error("'%s': TIFF compression %u is not supported (none, LZW, Deflate, PackBits and JPEG are)", path, pg.comp);
    return OI_IO_ERROR;
  }
  if (pg.comp == 7 && (pg.bps != 8 || pg.fmt != 1)) {
    oi::set_error("'%s': JPEG-compressed TIFF with %u-bit samples of format %u (8-bit unsigned is read)", path, pg.bps, pg.fmt);
    return OI_IO_ERROR;
  }
  const bool float_pred = pg.predictor == 3 && pg.fmt == 3 && (pg.bps == 32 || pg.bps == 64) && pg.spp == 1;
  if (pg.predictor != 1 && !(pg.predictor == 2 && pg.bps >= 8 && pg.fmt != 3) && !float_pred) {
    oi::set_error("'%s': TIFF predictor %u with %u-bit format %u is not supported", path, pg.predictor, pg.bps, pg.fmt);
    return OI_IO_ERROR;
  }
  const bool swap = f.le != host_is_le();
  std::vector<uint8_t> ptmp;
  const size_t bytes_ps = p0.bps >= 8 ? p0.bps / 8 : 1;
  // geometry of one chunk (strip or tile)
  const size_t cw = pg.tiled ? pg.tile_w : W;
  const size_t ch = pg.tiled ? pg.tile_h : (pg.rps == 0 || pg.rps > H ? H : pg.rps);
  if (cw == 0 || ch == 0) { oi::set_error("'%s': bad strip/tile geometry", path); return OI_IO_ERROR; }
  const size_t row_bytes = p0.bps == 1 ? (cw + 7) / 8 : cw * bytes_ps;
  const size_t across = pg.tiled ? (W + cw - 1) / cw : 1;
  const size_t down = (H + ch - 1) / ch;
  if (pg.offs.size() < across * down || pg.counts.size() < across * down) {
    oi::set_error("'%s': page %zu lacks strip/tile offsets", path, k);
    return OI_IO_ERROR;
  }
  for (size_t cy = 0; cy < down; ++cy)
    for (size_t cx = 0; cx < across; ++cx) {
      const size_t ci = cy * across + cx;
      const size_t rows_here = pg.tiled ? ch : std::min(ch, H - cy * ch);
      const size_t expect = row_bytes * rows_here;
      const uint64_t o = pg.offs[ci], n = pg.counts[ci];
      if (n > f.d.size() || o > f.d.size() - n) { oi::set_error("'%s': chunk outside file", path); return OI_IO_ERROR; }  // (o + n may wrap)
      const uint8_t* src = f.d.data() + o;
      size_t avail = n;
      if (pg.comp != 1 || pg.predictor != 1) {
        bool ok = true;
        std::string jwhy;
        if (pg.comp == 7) {
          ok = unjpeg(src, n, pg.jpeg_tables, buf, cw, rows_here, jwhy);
          if (!ok) { oi::set_error("'%s': JPEG strip / tile of page %zu: %s", path, k, jwhy.c_str()); return OI_IO_ERROR; }
        } else if (pg.comp == 32773) ok = unpackbits(src, n, buf, expect);
        else if (pg.comp == 5) ok = unlzw(src, n, buf, expect);
        else if (pg.comp == 8 || pg.comp == 32946) ok = uninflate(src, n, buf, expect);
        else buf.assign(src, src + std::min<size_t>(n, expect));
        if (!ok) { oi::set_error("'%s': corrupt compressed data (compression %u, page %zu)", path, pg.comp, k); return OI_IO_ERROR; }
        if (pg.predictor == 2) {
          const size_t rows_ok = buf.size() / row_bytes;
          if (bytes_ps == 1) unpredict_rows<uint8_t>(buf.data(), rows_ok, cw, false);
          else if (bytes_ps == 2) unpredict_rows<uint16_t>(buf.data(), rows_ok, cw, swap);
          else if (bytes_ps == 4) unpredict_rows<uint32_t>(buf.data(), rows_ok, cw, swap);
          else unpredict_rows<uint64_t>(buf.data(), rows_ok, cw, swap);
        } else if (float_pred) {
          unpredict_float_rows(buf.data(), buf.size() / row_bytes, cw, bytes_ps, f.le, ptmp);
        }
        src = buf.data();
        avail = buf.size();
      }
      // unsigned 8 / 16-bit samples (what tomograms are): `(double)v > threshold` is `v > floor(threshold)`,
      // a compare the compiler vectorises; every other type takes the generic per-sample path below
      const bool fast = (p0.bps == 8 || p0.bps == 16) && pg.fmt == 1 && avail >= row_bytes * rows_here;
      const long long ithr = threshold < 0.0 ? -1 : (threshold >= 65535.0 ? 65535 : (long long)std::floor(threshold));
      for (size_t r = 0; r < rows_here; ++r) {
        const size_t y = cy * ch + r;
        if (y >= H) break;
        if (fast) {
          const size_t x0 = cx * cw, nxr = std::min(cw, W - x0);
          uint8_t* dst = slice + y * W + x0;
          if (p0.bps == 8) {
            const uint8_t* sp = src + r * row_bytes;
            for (size_t c = 0; c < nxr; ++c) dst[c] = (long long)sp[c] > ithr ? 1 : 0;
          } else {
            const uint8_t* sp = src + r * row_bytes;
            for (size_t c = 0; c < nxr; ++c) {
              const unsigned v = f.le ? (unsigned)(sp[2 * c] | sp[2 * c + 1] << 8) : (unsigned)(sp[2 * c] << 8 | sp[2 * c + 1]);
              dst[c] = (long long)v > ithr ? 1 : 0;
            }
          }
          continue;
        }
        for (size_t c = 0; c < cw; ++c) {
          const size_t x = cx * cw + c;
          if (x >= W) break;
          double val = 0.0;
          if (p0.bps == 1) {  // TiffReader.cpp:419-426 (strips) / :376-381 (tiles)
            const size_t bi = r * row_bytes + c / 8;
            if (bi < avail) {
              const int bit = (int)(c % 8);
              // libtiff hands data back MSB-first unless the file says LSB2MSB and the
              // reference then picks the bit by FILLORDER (TiffReader.cpp:425)
              val = (double)((pg.fill == 2 ? (src[bi] >> bit) : (src[bi] >> (7 - bit))) & 1);
            }
          } else {
            const size_t bo = (r * cw + c) * bytes_ps;
            if (bo + bytes_ps <= avail) val = sample_as_double(src + bo, p0.bps, pg.fmt, swap);
          }
          slice[y * W + x] = val > threshold ? 1 : 0;  // TiffReader.cpp:434
        }
      }
    }
  return OI_OK;
}

}  // namespace

// planes [z0, z0 + nz_local) of a stack; nz_local < 0 = all planes from z0.  *nz reports the depth of the FILE.
extern "C" oi_status oi_read_tiff_planes(const char* path, double threshold, int z0, int nz_local, uint8_t** phase,
                                         int* nx, int* ny, int* nz, int* bits_per_sample) {
  if (!path || !phase || !nx || !ny || !nz || z0 < 0) return OI_BAD_ARG;
  *phase = nullptr;
  File f;
  std::vector<Ifd> pages;
  oi_status st = open_tiff(path, f, pages);
  if (st != OI_OK) return st;
  const Ifd& p0 = pages[0];
  if ((st = check_first_page(path, p0)) != OI_OK) return st;
  const size_t W = p0.w, H = p0.h, D = pages.size();
  // dimensions come from the file: they must fit the int interface and their product must not wrap before malloc sees it
  if (W == 0 || H == 0 || W > (size_t)INT_MAX || H > (size_t)INT_MAX || D > (size_t)INT_MAX || W > SIZE_MAX / H) {
    oi::set_error("'%s': image dimensions %zu x %zu x %zu are out of range", path, W, H, D);
    return OI_IO_ERROR;
  }
  if (nz_local < 0) nz_local = (int)D - z0;
  if (nz_local < 1 || (size_t)z0 + (size_t)nz_local > D) {
    oi::set_error("'%s': planes [%d, %d) are outside the stack of %zu", path, z0, z0 + nz_local, D);
    return OI_BAD_ARG;
  }
  if (W * H > SIZE_MAX / (size_t)nz_local) {
    oi::set_error("'%s': %zu x %zu x %d voxels overflow the address space", path, W, H, nz_local);
    return OI_IO_ERROR;
  }
  uint8_t* out = (uint8_t*)oi::host_alloc(W * H * (size_t)nz_local, false);
  if (!out) { oi::set_error("out of memory"); return OI_IO_ERROR; }
  // pages are independent: decode them on the host cores this process may use
  std::string first_error;
  std::mutex mu;
  std::atomic<int> failed{0};
  std::vector<std::vector<uint8_t>> bufs(oi::parallel_workers((size_t)nz_local));  // decode scratch per worker, owned by this call
  oi::parallel_for_w((size_t)nz_local, [&](size_t k, unsigned w) {
    if (failed.load()) return;
    std::vector<uint8_t>& buf = bufs[w];
    const oi_status s1 = decode_page(path, f, pages[z0 + k], p0, z0 + k, threshold, out + k * W * H, buf);
    if (s1 != OI_OK) {
      std::lock_guard<std::mutex> lk(mu);
      if (!failed.exchange((int)s1)) first_error = oi::last_error();  // (the message is thread-local)
    }
  });
  if (failed.load()) {
    oi_free_host(out);
    oi::set_error("%s", first_error.c_str());
    return (oi_status)failed.load();
  }
  *phase = out;
  *nx = (int)W; *ny = (int)H; *nz = (int)D;
  if (bits_per_sample) *bits_per_sample = (int)p0.bps;
  return OI_OK;
}

extern "C" oi_status oi_read_tiff(const char* path, double threshold, uint8_t** phase, int* nx, int* ny, int* nz,
                                  int* bits_per_sample) {
  return oi_read_tiff_planes(path, threshold, 0, -1, phase, nx, ny, nz, bits_per_sample);
}

// Numbered single-slice files <base><index, zero-padded to `digits`><suffix>, index = start_index ..
// start_index + num_files - 1 (TiffReader::readFileSequence, TiffReader.cpp:85-88, :209-285; the
// decode reads directory 0 of every file, :325-329); all slices must match the first one.
extern "C" oi_status oi_read_tiff_sequence(const char* base_pattern, int num_files, int start_index, int digits,
                                           const char* suffix, double threshold, uint8_t** phase, int* nx, int* ny,
                                           int* nz, int* bits_per_sample) {
  if (!base_pattern || !phase || !nx || !ny || !nz || num_files < 1 || digits < 0 || digits > 32) return OI_BAD_ARG;
  *phase = nullptr;
  uint8_t* out = nullptr;
  Ifd p0;
  auto name_of = [&](int k) {
    char num[64];
    snprintf(num, sizeof(num), "%0*d", digits, start_index + k);
    return std::string(base_pattern) + num + (suffix ? suffix : "");
  };
  // the first file fixes the slice size and type (and the output buffer) ...
  {
    const std::string name = name_of(0);
    File f;
    std::vector<Ifd> pages;
    oi_status st = open_tiff(name.c_str(), f, pages);
    if (st == OI_OK) {
      p0 = pages[0];
      st = check_first_page(name.c_str(), p0);
      if (st == OI_OK && (p0.w == 0 || p0.h == 0 || p0.w > (size_t)INT_MAX || p0.h > (size_t)INT_MAX ||
                          (size_t)p0.w > SIZE_MAX / (size_t)p0.h || (size_t)p0.w * p0.h > SIZE_MAX / (size_t)num_files)) {
        oi::set_error("'%s': image dimensions %llu x %llu x %d are out of range", name.c_str(), (unsigned long long)p0.w,
                      (unsigned long long)p0.h, num_files);
        st = OI_IO_ERROR;
      }
      if (st == OI_OK) {
        out = (uint8_t*)oi::host_alloc((size_t)p0.w * p0.h * (size_t)num_files, false);
        if (!out) { oi::set_error("out of memory"); st = OI_IO_ERROR; }
      }
    }
    std::vector<uint8_t> buf;
    if (st == OI_OK) st = decode_page(name.c_str(), f, pages[0], p0, 0, threshold, out, buf);
    if (st != OI_OK) { oi_free_host(out); return st; }
  }
  // ... the others are independent files: opened, decoded and thresholded on the host cores this process may use (round 4; a
  // tomogram stored as one TIFF per slice is the reference's readFileSequence case, TiffReader.cpp:209-285)
  std::string first_error;
  std::mutex mu;
  std::atomic<int> failed{0};
  std::vector<std::vector<uint8_t>> bufs(oi::parallel_workers((size_t)(num_files - 1)));  // decode scratch per worker, owned by this call
  oi::parallel_for_w((size_t)(num_files - 1), [&](size_t kk, unsigned w) {
    if (failed.load()) return;
    const int k = (int)kk + 1;
    const std::string name = name_of(k);
    File f;
    std::vector<Ifd> pages;
    std::vector<uint8_t>& buf = bufs[w];
    oi_status st = open_tiff(name.c_str(), f, pages);
    if (st == OI_OK) st = decode_page(name.c_str(), f, pages[0], p0, (size_t)k, threshold, out + (size_t)k * p0.w * p0.h, buf);
    if (st != OI_OK) {
      std::lock_guard<std::mutex> lk(mu);
      if (!failed.exchange((int)st)) first_error = oi::last_error();  // (the message is thread-local)
    }
  });
  if (failed.load()) {
    oi_free_host(out);
    oi::set_error("%s", first_error.c_str());
    return (oi_status)failed.load();
  }
  *phase = out;
  *nx = (int)p0.w; *ny = (int)p0.h; *nz = num_files;
  if (bits_per_sample) *bits_per_sample = (int)p0.bps;
  return OI_OK;
}

// planes [z0, z0 + nz_local) of a raw volume of nx * ny * nz voxels (only those bytes are touched)
extern "C" oi_status oi_read_raw_planes(const char* path, int nx, int ny, int nz, int dtype, double threshold, int z0,
                                        int nz_local, uint8_t** phase) {
  if (!path || !phase || nx < 1 || ny < 1 || nz < 1 || z0 < 0 || nz_local < 1 || z0 + nz_local > nz) return OI_BAD_ARG;
  *phase = nullptr;
  size_t bpv = 0;
  bool le = true;
  switch (dtype) {  // RawReader::getBytesPerVoxel / endianness table (RawReader.cpp:325-338)
    case OI_RAW_UINT8: case OI_RAW_INT8: bpv = 1; break;
    case OI_RAW_INT16_LE: case OI_RAW_UINT16_LE: bpv = 2; break;
    case OI_RAW_INT16_BE: case OI_RAW_UINT16_BE: bpv = 2; le = false; break;
    case OI_RAW_INT32_LE: case OI_RAW_UINT32_LE: case OI_RAW_FLOAT32_LE: bpv = 4; break;
    case OI_RAW_INT32_BE: case OI_RAW_UINT32_BE: case OI_RAW_FLOAT32_BE: bpv = 4; le = false; break;
    case OI_RAW_FLOAT64_LE: bpv = 8; break;
    case OI_RAW_FLOAT64_BE: bpv = 8; le = false; break;
    default: oi::set_error("unknown raw data type %d", dtype); return OI_BAD_ARG;
  }
  Bytes d;
  if (!d.open(path)) { oi::set_error("cannot read raw file '%s'", path); return OI_IO_ERROR; }
  const size_t ntot = (size_t)nx * ny * nz;
  if (d.size() != ntot * bpv) {  // the reference insists on an exact size match
    oi::set_error("'%s': size %zu does not match %dx%dx%d x %zu bytes", path, d.size(), nx, ny, nz, bpv);
    return OI_IO_ERROR;
  }
  const size_t n = (size_t)nx * ny * (size_t)nz_local, first = (size_t)nx * ny * (size_t)z0;
  uint8_t* out = (uint8_t*)oi::host_alloc(n, false);
  if (!out) { oi::set_error("out of memory"); return OI_IO_ERROR; }
  const bool swap = bpv > 1 && le != host_is_le();
  // `(double)sample > threshold`, strictly greater (RawReader.H:214), plane by plane on the host cores this process may use;
  // integers of up to 32 bits are compared in their own type (oi_hostpar.h: the same bits as the double compare)
  int cls = oi::kSampleUnsigned;
  switch (dtype) {
    case OI_RAW_INT8: case OI_RAW_INT16_LE: case OI_RAW_INT16_BE: case OI_RAW_INT32_LE: case OI_RAW_INT32_BE: cls = oi::kSampleSigned; break;
    case OI_RAW_FLOAT32_LE: case OI_RAW_FLOAT32_BE: case OI_RAW_FLOAT64_LE: case OI_RAW_FLOAT64_BE: cls = oi::kSampleFloat; break;
    default: break;
  }
  const size_t plane = (size_t)nx * ny;
  const uint8_t* base = d.data() + first * bpv;
  oi::parallel_for((size_t)nz_local, [&](size_t k) {
    (void)oi::threshold_run(base + k * plane * bpv, plane, (int)bpv, cls, swap, threshold, out + k * plane);
  });
  *phase = out;
  return OI_OK;
}

extern "C" oi_status oi_read_raw(const char* path, int nx, int ny, int nz, int dtype, double threshold,
                                 uint8_t** phase) {
  return oi_read_raw_planes(path, nx, ny, nz, dtype, threshold, 0, nz, phase);
}

// DatReader (src/io/DatReader.cpp:74-160 readFile, :204-241 threshold): little-endian
// 3 x int32 header (W, H, D) + uint16 voxels, x fastest; the threshold is compared as
// uint16 (the reference's DataType), extra trailing bytes are ignored.
extern "C" oi_status oi_read_dat(const char* path, double threshold, uint8_t** phase, int* nx, int* ny, int* nz) {
  if (!path || !phase || !nx || !ny || !nz) return OI_BAD_ARG;
  *phase = nullptr;
  Bytes d;
  if (!d.open(path) || d.size() < 12) { oi::set_error("cannot read DAT file '%s'", path); return OI_IO_ERROR; }
  const bool swap = !host_is_le();
  const int32_t w = load_swapped<int32_t>(d.data(), swap), h = load_swapped<int32_t>(d.data() + 4, swap),
                dep = load_swapped<int32_t>(d.data() + 8, swap);
  if (w <= 0 || h <= 0 || dep <= 0) { oi::set_error("'%s': invalid DAT header", path); return OI_IO_ERROR; }
  const size_t n = (size_t)w * h * dep;
  if (d.size() - 12 < 2 * n) { oi::set_error("'%s': DAT payload shorter than the header says", path); return OI_IO_ERROR; }
  uint8_t* out = (uint8_t*)oi::host_alloc(n, false);
  if (!out) { oi::set_error("out of memory"); return OI_IO_ERROR; }
  const uint16_t thr = (uint16_t)threshold;  // DatReader::threshold takes the raw type (uint16)
  const size_t plane = (size_t)w * h;
  const uint8_t* vox = d.data() + 12;
  oi::parallel_for((size_t)dep, [&](size_t k) {
    const uint8_t* sp = vox + 2 * k * plane;
    uint8_t* dp = out + k * plane;
    for (size_t v = 0; v < plane; ++v) dp[v] = oi::load_sample<uint16_t>(sp + 2 * v, swap) > thr ? 1 : 0;
  });
  *phase = out; *nx = w; *ny = h; *nz = dep;
  return OI_OK;
}

extern "C" void oi_free_host(void* p) {
  if (oi::g_host_free) oi::g_host_free(p);
  else free(p);
}
