// oi_hdf5.cpp -- dependency-light HDF5 dataset reader for the Diffusion front end.
//
// Replaces the reference's HDF5Reader (src/io/HDF5Reader.cpp: metadata :111-187, typed
// hyperslab reads :255-329, threshold :333-402) for the files that pipeline produces:
// rank-3 datasets of (u)int8/16/32/64 or float32/64, dims in C order (Z, Y, X) -> X is
// the unit-stride axis.  Implemented directly on the HDF5 file format (spec 1.x/2.0):
// superblock v0/v1, v1 object headers (+ continuation blocks), symbol-table groups (v1
// B-tree + local heap), contiguous, compact and chunked (v1 chunk B-tree) layouts, optional
// deflate / shuffle filters (zlib).  "Latest format" files (superblock v2/v3, v2 object
// headers, fractal-heap groups) are reported as unsupported, never guessed.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/openimpala_amd.h"

namespace oi {
void set_error(const char* fmt, ...);
}

namespace {

constexpr uint64_t kUndef = ~0ULL;

// read-only memory map of the file: a plane-range read only pages in the bytes it decodes
struct Mapped {
  const uint8_t* p = nullptr;
  size_t n = 0;
  Mapped() = default;
  Mapped(const Mapped&) = delete;
  Mapped& operator=(const Mapped&) = delete;
  ~Mapped() { if (p) munmap(const_cast<uint8_t*>(p), n); }
  bool open(const char* path) {
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size == 0) { ::close(fd); return false; }
    n = (size_t)st.st_size;
    void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (m == MAP_FAILED) { n = 0; return false; }
    p = (const uint8_t*)m;
    return true;
  }
  size_t size() const { return n; }
  const uint8_t& operator[](size_t i) const { return p[i]; }
};

struct H5 {
  Mapped f;
  uint64_t base = 0;
  std::string err;

  bool ok(uint64_t off, uint64_t n) const { return off <= f.size() && n <= f.size() - off; }
  uint64_t le(uint64_t off, int n) const {
    uint64_t v = 0;
    for (int i = 0; i < n; ++i) v |= (uint64_t)f[off + i] << (8 * i);
    return v;
  }
};

struct Msg {
  uint16_t type;
  uint64_t off;   // offset of the message body in the file
  uint16_t size;
};

// v1 object header: all messages, following continuation messages (type 0x10)
bool read_object_header(const H5& h, uint64_t addr, std::vector<Msg>& out) {
  if (!h.ok(addr, 16)) return false;
  if (h.f[addr] != 1) return false;  // version 2 headers start with "OHDR"
  const int nmsgs = (int)h.le(addr + 2, 2);
  struct Block { uint64_t off, len; };
  std::vector<Block> blocks{{addr + 16, h.le(addr + 8, 4)}};
  for (size_t b = 0; b < blocks.size() && (int)out.size() < nmsgs; ++b) {
    uint64_t p = blocks[b].off;
    const uint64_t end = blocks[b].off + blocks[b].len;
    if (!h.ok(blocks[b].off, blocks[b].len)) return false;
    while (p + 8 <= end && (int)out.size() < nmsgs) {
      Msg m;
      m.type = (uint16_t)h.le(p, 2);
      m.size = (uint16_t)h.le(p + 2, 2);
      m.off = p + 8;
      if (m.off + m.size > end) return false;
      if (m.type == 0x10 && m.size >= 16) blocks.push_back({h.base + h.le(m.off, 8), h.le(m.off + 8, 8)});
      out.push_back(m);
      p = m.off + m.size;
    }
  }
  return true;
}

// look `name` up in a symbol-table group: walks the v1 group B-tree to its SNOD leaves
bool group_lookup(const H5& h, uint64_t btree, uint64_t heap, const std::string& name, uint64_t& obj_addr) {
  if (!h.ok(heap, 32) || memcmp(&h.f[heap], "HEAP", 4) != 0) return false;
  const uint64_t heap_data = h.base + h.le(heap + 24, 8);
  std::vector<uint64_t> todo{btree};
  size_t visited = 0;
  while (!todo.empty() && visited++ < 100000) {
    const uint64_t n = todo.back();
    todo.pop_back();
    if (!h.ok(n, 24)) return false;
    if (memcmp(&h.f[n], "TREE", 4) == 0) {
      if (h.f[n + 4] != 0) return false;  // node type 0 = group
      const int used = (int)h.le(n + 6, 2);
      // keys and children alternate after the 24-byte header: key0 child0 key1 child1 ... key_n
      for (int c = 0; c < used; ++c) {
        const uint64_t child = n + 24 + 8 + (uint64_t)c * 16;
        if (!h.ok(child, 8)) return false;
        todo.push_back(h.base + h.le(child, 8));
      }
    } else if (memcmp(&h.f[n], "SNOD", 4) == 0) {
      const int nsym = (int)h.le(n + 6, 2);
      for (int s = 0; s < nsym; ++s) {
        const uint64_t e = n + 8 + (uint64_t)s * 40;
        if (!h.ok(e, 40)) return false;
        const uint64_t noff = heap_data + h.le(e, 8);
        if (noff >= h.f.size()) return false;
        const char* nm = (const char*)&h.f[noff];
        const size_t maxlen = h.f.size() - noff;
        if (strnlen(nm, maxlen) < maxlen && name == nm) {
          obj_addr = h.base + h.le(e + 8, 8);
          return true;
        }
      }
    } else {
      return false;
    }
  }
  return false;
}

struct Dataset {
  int rank = 0;
  uint64_t dims[4] = {0, 0, 0, 0};
  int type_class = -1, type_size = 0;
  bool big_endian = false, is_signed = false;
  int layout = -1;  // 0 compact, 1 contiguous, 2 chunked
  uint64_t data_addr = kUndef, data_size = 0;
  uint64_t compact_off = 0;
  uint64_t chunk_btree = kUndef;
  int chunk_rank = 0;  // rank + 1
  uint64_t chunk_dims[5] = {0};
  bool f_deflate = false, f_shuffle = false, f_other = false;
};

bool parse_dataset(const H5& h, uint64_t addr, Dataset& d, std::string& why) {
  std::vector<Msg> msgs;
  if (!read_object_header(h, addr, msgs)) { why = "unsupported object header (only version 1 headers are read)"; return false; }
  for (const Msg& m : msgs) {
    const uint64_t o = m.off;
    if (m.type == 0x0001) {  // dataspace
      const int ver = h.f[o];
      d.rank = h.f[o + 1];
      const uint64_t dimoff = o + (ver == 1 ? 8 : 4);
      if (d.rank > 4) { why = "rank > 4"; return false; }
      for (int i = 0; i < d.rank; ++i) d.dims[i] = h.le(dimoff + 8ULL * i, 8);
    } else if (m.type == 0x0003) {  // datatype
      const uint32_t cv = (uint32_t)h.le(o, 4);
      d.type_class = cv & 0x0f;
      const uint32_t bits0 = (cv >> 8) & 0xff;
      d.big_endian = bits0 & 1;
      d.is_signed = d.type_class == 0 && (bits0 & 0x08);
      d.type_size = (int)h.le(o + 4, 4);
    } else if (m.type == 0x0008) {  // data layout
      const int ver = h.f[o];
      if (ver != 3) { why = "data layout message version " + std::to_string(ver) + " (only 3 is read)"; return false; }
      d.layout = h.f[o + 1];
      if (d.layout == 1) {
        d.data_addr = h.le(o + 2, 8);
        d.data_size = h.le(o + 10, 8);
      } else if (d.layout == 0) {
        d.data_size = h.le(o + 2, 2);
        d.compact_off = o + 4;
      } else if (d.layout == 2) {
        d.chunk_rank = h.f[o + 2];
        d.chunk_btree = h.le(o + 3, 8);
        if (d.chunk_rank > 5) { why = "chunk rank > 5"; return false; }
        for (int i = 0; i < d.chunk_rank; ++i) d.chunk_dims[i] = h.le(o + 11 + 4ULL * i, 4);
      }
    } else if (m.type == 0x000B) {  // filter pipeline
      const int ver = h.f[o];
      const int nf = h.f[o + 1];
      uint64_t p = o + (ver == 1 ? 8 : 2);
      for (int i = 0; i < nf; ++i) {
        const int id = (int)h.le(p, 2);
        int namelen = (ver == 1 || id >= 256) ? (int)h.le(p + 2, 2) : 0;
        const uint64_t hdr = (ver == 1 || id >= 256) ? 8 : 6;
        const int ncd = (int)h.le(p + hdr - 2, 2);
        if (ver == 1) namelen = (namelen + 7) & ~7;
        uint64_t adv = hdr + namelen + 4ULL * ncd;
        if (ver == 1 && (ncd & 1)) adv += 4;
        if (id == 1) d.f_deflate = true;
        else if (id == 2) d.f_shuffle = true;
        else d.f_other = true;
        p += adv;
      }
    }
  }
  if (d.rank == 0 || d.type_class < 0 || d.layout < 0) { why = "dataset lacks dataspace/datatype/layout messages"; return false; }
  return true;
}

double elem_as_double(const uint8_t* p, const Dataset& d) {
  uint8_t b[8];
  for (int i = 0; i < d.type_size; ++i) b[i] = d.big_endian ? p[d.type_size - 1 - i] : p[i];
  if (d.type_class == 1) {
    if (d.type_size == 4) { float v; memcpy(&v, b, 4); return v; }
    double v; memcpy(&v, b, 8); return v;
  }
  switch (d.type_size) {
    case 1: return d.is_signed ? (double)(int8_t)b[0] : (double)b[0];
    case 2: { uint16_t v; memcpy(&v, b, 2); return d.is_signed ? (double)(int16_t)v : (double)v; }
    case 4: { uint32_t v; memcpy(&v, b, 4); return d.is_signed ? (double)(int32_t)v : (double)v; }
    default: { uint64_t v; memcpy(&v, b, 8); return d.is_signed ? (double)(int64_t)v : (double)v; }
  }
}

// walks the v1 chunk B-tree and hands every chunk to `fn(offsets, addr, nbytes)`
template <typename F>
bool for_each_chunk(const H5& h, const Dataset& d, uint64_t node, F&& fn, int depth = 0) {
  if (depth > 32 || !h.ok(node, 24) || memcmp(&h.f[node], "TREE", 4) != 0 || h.f[node + 4] != 1) return false;
  const int level = h.f[node + 5];
  const int used = (int)h.le(node + 6, 2);
  const uint64_t keysz = 8 + 8ULL * d.chunk_rank;
  uint64_t p = node + 24;
  for (int c = 0; c < used; ++c) {
    if (!h.ok(p, keysz + 8)) return false;
    const uint64_t nbytes = h.le(p, 4);
    uint64_t offs[5];
    for (int i = 0; i < d.chunk_rank; ++i) offs[i] = h.le(p + 8 + 8ULL * i, 8);
    const uint64_t child = h.base + h.le(p + keysz, 8);
    if (level > 0) {
      if (!for_each_chunk(h, d, child, fn, depth + 1)) return false;
    } else if (!fn(offs, child, nbytes)) {
      return false;
    }
    p += keysz + 8;
  }
  return true;
}


}  // namespace

// planes [z0, z0 + nz_local) of the dataset (nz_local < 0: all from z0; the hyperslab each rank of the reference
// reads, HDF5Reader.cpp:255-331); *nz = depth of the dataset
extern "C" oi_status oi_read_hdf5_planes(const char* path, const char* dataset, double threshold, int z0, int nz_local,
                                         uint8_t** phase, int* nx, int* ny, int* nz) {
  if (!path || !dataset || !phase || !nx || !ny || !nz || z0 < 0) return OI_BAD_ARG;
  *phase = nullptr;
  H5 h;
  if (!h.f.open(path)) { oi::set_error("cannot read HDF5 file '%s'", path); return OI_IO_ERROR; }
  static const uint8_t sig[8] = {0x89, 'H', 'D', 'F', '\r', '\n', 0x1a, '\n'};
  uint64_t sb = kUndef;
  for (uint64_t off = 0; off + 8 <= h.f.size(); off = off ? off * 2 : 512)  // signature at 0, 512, 1024, ...
    if (memcmp(&h.f[off], sig, 8) == 0) { sb = off; break; }
  if (sb == kUndef) { oi::set_error("'%s' is not an HDF5 file", path); return OI_IO_ERROR; }
  const int sbver = h.f[sb + 8];
  if (sbver > 1) {
    oi::set_error("'%s': HDF5 superblock version %d (\"latest format\") is not supported by the built-in reader; "
                  "rewrite the file with the default (earliest) format", path, sbver);
    return OI_IO_ERROR;
  }
  if (h.f[sb + 13] != 8 || h.f[sb + 14] != 8) { oi::set_error("'%s': only 8-byte offsets/lengths are supported", path); return OI_IO_ERROR; }
  const uint64_t p = sb + 24 + (sbver == 1 ? 4 : 0);
  h.base = h.le(p, 8);
  const uint64_t root_entry = p + 32;  // after base, free-space, end-of-file, driver addresses
  uint64_t obj = h.base + h.le(root_entry + 8, 8);

  // walk the dataset path through symbol-table groups
  std::string rest = dataset;
  while (!rest.empty() && rest[0] == '/') rest.erase(0, 1);
  while (!rest.empty()) {
    const size_t slash = rest.find('/');
    const std::string part = rest.substr(0, slash);
    rest = slash == std::string::npos ? "" : rest.substr(slash + 1);
    if (part.empty()) continue;
    std::vector<Msg> msgs;
    if (!read_object_header(h, obj, msgs)) { oi::set_error("'%s': unsupported group object header", path); return OI_IO_ERROR; }
    uint64_t bt = kUndef, hp = kUndef;
    for (const Msg& m : msgs)
      if (m.type == 0x0011) { bt = h.base + h.le(m.off, 8); hp = h.base + h.le(m.off + 8, 8); }
    if (bt == kUndef) { oi::set_error("'%s': group without a symbol table (new-style groups are not supported)", path); return OI_IO_ERROR; }
    uint64_t next = 0;
    if (!group_lookup(h, bt, hp, part, next)) { oi::set_error("'%s': no object '%s' (dataset '%s')", path, part.c_str(), dataset); return OI_IO_ERROR; }
    obj = next;
  }

  Dataset d;
  std::string why;
  if (!parse_dataset(h, obj, d, why)) { oi::set_error("'%s' [%s]: %s", path, dataset, why.c_str()); return OI_IO_ERROR; }
  if (d.rank != 3) {  // HDF5Reader.cpp:126-130
    oi::set_error("'%s': dataset '%s' has rank %d, expected 3", path, dataset, d.rank);
    return OI_IO_ERROR;
  }
  const bool int_ok = d.type_class == 0 && (d.type_size == 1 || d.type_size == 2 || d.type_size == 4 || d.type_size == 8);
  const bool flt_ok = d.type_class == 1 && (d.type_size == 4 || d.type_size == 8);
  if (!int_ok && !flt_ok) {
    oi::set_error("'%s': unsupported HDF5 data type (class %d, %d bytes)", path, d.type_class, d.type_size);
    return OI_IO_ERROR;
  }
  if (d.f_other) { oi::set_error("'%s': only the deflate and shuffle filters are supported", path); return OI_IO_ERROR; }
  const uint64_t Z = d.dims[0], Y = d.dims[1], X = d.dims[2];  // C order: (Z, Y, X), HDF5Reader.cpp:135-138
  if (X == 0 || Y == 0 || Z == 0 || X > 0x7fffffffULL || Y > 0x7fffffffULL || Z > 0x7fffffffULL) {
    oi::set_error("'%s': invalid dataset dimensions", path);
    return OI_IO_ERROR;
  }
  if (nz_local < 0) nz_local = (int)Z - z0;
  if (nz_local < 1 || (uint64_t)z0 + (uint64_t)nz_local > Z) {
    oi::set_error("'%s': planes [%d, %d) are outside the dataset of %llu", path, z0, z0 + nz_local, (unsigned long long)Z);
    return OI_BAD_ARG;
  }
  const uint64_t za = (uint64_t)z0, zb = za + (uint64_t)nz_local;  // planes [za, zb)
  const uint64_t n = X * Y * (zb - za), es = (uint64_t)d.type_size;
  uint8_t* out = (uint8_t*)calloc(n, 1);  // unallocated storage reads as fill value 0
  if (!out) { oi::set_error("out of memory"); return OI_IO_ERROR; }
  auto fail = [&](const char* msg) { free(out); oi::set_error("'%s' [%s]: %s", path, dataset, msg); return OI_IO_ERROR; };

  if (d.layout == 1 || d.layout == 0) {
    const uint64_t src = d.layout == 1 ? h.base + d.data_addr : d.compact_off;
    if (d.layout == 1 && d.data_addr == kUndef) {  // never written: all fill value
      for (uint64_t v = 0; v < n; ++v) out[v] = 0.0 > threshold ? 1 : 0;
    } else {
      if (!h.ok(src, X * Y * Z * es)) return fail("raw data outside the file");
      const uint64_t first = X * Y * za;
      for (uint64_t v = 0; v < n; ++v) out[v] = elem_as_double(&h.f[src + (first + v) * es], d) > threshold ? 1 : 0;
    }
  } else if (d.layout == 2) {
    if (d.chunk_rank != 4) return fail("chunk rank does not match the dataset rank");
    if (d.chunk_dims[3] != es) return fail("chunk element size mismatch");
    const uint64_t cz = d.chunk_dims[0], cy = d.chunk_dims[1], cx = d.chunk_dims[2];
    const uint64_t cbytes = cz * cy * cx * es;
    if (0.0 > threshold) memset(out, 1, n);  // chunks that were never written hold the fill value 0
    std::vector<uint8_t> raw(cbytes), tmp(cbytes);
    bool good = true;
    std::string cerr;
    if (d.chunk_btree != kUndef) {
      good = for_each_chunk(h, d, h.base + d.chunk_btree, [&](const uint64_t* offs, uint64_t addr, uint64_t nbytes) {
        if (offs[0] >= zb || offs[0] + cz <= za) return true;  // chunk outside the requested planes
        if (!h.ok(addr, nbytes)) { cerr = "chunk outside the file"; return false; }
        const uint8_t* src = &h.f[addr];
        if (d.f_deflate) {
          uLongf dl = (uLongf)cbytes;
          if (uncompress(raw.data(), &dl, src, (uLong)nbytes) != Z_OK || dl != cbytes) { cerr = "deflate failed"; return false; }
          src = raw.data();
        } else if (nbytes < cbytes) { cerr = "short chunk"; return false; }
        if (d.f_shuffle && es > 1) {  // un-shuffle: byte planes -> elements
          const uint64_t ne = cbytes / es;
          for (uint64_t e = 0; e < ne; ++e)
            for (uint64_t b = 0; b < es; ++b) tmp[e * es + b] = src[b * ne + e];
          src = tmp.data();
        }
        for (uint64_t k = 0; k < cz && offs[0] + k < zb; ++k) {
          if (offs[0] + k < za) continue;
          for (uint64_t j = 0; j < cy && offs[1] + j < Y; ++j)
            for (uint64_t i = 0; i < cx && offs[2] + i < X; ++i) {
              const uint64_t se = (k * cy + j) * cx + i;
              out[((offs[0] + k - za) * Y + offs[1] + j) * X + offs[2] + i] = elem_as_double(src + se * es, d) > threshold ? 1 : 0;
            }
        }
        return true;
      });
    }
    if (!good) return fail(cerr.empty() ? "corrupt chunk index" : cerr.c_str());
  } else {
    return fail("unknown data layout class");
  }
  *phase = out;
  *nx = (int)X; *ny = (int)Y; *nz = (int)Z;
  return OI_OK;
}

extern "C" oi_status oi_read_hdf5(const char* path, const char* dataset, double threshold, uint8_t** phase, int* nx,
                                  int* ny, int* nz) {
  return oi_read_hdf5_planes(path, dataset, threshold, 0, -1, phase, nx, ny, nz);
}
